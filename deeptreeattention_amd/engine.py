"""Fused train step for the Hang2020 hot path: forward + class-weighted cross-entropy + backward + Adam as four
C-ABI calls on flat device buffers, optionally data-parallel (one process per GPU, RCCL all-reduce of the flat
gradient on a side HIP stream overlapped with the first conv's weight gradient).

Replaces, for one model, the reference's TreeModel.training_step + loss.backward() + Adam.step()
(/root/reference/src/main.py:71-80, :135-137) and Lightning's DDP wrapper (train.py:89-98, `gpus>1`).
BatchNorm statistics stay per-rank (the reference does not enable sync_batchnorm); the loss each rank
back-propagates is its own weighted mean (DDP semantics: gradients are averaged over ranks).
"""
import ctypes as C

import torch

from . import _lib
from . import Hang2020 as H
from .dist import GradSync


class FusedTrainer:
    """Owns flat fp32 parameter / gradient / Adam-moment buffers; the model's Parameters become views of the
    flat parameter buffer (state_dict keys and shapes are unchanged).

    model        : deeptreeattention_amd.Hang2020.{Hang2020, spectral_network, spatial_network, vanilla_CNN}
    loss_weight  : (classes,) tensor or None (= ones, reference src/main.py:69)
    process_group: torch.distributed group (None = single process)
    """

    def __init__(self, model, lr, loss_weight=None, betas=(0.9, 0.999), eps=1e-8, process_group=None,
                 overlap_comm=True, keep_grads=False):
        if not isinstance(model, H._Net):
            raise TypeError("FusedTrainer needs a deeptreeattention_amd network module")
        self.model = model
        self.lr, self.betas, self.eps = float(lr), betas, float(eps)
        self.step_count = 0
        self.keep_grads = bool(keep_grads)   # True: gradients stay readable (grad_of) after train_step
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.overlap = overlap_comm and self.world > 1
        self.hang = model._net_code == _lib.NET_HANG2020
        plist = model._param_list()
        dev = plist[-1].device
        if dev.type != "cuda":
            raise RuntimeError("FusedTrainer needs the model on a ROCm device (model.cuda()); there is no CPU path")
        self.device = dev
        fp32 = [p for p in plist if p.dtype == torch.float32]
        # flat layout: [everything except the first conv's weights | first conv's weights] so that the gradient
        # all-reduce of the first part overlaps with the first conv's weight-gradient kernel (the last to finish)
        first = [p for kind, mod, names in model._subnets() for p in [H._get(mod, "conv1.conv_layer.weight")]]
        first_ids = {id(p) for p in first}
        order = [p for p in fp32 if id(p) not in first_ids] + first
        n = sum(p.numel() for p in order)
        self.n = n
        self.split = n - sum(p.numel() for p in first)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self._grads_clear = True       # flat_g / alpha_g hold zeros (kept so by dta_adam_step_zero_grad)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._gview = {}
        off = 0
        with torch.no_grad():
            for p in order:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view(p.shape)
                self._gview[id(p)] = self.flat_g[off:off + k].view(p.shape)
                off += k
        if self.hang:
            self.alpha = model.alpha
            self.alpha_g = torch.zeros((), dtype=torch.float64, device=dev)
            self.alpha_m = torch.zeros((), dtype=torch.float64, device=dev)
            self.alpha_v = torch.zeros((), dtype=torch.float64, device=dev)
        self.loss_weight = None if loss_weight is None else loss_weight.to(dev, torch.float32).contiguous()
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._ws = None
        self._ws_key = None
        self._side = torch.cuda.Stream(device=dev) if self.overlap else None
        self.sync = GradSync(self.world, self.pg, self._side)
        if self.world > 1:
            self.broadcast_parameters()

    # ------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        """DDP start-up semantics: every rank starts from rank `src`'s parameters and buffers."""
        self.sync.broadcast([self.flat_p] + ([self.alpha.data] if self.hang else []) + list(self.model.buffers()), src)

    def grad_of(self, param):
        """Gradient view (inside the flat gradient buffer) of one of the model's fp32 parameters.  After train_step
        it holds the step's gradient only when the trainer was built with keep_grads=True (by default the optimizer
        pass clears the buffer, as optimizer.zero_grad() would)."""
        return self._gview[id(param)]

    def _structs(self):
        m = self.model
        subnets = m._subnets()
        nets = (_lib.SubnetParams * len(subnets))()
        grads = (_lib.SubnetGrads * len(subnets))()
        for i, (kind, mod, names) in enumerate(subnets):
            tensors = {n: H._get(mod, n) for n in names}
            gt = {}
            for n in names:
                if self.hang and ("classifier1" in n or "classifier2" in n):
                    continue   # Hang2020.forward keeps only the last heads (reference :256-257): no gradient
                gt[n] = self._gview[id(tensors[n])]
            for Lv in (1, 2, 3):
                bn = H._get(mod, f"conv{Lv}.bn1")
                tensors[f"conv{Lv}.bn1.running_mean"] = bn.running_mean
                tensors[f"conv{Lv}.bn1.running_var"] = bn.running_var
                tensors[f"conv{Lv}.bn1.num_batches_tracked"] = bn.num_batches_tracked
            H._fill_struct(nets[i], kind, tensors, False)
            H._fill_struct(grads[i], kind, gt, True)
        return nets, grads

    def _prepare(self, x):
        B, bands, Hh, Ww = x.shape
        m = self.model
        key = (B, bands, Hh, Ww, m.precision, m.training)
        if key != self._ws_key:
            L = _lib.lib()
            self.desc = _lib.NetDesc(B, bands, Hh, Ww, m._classes, m._net_code, _lib.dtype_code(m.precision),
                                     1 if m.training else 0, 4 if m._net_code in (_lib.NET_HANG2020, _lib.NET_VANILLA) else 7,
                                     H.BN_MOMENTUM, H.BN_EPS)
            nbytes = L.dta_net_workspace_bytes(C.byref(self.desc))
            if nbytes == 0:
                raise RuntimeError("dta_net_workspace_bytes: " + L.dta_last_error().decode())
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.logits = torch.empty(B, m._classes, dtype=torch.float32, device=self.device)
            self.dlogits = torch.empty_like(self.logits)
            self.ce_scratch = torch.empty(B + 1, dtype=torch.float32, device=self.device)
            self.nets, self.grads = self._structs()
            self._ws_key = key

    def train_step(self, x, y):
        """One optimisation step on the batch (x: float32 NCHW on the device, y: int64 labels).  Returns the loss
        as a 0-d device tensor (no host sync)."""
        L = _lib.lib()
        m = self.model
        if m._net_code not in (_lib.NET_HANG2020, _lib.NET_VANILLA):
            raise RuntimeError("train_step supports Hang2020 and vanilla_CNN (single-score models)")
        x = H._check_input(x)
        if y.dtype != torch.int64 or not y.is_cuda:
            y = y.to(self.device, torch.int64)
        self._prepare(x)
        st = _lib.current_stream_ptr()
        d = C.byref(self.desc)
        alpha = _lib.ptr(self.alpha) if self.hang else None
        table = _lib.ScoreTable()
        _lib.check(L.dta_net_forward(d, self.nets, alpha, _lib.ptr(x), _lib.ptr(self._ws), C.byref(table),
                                     _lib.ptr(self.logits), st), "dta_net_forward")
        _lib.check(L.dta_weighted_ce(_lib.ptr(self.logits), _lib.ptr(y), _lib.ptr(self.loss_weight), x.shape[0],
                                     m._classes, _lib.ptr(self.loss), _lib.ptr(self.dlogits), _lib.ptr(self.ce_scratch),
                                     st), "dta_weighted_ce")
        dalpha = _lib.ptr(self.alpha_g) if self.hang else None
        if not self._grads_clear:      # C-ABI contract: gradient buffers (and dalpha) arrive zero-filled
            self.flat_g.zero_()
            if self.hang:
                self.alpha_g.zero_()
        if self.world == 1:
            _lib.check(L.dta_net_backward(d, self.nets, alpha, _lib.ptr(self._ws), C.byref(table),
                                          _lib.ptr(self.dlogits), self.grads, dalpha, 3, st), "dta_net_backward")
        else:
            # phase 1: everything but the first conv's weight gradient; its all-reduce (side stream when overlap is
            # on) runs while phase 2, the first conv's weight gradient, is computed
            _lib.check(L.dta_net_backward(d, self.nets, alpha, _lib.ptr(self._ws), C.byref(table),
                                          _lib.ptr(self.dlogits), self.grads, dalpha, 1, st), "dta_net_backward")
            self.sync.reduce_early(self.flat_g[:self.split], self.alpha_g if self.hang else None)
            _lib.check(L.dta_net_backward(d, self.nets, alpha, _lib.ptr(self._ws), C.byref(table),
                                          _lib.ptr(self.dlogits), self.grads, dalpha, 2, st), "dta_net_backward")
            self.sync.reduce_late(self.flat_g[self.split:])
            self.sync.finish()
        self.step_count += 1
        # default: step + zero_grad in one pass, so the next backward finds its gradient buffers already cleared
        adam = L.dta_adam_step if self.keep_grads else L.dta_adam_step_zero_grad
        _lib.check(adam(_lib.ptr(self.flat_p), _lib.ptr(self.flat_g), _lib.ptr(self.flat_m), _lib.ptr(self.flat_v),
                        self.n,
                        _lib.ptr(self.alpha) if self.hang else None,
                        _lib.ptr(self.alpha_g) if self.hang else None,
                        _lib.ptr(self.alpha_m) if self.hang else None,
                        _lib.ptr(self.alpha_v) if self.hang else None,
                        self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                        self.sync.grad_scale, st), "dta_adam_step")
        self._grads_clear = not self.keep_grads
        return self.loss

    def forward_loss(self, x, y):
        """Forward + loss only (validation_step, reference src/main.py:82-94); returns (logits, loss)."""
        L = _lib.lib()
        x = H._check_input(x)
        self._prepare(x)
        st = _lib.current_stream_ptr()
        table = _lib.ScoreTable()
        _lib.check(L.dta_net_forward(C.byref(self.desc), self.nets, _lib.ptr(self.alpha) if self.hang else None,
                                     _lib.ptr(x), _lib.ptr(self._ws), C.byref(table), _lib.ptr(self.logits), st),
                   "dta_net_forward")
        _lib.check(L.dta_weighted_ce(_lib.ptr(self.logits), _lib.ptr(y), _lib.ptr(self.loss_weight), x.shape[0],
                                     self.model._classes, _lib.ptr(self.loss), None, _lib.ptr(self.ce_scratch), st),
                   "dta_weighted_ce")
        return self.logits, self.loss


def predict(model, images, return_probs=True):
    """Inference step of the reference (`MultiStage.predict_step` / `TreeModel.predict_dataloader`): eval-mode forward,
    softmax over classes and the top-2 labels/scores, all on the device.  Returns (probs or None, top_idx [B,2] int64,
    top_score [B,2] float32)."""
    L = _lib.lib()
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            logits = model(images)
    finally:
        model.train(was_training)
    if isinstance(logits, (list, tuple)):
        logits = logits[-1]
    logits = logits.contiguous().float()
    Bn, classes = logits.shape
    probs = torch.empty_like(logits) if return_probs else None
    idx = torch.empty(Bn, 2, dtype=torch.int64, device=logits.device)
    score = torch.empty(Bn, 2, dtype=torch.float32, device=logits.device)
    _lib.check(L.dta_softmax_top2(_lib.ptr(logits), Bn, classes, _lib.ptr(probs), _lib.ptr(idx), _lib.ptr(score),
                                  _lib.current_stream_ptr()), "dta_softmax_top2")
    return probs, idx, score
