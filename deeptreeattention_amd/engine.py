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
from .dist import GradSync, kept_anywhere


class FusedTrainer:
    """Owns flat fp32 parameter / gradient / Adam-moment buffers; the model's Parameters become views of the
    flat parameter buffer (state_dict keys and shapes are unchanged).

    model        : deeptreeattention_amd.Hang2020.{Hang2020, spectral_network, spatial_network, vanilla_CNN}
    loss_weight  : (classes,) tensor or None (= ones, reference src/main.py:69)
    process_group: torch.distributed group (None = single process)
    """

    def __init__(self, model, lr, loss_weight=None, betas=(0.9, 0.999), eps=1e-8, process_group=None,
                 overlap_comm=True, keep_grads=False, last_head_only=False):
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
        self.single_score = model._net_code in (_lib.NET_HANG2020, _lib.NET_VANILLA)
        # a spectral/spatial network whose caller keeps only the last head's scores (year ensemble, year.py:30)
        self.last_head_only = bool(last_head_only) and not self.single_score
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
        self._desc_key = None
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
                if (self.hang or self.last_head_only) and ("classifier1" in n or "classifier2" in n):
                    continue   # only the last heads reach the loss (reference Hang2020.py:256-257, year.py:30)
                gt[n] = self._gview[id(tensors[n])]
            for Lv in (1, 2, 3):
                bn = H._get(mod, f"conv{Lv}.bn1")
                tensors[f"conv{Lv}.bn1.running_mean"] = bn.running_mean
                tensors[f"conv{Lv}.bn1.running_var"] = bn.running_var
                tensors[f"conv{Lv}.bn1.num_batches_tracked"] = bn.num_batches_tracked
            H._fill_struct(nets[i], kind, tensors, False)
            H._fill_struct(grads[i], kind, gt, True)
        return nets, grads

    def _describe(self, x):
        """Descriptor + parameter / gradient pointer structs for this input shape (no device allocation)."""
        B, bands, Hh, Ww = x.shape
        m = self.model
        key = (B, bands, Hh, Ww, m.precision, m.training)
        if key != self._desc_key:
            self.desc = _lib.NetDesc(B, bands, Hh, Ww, m._classes, m._net_code, _lib.dtype_code(m.precision),
                                     1 if m.training else 0, 4 if (self.single_score or self.last_head_only) else 7,
                                     H.BN_MOMENTUM, H.BN_EPS)
            self.nets, self.grads = self._structs()
            self._desc_key = key
        return key

    def _prepare(self, x):
        B = x.shape[0]
        m = self.model
        key = self._describe(x)
        if key != self._ws_key:
            L = _lib.lib()
            nbytes = L.dta_net_workspace_bytes(C.byref(self.desc))
            if nbytes == 0:
                raise RuntimeError("dta_net_workspace_bytes: " + L.dta_last_error().decode())
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self.logits = torch.empty(B, m._classes, dtype=torch.float32, device=self.device)
            self.dlogits = torch.empty_like(self.logits)
            self.ce_scratch = torch.empty(B + 1, dtype=torch.float32, device=self.device)
            self._ws_key = key

    def _forward_scores(self, x):
        """Enqueue the network forward; the (B, classes) scores the loss consumes land in self.logits."""
        L = _lib.lib()
        x = H._check_input(x)
        self._prepare(x)
        table = _lib.ScoreTable()
        joint = _lib.ptr(self.logits)
        if not self.single_score:
            if not self.last_head_only:
                raise RuntimeError("the fused step needs a single-score model (Hang2020, vanilla_CNN) or a "
                                   "spectral/spatial network built with last_head_only=True")
            table[0][2] = self.logits.data_ptr()
            joint = None
        _lib.check(L.dta_net_forward(C.byref(self.desc), self.nets, _lib.ptr(self.alpha) if self.hang else None,
                                     _lib.ptr(x), _lib.ptr(self._ws), C.byref(table), joint,
                                     _lib.current_stream_ptr()), "dta_net_forward")
        return self.logits

    def _loss(self, logits, y, want_grad):
        L = _lib.lib()
        _lib.check(L.dta_weighted_ce(_lib.ptr(logits), _lib.ptr(y), _lib.ptr(self.loss_weight), logits.shape[0],
                                     logits.shape[1], _lib.ptr(self.loss),
                                     _lib.ptr(self.dlogits) if want_grad else None, _lib.ptr(self.ce_scratch),
                                     _lib.current_stream_ptr()), "dta_weighted_ce")
        return self.loss

    def _backward(self, dlogits):
        """Enqueue the backward of the last forward from d(loss)/d(scores); gradients land in the flat buffer
        (summed over ranks when data-parallel)."""
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        d = C.byref(self.desc)
        alpha = _lib.ptr(self.alpha) if self.hang else None
        dalpha = _lib.ptr(self.alpha_g) if self.hang else None
        table = _lib.ScoreTable()
        djoint = _lib.ptr(dlogits)
        if not self.single_score:
            table[0][2] = dlogits.data_ptr()
            djoint = None
        self._zero_grads()             # C-ABI contract: gradient buffers (and dalpha) arrive zero-filled
        if self.world == 1:
            _lib.check(L.dta_net_backward(d, self.nets, alpha, _lib.ptr(self._ws), C.byref(table), djoint, self.grads,
                                          dalpha, 3, st), "dta_net_backward")
        else:
            # phase 1: everything but the first conv's weight gradient; its all-reduce (side stream when overlap is
            # on) runs while phase 2, the first conv's weight gradient, is computed
            _lib.check(L.dta_net_backward(d, self.nets, alpha, _lib.ptr(self._ws), C.byref(table), djoint, self.grads,
                                          dalpha, 1, st), "dta_net_backward")
            self.sync.reduce_early(self.flat_g[:self.split], self.alpha_g if self.hang else None)
            _lib.check(L.dta_net_backward(d, self.nets, alpha, _lib.ptr(self._ws), C.byref(table), djoint, self.grads,
                                          dalpha, 2, st), "dta_net_backward")
            self.sync.reduce_late(self.flat_g[self.split:])
            self.sync.finish()
        self._grads_clear = False

    def _zero_grads(self):
        if not self._grads_clear:
            self.flat_g.zero_()
            if self.hang:
                self.alpha_g.zero_()
            self._grads_clear = True

    def _adam(self):
        L = _lib.lib()
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
                        self.sync.grad_scale, _lib.current_stream_ptr()), "dta_adam_step")
        self._grads_clear = not self.keep_grads

    def _labels(self, y):
        if y.dtype != torch.int64 or not y.is_cuda:
            y = y.to(self.device, torch.int64)
        return y

    def train_step(self, x, y):
        """One optimisation step on the batch (x: float32 NCHW on the device, y: int64 labels).  Returns the loss
        as a 0-d device tensor (no host sync)."""
        if not self.single_score:
            raise RuntimeError("train_step supports Hang2020 and vanilla_CNN (single-score models); a year ensemble "
                               "of spectral networks trains through EnsembleTrainer")
        y = self._labels(y)
        logits = self._forward_scores(x)
        self._loss(logits, y, True)
        self._backward(self.dlogits)
        self._adam()
        return self.loss

    def training_step(self, batch, batch_idx=0):
        """The reference's TreeModel.training_step unpacking (src/main.py:71-80): batch = (individual, inputs, y),
        images = inputs["HSI"]; runs the whole optimisation step and returns the loss."""
        individual, inputs, y = batch
        return self.train_step(inputs["HSI"], y)

    def validation_step(self, batch, batch_idx=0):
        """TreeModel.validation_step (src/main.py:82-94): forward + weighted CE, no update."""
        individual, inputs, y = batch
        return self.forward_loss(inputs["HSI"], y)[1]

    def forward_loss(self, x, y):
        """Forward + loss only (validation_step, reference src/main.py:82-94); returns (logits, loss)."""
        logits = self._forward_scores(x)
        return logits, self._loss(logits, self._labels(y), False)


class EnsembleTrainer:
    """Fused train step of the year ensemble (reference src/models/year.py:9-33) as the reference's MultiStage loop
    drives one level of it (src/models/multi_stage.py:277-288 training_step, :258-275 one Adam per level):
    scores = mean over the kept years of each year's last-head scores, loss = weighted CE, backward, Adam.

    Every year's spectral_network owns flat parameter / gradient / moment buffers (a FusedTrainer restricted to the
    last head).  A year whose whole batch tensor sums to zero is skipped exactly as the reference skips it: no
    forward (BatchNorm running statistics and num_batches_tracked untouched), no gradient, and -- since torch's Adam
    passes over parameters whose grad is None -- no moment decay and no step-count advance for that year.  The
    zero-year test costs ONE host transfer per step (the reference: one blocking comparison per year); callers that
    know which years are present (the reference's dataset zero-fills missing years, src/data.py) pass `present`
    and the step enqueues without any host synchronisation.

    Data-parallel: a year is stepped when any rank kept it; ranks that skipped it contribute zero gradients to the
    same all-reduces (what DDP does for unused parameters)."""

    def __init__(self, model, lr, loss_weight=None, betas=(0.9, 0.999), eps=1e-8, process_group=None,
                 overlap_comm=True, keep_grads=False):
        from .year import learned_ensemble
        if not isinstance(model, learned_ensemble):
            raise TypeError("EnsembleTrainer needs a deeptreeattention_amd.year.learned_ensemble")
        self.model = model
        self.years = [FusedTrainer(m, lr, loss_weight, betas, eps, process_group, overlap_comm, keep_grads,
                                   last_head_only=True) for m in model.year_models]
        first = self.years[0]
        self.device, self.world, self.pg = first.device, first.world, first.pg
        self.loss_weight = first.loss_weight
        self.loss = torch.zeros((), dtype=torch.float32, device=self.device)
        self._shape = None
        self._ws = None
        self._ws_key = None

    @property
    def lr(self):
        return self.years[0].lr

    @lr.setter
    def lr(self, value):       # ReduceLROnPlateau-style schedulers set one rate per level
        for t in self.years:
            t.lr = float(value)

    def grad_of(self, param):
        for t in self.years:
            if id(param) in t._gview:
                return t._gview[id(param)]
        raise KeyError("not a parameter of this ensemble")

    def _kept(self, images, present):
        if len(images) != len(self.years):
            raise ValueError("expected one image tensor per year ({}), got {}".format(len(self.years), len(images)))
        if present is None:
            # reference year.py:27 (`x.sum() == 0`), all years in one host transfer
            present = (torch.stack([x.sum() for x in images]) != 0).tolist()
        local = [bool(k) for k in present]
        if not any(local):
            raise RuntimeError("every year of the batch is all-zero: the reference has nothing to average (year.py:33)")
        anywhere = local
        if self.world > 1:
            anywhere = kept_anywhere(local, self.pg, self.device)
        return local, anywhere

    def _buffers(self, B, classes):
        if self._shape != (B, classes):
            self.scores = torch.empty(B, classes, dtype=torch.float32, device=self.device)
            self.dscores = torch.empty_like(self.scores)
            self.ce_scratch = torch.empty(B + 1, dtype=torch.float32, device=self.device)
            self._shape = (B, classes)

    def _forward(self, images, local):
        """All kept years as the groups of one set of launches (dta_ensemble_forward); self.scores = their mean."""
        L = _lib.lib()
        kept = [i for i, k in enumerate(local) if k]
        if len(kept) > _lib.MAX_YEARS:
            raise RuntimeError("at most {} years per grouped launch".format(_lib.MAX_YEARS))
        xs = [H._check_input(images[i]) for i in kept]
        if any(x.shape != xs[0].shape for x in xs):
            raise ValueError("all years of a batch must have the same shape")
        keys = [self.years[i]._describe(x) for i, x in zip(kept, xs)]
        B, classes = xs[0].shape[0], self.model.year_models[0]._classes
        self._buffers(B, classes)
        n = len(kept)
        self._nets = (_lib.SubnetParams * n)(*[self.years[i].nets[0] for i in kept])
        self._grads = (_lib.SubnetGrads * n)(*[self.years[i].grads[0] for i in kept])
        self._xptr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        self._desc = self.years[kept[0]].desc
        ws_key = (n,) + keys[0]
        if ws_key != self._ws_key:
            nbytes = L.dta_ensemble_workspace_bytes(C.byref(self._desc), n)
            if nbytes == 0:
                raise RuntimeError("dta_ensemble_workspace_bytes: " + L.dta_last_error().decode())
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_key = ws_key
        _lib.check(L.dta_ensemble_forward(C.byref(self._desc), n, self._nets, self._xptr, _lib.ptr(self._ws),
                                          _lib.ptr(self.scores), _lib.current_stream_ptr()), "dta_ensemble_forward")
        self._live = xs     # inputs stay referenced until the step's launches are enqueued
        return kept

    def _backward(self, kept):
        L = _lib.lib()
        for i in kept:
            self.years[i]._zero_grads()          # C-ABI contract: gradient buffers arrive zero-filled
        _lib.check(L.dta_ensemble_backward(C.byref(self._desc), len(kept), self._nets, _lib.ptr(self._ws),
                                           _lib.ptr(self.dscores), self._grads, _lib.current_stream_ptr()),
                   "dta_ensemble_backward")
        for i in kept:
            self.years[i]._grads_clear = False

    def _ce(self, y, want_grad):
        L = _lib.lib()
        _lib.check(L.dta_weighted_ce(_lib.ptr(self.scores), _lib.ptr(y), _lib.ptr(self.loss_weight),
                                     self.scores.shape[0], self.scores.shape[1], _lib.ptr(self.loss),
                                     _lib.ptr(self.dscores) if want_grad else None, _lib.ptr(self.ce_scratch),
                                     _lib.current_stream_ptr()), "dta_weighted_ce")

    def train_step(self, images, y, present=None):
        """images: list of (B, bands, H, W) float32 device tensors, one per year; y: int64 labels.  Returns the loss
        as a 0-d device tensor."""
        local, anywhere = self._kept(images, present)
        y = self.years[0]._labels(y)
        kept = self._forward(images, local)
        self._ce(y, True)
        self.dscores.mul_(1.0 / len(kept))      # d(mean over kept years)/d(year score)
        self._backward(kept)
        # gradient exchange in YEAR ORDER on every rank (the ranks may have kept different years, and collectives pair
        # up by issue order): a year kept anywhere is reduced by all ranks, those that skipped it send zeros
        for i, t in enumerate(self.years):
            if not anywhere[i]:
                t._zero_grads()                 # skipped everywhere: grad None in the reference
            elif self.world > 1:
                if not local[i]:
                    t._zero_grads()
                t.sync.reduce_early(t.flat_g[:t.split])
                t.sync.reduce_late(t.flat_g[t.split:])
                t.sync.finish()
                t._grads_clear = False
        for i, t in enumerate(self.years):
            if anywhere[i]:
                t._adam()
        return self.loss

    def forward_loss(self, images, y, present=None):
        """validation_step of the level (multi_stage.py:290-304): ensemble scores + weighted CE, no update."""
        local, _ = self._kept(images, present)
        self._forward(images, local)
        self._ce(self.years[0]._labels(y), False)
        return self.scores, self.loss


class MultiStageTrainer:
    """Step driver of the reference's hierarchical model (src/models/multi_stage.py): one year ensemble, one class
    weight vector, one Adam and one learning rate per level, selected by Lightning's optimizer_idx /
    dataloader_idx.  Batches keep the reference's structure: (individual, {"HSI": [year tensors]}, labels)."""

    def __init__(self, models, lrs, loss_weights=None, **kwargs):
        loss_weights = loss_weights or [None] * len(models)
        self.levels = [EnsembleTrainer(m, lr, w, **kwargs) for m, lr, w in zip(models, lrs, loss_weights)]

    def training_step(self, batch, batch_idx, optimizer_idx, present=None):
        """multi_stage.py:277-288: the level's batch is batch[optimizer_idx]."""
        individual, inputs, y = batch[optimizer_idx]
        return self.levels[optimizer_idx].train_step(inputs["HSI"], y, present)

    def validation_step(self, batch, batch_idx, dataloader_idx, present=None):
        """multi_stage.py:290-304: returns the level's softmax scores with the loss."""
        individual, inputs, y = batch
        scores, loss = self.levels[dataloader_idx].forward_loss(inputs["HSI"], y, present)
        return {"individual": individual, "yhat": torch.softmax(scores, dim=1), "label": y, "val_loss": loss}

    def predict_step(self, batch, batch_idx=0, present=None):
        """multi_stage.py:306-318: every level's softmax scores for the same crops (eval-mode forward through a cached
        Predictor per level; the year ensembles' zero years are skipped as in training)."""
        individual, inputs = batch
        if not hasattr(self, "_predictors"):
            self._predictors = [Predictor(t.model) for t in self.levels]
        return individual, [pr(inputs["HSI"], True, present)[0].clone() for pr in self._predictors]


class MetadataTrainer:
    """Fused train step of the site-metadata fusion model (reference src/models/metadata.py): the step
    MetadataModel.training_step defines (:52-63: unweighted F.cross_entropy(model(images, site), y)) with Adam.
    The HSI branch (Hang2020, >99.9 % of the work) runs through the fused C-ABI pieces on flat buffers; the 16-wide
    site MLP and the 2*classes -> classes fusion layer (<0.2 MFLOP per sample) stay a small torch
    autograd graph with their own torch Adam, joined to the HSI branch at its (B, classes) scores."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, process_group=None, overlap_comm=True,
                 keep_grads=False):
        from .metadata import metadata_sensor_fusion
        if not isinstance(model, metadata_sensor_fusion):
            raise TypeError("MetadataTrainer needs a deeptreeattention_amd.metadata.metadata_sensor_fusion")
        self.model = model
        self.sensor = FusedTrainer(model.sensor_model, lr, None, betas, eps, process_group, overlap_comm, keep_grads)
        self.small = list(model.metadata_model.parameters()) + list(model.fc1.parameters())
        self.opt = torch.optim.Adam(self.small, lr=lr, betas=betas, eps=eps)
        self.world, self.pg = self.sensor.world, self.sensor.pg
        if self.world > 1:
            for t in self.small + list(model.metadata_model.buffers()):
                torch.distributed.broadcast(t.data, 0, group=self.pg)

    def _head(self, scores, site):
        meta = self.model.metadata_model(site)
        return torch.relu(self.model.fc1(torch.cat([meta, scores], dim=1)))

    def train_step(self, images, site, y):
        """images (B, bands, 11, 11) float32, site (B,) int64 site indices, y (B,) int64 labels -> loss (0-d tensor)."""
        y = self.sensor._labels(y)
        scores = self.sensor._forward_scores(images).detach().requires_grad_(True)
        self.opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(self._head(scores, site), y)
        loss.backward()                                   # the small graph: MLP / fusion grads and d(loss)/d(scores)
        self.sensor._backward(scores.grad.contiguous())
        self.sensor._adam()
        if self.world > 1:
            for p in self.small:
                torch.distributed.all_reduce(p.grad, group=self.pg)
                p.grad.mul_(1.0 / self.world)
        self.opt.step()
        return loss.detach()

    def training_step(self, batch, batch_idx=0):
        """metadata.py:52-63 unpacking: batch = (individual, {"HSI": images, "site": site}, y)."""
        individual, inputs, y = batch
        return self.train_step(inputs["HSI"], inputs["site"], y)

    def validation_step(self, batch, batch_idx=0):
        """metadata.py:65-83: forward + unweighted CE, no update."""
        individual, inputs, y = batch
        with torch.no_grad():
            scores = self.sensor._forward_scores(inputs["HSI"])
            return torch.nn.functional.cross_entropy(self._head(scores, inputs["site"]), self.sensor._labels(y))


class Predictor:
    """Inference step of the reference (`MultiStage.predict_step` src/models/multi_stage.py:306-318,
    `TreeModel.predict_dataloader` src/main.py:165-205): eval-mode forward, softmax over the classes and the top-2
    labels / scores, all on the device, with the descriptor, pointer tables, workspace and output buffers cached
    across calls (two C-ABI calls per batch, no per-call Python walk over the module tree).

    model: a network of this package (Hang2020, vanilla_CNN; spectral/spatial_network -> last head) or a
    year.learned_ensemble (zero years are skipped as in training: one host transfer per call, or pass `present`).
    The pointer tables are rebuilt when the batch shape changes; call refresh() after replacing parameter tensors
    (in-place updates, e.g. by FusedTrainer or load_state_dict, need nothing)."""

    def __init__(self, model):
        from .year import learned_ensemble
        self.model = model
        self.ensemble = isinstance(model, learned_ensemble)
        nets = list(model.year_models) if self.ensemble else [model]
        if not all(isinstance(n, H._Net) for n in nets):
            raise TypeError("Predictor needs a deeptreeattention_amd network or learned_ensemble")
        self.nets_mod = nets
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("Predictor needs the model on a ROCm device (model.cuda()); there is no CPU path")
        self._key = None

    def refresh(self):
        self._key = None

    def _tables(self, mods):
        out = []
        for m in mods:
            subnets = m._subnets()
            arr = (_lib.SubnetParams * len(subnets))()
            for i, (kind, mod, names) in enumerate(subnets):
                tensors = {n: H._get(mod, n) for n in names}
                for Lv in (1, 2, 3):
                    bn = H._get(mod, f"conv{Lv}.bn1")
                    tensors[f"conv{Lv}.bn1.running_mean"] = bn.running_mean
                    tensors[f"conv{Lv}.bn1.running_var"] = bn.running_var
                    tensors[f"conv{Lv}.bn1.num_batches_tracked"] = bn.num_batches_tracked
                H._fill_struct(arr[i], kind, tensors, False)
            out.append(arr)
        return out

    def _prepare(self, shape, kept):
        L = _lib.lib()
        m0 = self.nets_mod[0]
        key = (tuple(shape), m0.precision, tuple(kept))
        if key == self._key:
            return
        B, bands, Hh, Ww = shape
        single = m0._net_code in (_lib.NET_HANG2020, _lib.NET_VANILLA)
        self.desc = _lib.NetDesc(B, bands, Hh, Ww, m0._classes, m0._net_code, _lib.dtype_code(m0.precision), 0,
                                 4 | _lib.FORWARD_ONLY, H.BN_MOMENTUM, H.BN_EPS)
        tables = self._tables([self.nets_mod[i] for i in kept])
        if self.ensemble:
            self.nets = (_lib.SubnetParams * len(kept))(*[t[0] for t in tables])
            nbytes = L.dta_ensemble_workspace_bytes(C.byref(self.desc), len(kept))
        else:
            self.nets = tables[0]
            nbytes = L.dta_net_workspace_bytes(C.byref(self.desc))
        if nbytes == 0:
            raise RuntimeError("workspace: " + L.dta_last_error().decode())
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.logits = torch.empty(B, m0._classes, dtype=torch.float32, device=self.device)
        self.probs = torch.empty_like(self.logits)
        self.top_idx = torch.empty(B, 2, dtype=torch.int64, device=self.device)
        self.top_score = torch.empty(B, 2, dtype=torch.float32, device=self.device)
        self.single = single
        self._key = key

    def logits_of(self, images, present=None):
        """Eval-mode scores (B, classes) of the batch; the returned tensor is reused by the next call."""
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        if self.ensemble:
            if present is None:
                present = (torch.stack([x.sum() for x in images]) != 0).tolist()
            kept = [i for i, k in enumerate(present) if k]
            if not kept:
                raise RuntimeError("every year of the batch is all-zero: nothing to average (reference year.py:33)")
            xs = [H._check_input(images[i]) for i in kept]
            self._prepare(xs[0].shape, kept)
            xptr = (C.c_void_p * len(kept))(*[x.data_ptr() for x in xs])
            _lib.check(L.dta_ensemble_forward(C.byref(self.desc), len(kept), self.nets, xptr, _lib.ptr(self.ws),
                                              _lib.ptr(self.logits), st), "dta_ensemble_forward")
            return self.logits
        x = H._check_input(images)
        self._prepare(x.shape, [0])
        m = self.nets_mod[0]
        table = _lib.ScoreTable()
        joint = _lib.ptr(self.logits)
        if not self.single:
            table[0][2] = self.logits.data_ptr()
            joint = None
        alpha = _lib.ptr(m.alpha) if m._net_code == _lib.NET_HANG2020 else None
        _lib.check(L.dta_net_forward(C.byref(self.desc), self.nets, alpha, _lib.ptr(x), _lib.ptr(self.ws),
                                     C.byref(table), joint, st), "dta_net_forward")
        return self.logits

    def __call__(self, images, return_probs=True, present=None):
        """Returns (probs or None, top_idx [B,2] int64, top_score [B,2] float32); buffers are reused across calls."""
        L = _lib.lib()
        logits = self.logits_of(images, present)
        Bn, classes = logits.shape
        _lib.check(L.dta_softmax_top2(_lib.ptr(logits), Bn, classes, _lib.ptr(self.probs) if return_probs else None,
                                      _lib.ptr(self.top_idx), _lib.ptr(self.top_score), _lib.current_stream_ptr()),
                   "dta_softmax_top2")
        return (self.probs if return_probs else None), self.top_idx, self.top_score


def predict(model, images, return_probs=True):
    """Inference step of the reference (`MultiStage.predict_step` / `TreeModel.predict_dataloader`): eval-mode forward
    (whatever the module's current train/eval flag), softmax over classes and the top-2 labels/scores, all on the
    device, through a Predictor cached on the module.  Returns (probs or None, top_idx [B,2] int64, top_score [B,2]
    float32); the tensors are reused by the next call on the same module."""
    pr = model.__dict__.get("_dta_predictor")
    if pr is None:
        pr = Predictor(model)
        model.__dict__["_dta_predictor"] = pr
    return pr(images, return_probs)
