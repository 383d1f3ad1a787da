"""Fused train step for the Hang2020 hot path: forward + class-weighted cross-entropy + backward + Adam as four
C-ABI calls on flat device buffers, optionally data-parallel (one process per GPU, RCCL all-reduce of the flat
gradient asynchronously, overlapped with the first conv's weight gradient).

Replaces, for one model, the reference's TreeModel.training_step + loss.backward() + Adam.step()
(/root/reference/src/main.py:71-80, :135-137) and Lightning's DDP wrapper (train.py:89-98, `gpus>1`).
BatchNorm statistics stay per-rank (the reference does not enable sync_batchnorm); the loss each rank
back-propagates is its own weighted mean (DDP semantics: gradients are averaged over ranks).
"""
import ctypes as C
import os

import torch

from . import _lib
from . import Hang2020 as H
from .dist import GradSync, PeerExchange, RcclDirect, choose_exchange


def _round4(n):
    return (int(n) + 3) & ~3


def _aligned_offsets(params):
    """Element offsets of `params` laid out back to back with every start on a multiple of 4 elements (16 bytes);
    returns (offsets, total rounded up to 4)."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off = _round4(off + p.numel())
    return offs, off


class FusedTrainer:
    """Owns flat fp32 parameter / gradient / Adam-moment buffers; the model's Parameters become views of the
    flat parameter buffer (state_dict keys and shapes are unchanged).

    model        : deeptreeattention_amd.Hang2020.{Hang2020, spectral_network, spatial_network, vanilla_CNN}
    loss_weight  : (classes,) tensor or None (= ones, reference src/main.py:69)
    process_group: torch.distributed group (None = single process)
    """

    _warned_force = False

    def __init__(self, model, lr, loss_weight=None, betas=(0.9, 0.999), eps=1e-8, process_group=None,
                 overlap_comm=True, keep_grads=False, last_head_only=False, three_head_loss=False,
                 extra_grad_slots=0, storage=None, exchange=None, exchange_opts=None, broadcast_buffers=False,
                 _dev_skip_allreduce=False):
        """three_head_loss: train on the SUM of the class-weighted cross-entropies of every classifier head (the Hang
        et al. recipe BASELINE.json's north_star words as "three-head weighted cross-entropy"): three heads for a
        spectral / spatial network, all six for Hang2020 (whose sigmoid(alpha) blend is then not on the graph, so
        alpha is left alone, as torch leaves a parameter whose grad is None).  Default False = the reference's own
        step, which keeps the last head only (Hang2020.py:256-257, src/main.py:78).
        extra_grad_slots: fp32 slots appended to the first gradient bucket for a caller's own small gradients
        (MetadataTrainer: the site MLP / fusion layer travel in the same collective).
        storage: (p, g, m, v) flat fp32 tensors to carve the buffers from (EnsembleTrainer: one buffer for all years).
        exchange: how gradients cross the ranks when data-parallel -- "peer": the sum over ranks and Adam are ONE launch
        on the compute stream through IPC-mapped peer memory (csrc/xchg.hip; ranks = processes of one node); "rccl":
        one ncclAllReduce of the flat buffer enqueued on the compute stream (librccl called directly); "torch":
        torch.distributed all-reduces (two buckets overlapped with the first conv's weight gradient when
        overlap_comm); None: peer if the crash-isolated probe passes on every rank, else rccl / torch.
        exchange_opts: keyword arguments of dist.PeerExchange (timeout_s, max_workgroups).
        broadcast_buffers: Lightning's DDP wrapper (torch DDP default broadcast_buffers=True) copies rank 0's BatchNorm
        running statistics / counters to every rank before each forward, so all ranks' buffers -- and a checkpoint
        written by any rank -- equal rank 0's view.  They never enter a training-mode forward or a gradient, so the
        default here is False (per-rank statistics, one collective less per step); True reproduces the reference's
        buffers exactly at the cost of one small broadcast per step (data-parallel only)."""
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
        # DTA_FORCE_COLLECTIVES=1: a one-rank process group still goes through the collectives (the only way to run the
        # RCCL code path -- streams, buckets, the alpha slot -- on a single-GPU box; tests/test_rccl_single_gpu.py)
        self.comm = self.world > 1 or (os.environ.get("DTA_FORCE_COLLECTIVES") == "1" and torch.distributed.is_available()
                                       and torch.distributed.is_initialized())
        if self.comm and self.world == 1 and not FusedTrainer._warned_force:
            import warnings
            warnings.warn("DTA_FORCE_COLLECTIVES=1: a one-rank process group still issues the gradient exchange "
                          "(development measurement switch)")
            FusedTrainer._warned_force = True
        self.exchange = None           # "peer" / "rccl" / "torch" when this trainer exchanges gradients itself
        if self.comm and storage is None:
            self.exchange = choose_exchange(exchange, process_group) if self.world > 1 else (exchange or "torch")
        # overlapped exchange: "peer": the head bucket's sum rides in the first conv's weight-gradient launch; "rccl" / "torch":
        # two buckets, the first all-reduced on a side stream while the first conv's weight gradient is computed
        # (a one-rank group under DTA_FORCE_COLLECTIVES takes the overlapped launch order too: that is what its cost measurement is for)
        # "rccl": the side-stream two-bucket form (north_star's literal design) only on request, exchange_opts={"side_stream":
        # True}: measured with one rank (tools/dp_one_rank.py, profiles/r05_dp_one_rank.txt) its two event forks and the join cost
        # +35 us per 0.52 ms step before a byte moves, the single collective on the compute stream about nothing -- more than the
        # ~25 us of wire time the overlap can hide at 8 ranks
        opts = dict(exchange_opts or {})
        side_stream = bool(opts.pop("side_stream", False))
        exchange_opts = opts or None
        self.overlap = bool(overlap_comm and self.comm and (self.exchange in (None, "torch", "peer") or (self.exchange == "rccl" and side_stream)))
        self.hang = model._net_code == _lib.NET_HANG2020
        self.single_score = model._net_code in (_lib.NET_HANG2020, _lib.NET_VANILLA)
        self.three_head = bool(three_head_loss)
        if self.three_head and model._net_code == _lib.NET_VANILLA:
            raise ValueError("vanilla_CNN has a single head")
        # a spectral/spatial network whose caller keeps only the last head's scores (year ensemble, year.py:30)
        self.last_head_only = bool(last_head_only) and not self.single_score and not self.three_head
        self.alpha_on_graph = self.hang and not self.three_head
        plist = model._param_list()
        dev = plist[-1].device
        if dev.type != "cuda":
            raise RuntimeError("FusedTrainer needs the model on a ROCm device (model.cuda()); there is no CPU path")
        self.device = dev
        fp32 = [p for p in plist if p.dtype == torch.float32]
        # flat layout: [everything except the first conv's weights | caller's extra slots | alpha's exchange slot |
        # first conv's weights]: the gradient all-reduce of the first part overlaps with the first conv's
        # weight-gradient kernel (the last to finish); alpha's float64 gradient crosses the ranks in its fp32 slot
        first = [p for kind, mod, names in model._subnets() for p in [H._get(mod, "conv1.conv_layer.weight")]]
        first_ids = {id(p) for p in first}
        rest = [p for p in fp32 if id(p) not in first_ids]
        # every tensor starts on a 16-byte boundary of the flat buffers (the small GEMMs and the Adam pass use 16-byte
        # accesses when they can: a classifier weight that starts 4, 8 or 12 bytes off falls back to scalar loads)
        rest_off, n_rest = _aligned_offsets(rest)
        first_off, n_first = _aligned_offsets(first)
        self.extra_off, self.extra_n = n_rest, int(extra_grad_slots)
        self.split = _round4(n_rest + self.extra_n + (1 if self.hang else 0))
        self.alpha_slot_off = n_rest + self.extra_n
        n = self.split + n_first
        self.n, self.n_first = n, n_first
        self.ex = None
        if storage is None:
            self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
            if self.exchange == "peer":
                # the gradient buffer is the exchange's: library-owned device memory every peer has mapped; with
                # overlap_comm its head segment [everything but the first conv's weights] is summed over the ranks by side
                # workgroups of the first conv's weight-gradient launch (dta_net_backward_xchg)
                self.ex = PeerExchange(n, process_group, split=(self.split if self.overlap else 0),
                                       **(exchange_opts or {}))
                assert self.ex.capacity == n
                self.flat_g = self.ex.grad
            else:
                self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
            self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
            self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
            head = tuple(t[:self.split] for t in (self.flat_p, self.flat_g, self.flat_m, self.flat_v))
            tail = tuple(t[self.split:] for t in (self.flat_p, self.flat_g, self.flat_m, self.flat_v))
        else:
            # two separately placed segments (EnsembleTrainer: all years' first buckets are adjacent, so are their
            # second buckets): (p, g, m, v) of the head [rest | extra | alpha slot] and of the tail [first-conv weights]
            head, tail = storage
            assert all(t.numel() == self.split and t.dtype == torch.float32 and t.is_contiguous() for t in head)
            assert all(t.numel() == n_first and t.dtype == torch.float32 and t.is_contiguous() for t in tail)
            self.flat_p = self.flat_g = self.flat_m = self.flat_v = None
        self.p_head, self.g_head, self.m_head, self.v_head = head
        self.p_tail, self.g_tail, self.m_tail, self.v_tail = tail
        self._grads_clear = True       # gradient buffers / alpha_g hold zeros (kept so by dta_adam_step_zero_grad)
        self._gview = {}
        with torch.no_grad():
            for group, offs, pbuf, gbuf in ((rest, rest_off, self.p_head, self.g_head),
                                            (first, first_off, self.p_tail, self.g_tail)):
                for p, off in zip(group, offs):
                    k = p.numel()
                    pbuf[off:off + k].copy_(p.reshape(-1))
                    p.data = pbuf[off:off + k].view(p.shape)
                    self._gview[id(p)] = gbuf[off:off + k].view(p.shape)
        self.extra_g = self.g_head[self.extra_off:self.extra_off + self.extra_n]
        if self.hang:
            self.alpha = model.alpha
            self.alpha_g = torch.zeros((), dtype=torch.float64, device=dev)
            self.alpha_m = torch.zeros((), dtype=torch.float64, device=dev)
            self.alpha_v = torch.zeros((), dtype=torch.float64, device=dev)
            self.alpha_slot = self.g_head[self.alpha_slot_off:self.alpha_slot_off + 1]
        self.external_loss = False     # True: the caller computes its own loss from the scores (MetadataTrainer)
        self.fused_loss = False
        self.loss_weight = None if loss_weight is None else loss_weight.to(dev, torch.float32).contiguous()
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._ws = None
        self._ws_key = None
        self._desc_key = None
        self._desc_cache, self._ws_cache = {}, {}      # the last two (shape, mode) keys: descriptors / device buffers
        self.broadcast_buffers = bool(broadcast_buffers)
        self._reduced = False          # peer exchange: the gradient buffer already holds the sum (reduce_now)
        self.sync = GradSync(self.world, self.pg, rccl=RcclDirect(self.pg) if self.exchange == "rccl" else None,
                             skip_allreduce=_dev_skip_allreduce)
        if self.comm and storage is None:
            self.broadcast_parameters()

    @staticmethod
    def bucket_sizes(model, extra_grad_slots=0):
        """(head, tail) element counts of the two gradient buckets a FusedTrainer for `model` uses (see `storage`)."""
        first = [H._get(mod, "conv1.conv_layer.weight") for kind, mod, names in model._subnets()]
        ids = {id(p) for p in first}
        rest = [p for p in model._param_list() if p.dtype == torch.float32 and id(p) not in ids]
        n_rest, n_first = _aligned_offsets(rest)[1], _aligned_offsets(first)[1]
        return _round4(n_rest + int(extra_grad_slots) + (1 if model._net_code == _lib.NET_HANG2020 else 0)), n_first

    # ------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        """DDP start-up semantics: every rank starts from rank `src`'s parameters and buffers."""
        if self.world > 1:
            self.sync.broadcast([self.p_head, self.p_tail] + ([self.alpha.data] if self.hang else []) + list(self.model.buffers()), src)

    def sync_buffers(self, src=0):
        """Every rank's BatchNorm running statistics / counters := rank `src`'s (one packed float64 broadcast: int64
        counters and float32 statistics are exact in it)."""
        if self.world <= 1:
            return
        bufs = list(self.model.buffers())
        flat = torch.cat([b.detach().reshape(-1).to(torch.float64) for b in bufs])
        torch.distributed.broadcast(flat, src, group=self.pg)
        off = 0
        with torch.no_grad():
            for b in bufs:
                n = b.numel()
                b.copy_(flat[off:off + n].view(b.shape).to(b.dtype))
                off += n

    def reduce_now(self):
        """Peer exchange only: sum the gradient buffer over the ranks NOW (one launch) instead of inside the optimizer
        launch -- for callers that read summed gradients before the step (MetadataTrainer's spare slots)."""
        al = self.alpha_on_graph
        self.ex.allreduce(self.alpha_g if al else None, self.alpha_slot_off)
        self._reduced = True

    def check_exchange(self):
        """Raise if a peer-exchange launch timed out waiting for a rank (a pinned host word: no synchronisation; train_step
        calls it first thing, so a failed exchange is fatal within the steps already enqueued -- the launches after the
        failure return without applying anything)."""
        if self.ex is not None:
            self.ex.check()

    def close(self):
        """Collective: release the peer exchange / RCCL communicator (every rank calls it)."""
        if self.ex is not None:
            self.flat_g = self.g_head = self.g_tail = None
            self._gview = {}
            self.ex.close()
            self.ex = None
        if self.sync.rccl is not None:
            self.sync.rccl.close()
            self.sync.rccl = None

    # ---- optimizer state in torch.optim.Adam's layout (Lightning checkpoints: "optimizer_states") ----
    def _moment_views(self, p):
        """(exp_avg, exp_avg_sq) of a parameter of the model as views of the flat moment buffers."""
        if self.hang and p is self.alpha:
            return self.alpha_m, self.alpha_v
        for pbuf, mbuf, vbuf in ((self.p_head, self.m_head, self.v_head), (self.p_tail, self.m_tail, self.v_tail)):
            off = (p.data_ptr() - pbuf.data_ptr()) // 4
            if 0 <= off and off + p.numel() <= pbuf.numel() and p.data_ptr() >= pbuf.data_ptr():
                return mbuf[off:off + p.numel()].view(p.shape), vbuf[off:off + p.numel()].view(p.shape)
        raise KeyError("not a parameter of this trainer")

    def optimizer_state_dict(self):
        """state_dict() of the torch.optim.Adam(self.model.parameters(), lr) the reference builds (src/main.py:135-137) after
        the same steps: loads into that optimizer, and into optim.DtaAdam."""
        params = list(self.model.parameters())
        state = {}
        for i, p in enumerate(params):
            m, v = self._moment_views(p)
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(params)))}
        return {"state": state if self.step_count > 0 else {}, "param_groups": [group]}

    @torch.no_grad()
    def load_optimizer_state_dict(self, sd):
        """Resume from a torch.optim.Adam / optim.DtaAdam / FusedTrainer optimizer state dict over self.model.parameters()."""
        params = list(self.model.parameters())
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(params):
            raise ValueError("expected one parameter group over the model's {} parameters".format(len(params)))
        g = groups[0]
        if g.get("weight_decay") or g.get("amsgrad") or g.get("maximize"):
            raise ValueError("the fused step is Adam without weight decay / amsgrad / maximize (the reference's setting)")
        steps = set()
        for key, p in zip(g["params"], params):
            st = sd["state"].get(key)
            m, v = self._moment_views(p)
            if st is None:                    # never stepped by torch (grad None every time): zero moments
                m.zero_(); v.zero_()
                continue
            m.copy_(st["exp_avg"].to(m)); v.copy_(st["exp_avg_sq"].to(v))
            if float(st["step"]) > 0:
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("parameters carry different step counts: {}".format(sorted(steps)))
        self.step_count = steps.pop() if steps else 0
        self.lr, self.betas, self.eps = float(g["lr"]), tuple(g["betas"]), float(g["eps"])

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
                if ((self.hang and not self.three_head) or self.last_head_only) and ("classifier1" in n or "classifier2" in n):
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
        """Descriptor + parameter / gradient pointer structs for this input shape (no device allocation).  The last two
        shapes stay cached: an epoch's ragged last batch does not rebuild the full batch's tables (or workspace) twice."""
        B, bands, Hh, Ww = x.shape
        m = self.model
        key = (B, bands, Hh, Ww, m.precision, m.training)
        if key != self._desc_key:
            hit = self._desc_cache.get(key)
            if hit is None:
                all_heads = self.three_head or not (self.single_score or self.last_head_only)
                # single-score step: the blend (Hang2020) / the scores go straight into the fused loss launch (dta_net_loss)
                fused_loss = self.single_score and not self.three_head and not self.external_loss
                mask = (7 if all_heads else 4) | (_lib.SKIP_BLEND if (fused_loss and self.hang) else 0)
                desc = _lib.NetDesc(B, bands, Hh, Ww, m._classes, m._net_code, _lib.dtype_code(m.precision),
                                    1 if m.training else 0, mask, H.BN_MOMENTUM, H.BN_EPS)
                nets, grads = self._structs()
                hit = (desc, nets, grads, fused_loss, self.external_loss)
                if len(self._desc_cache) >= 2:
                    self._desc_cache.pop(next(iter(self._desc_cache)))
                self._desc_cache[key] = hit
            if hit[4] != self.external_loss:          # (MetadataTrainer flips the flag right after construction)
                self._desc_cache.pop(key)
                self._desc_key = None
                return self._describe(x)
            self.desc, self.nets, self.grads, self.fused_loss = hit[:4]
            self._desc_key = key
        return key

    def _prepare(self, x):
        B = x.shape[0]
        m = self.model
        key = self._describe(x)
        if key != self._ws_key:
            bufs = self._ws_cache.get(key)
            if bufs is None:
                L = _lib.lib()
                nbytes = L.dta_net_workspace_bytes(C.byref(self.desc))
                if nbytes == 0:
                    raise RuntimeError("dta_net_workspace_bytes: " + L.dta_last_error().decode())
                bufs = {"_ws": torch.empty(nbytes, dtype=torch.uint8, device=self.device),
                        "logits": torch.empty(B, m._classes, dtype=torch.float32, device=self.device),
                        "ce_scratch": torch.zeros(B + 2, dtype=torch.float32, device=self.device)}     # (last word: dta_net_loss's block counter)
                bufs["dlogits"] = torch.empty_like(bufs["logits"])
                if self.three_head:      # per-head scores / score gradients / losses: [net][head]
                    nn_ = 2 if self.hang else 1
                    bufs["head_scores"] = torch.empty(nn_, 3, B, m._classes, dtype=torch.float32, device=self.device)
                    bufs["head_dscores"] = torch.empty_like(bufs["head_scores"])
                    bufs["head_losses"] = torch.zeros(nn_ * 3, dtype=torch.float32, device=self.device)
                if len(self._ws_cache) >= 2:
                    self._ws_cache.pop(next(iter(self._ws_cache)))
                self._ws_cache[key] = bufs
            for k, v in bufs.items():
                setattr(self, k, v)
            self._ws_key = key

    def _forward_scores(self, x):
        """Enqueue the network forward; the (B, classes) scores the loss consumes land in self.logits (three-head
        mode: every head's scores in self.head_scores[net][head]; self.logits holds Hang2020's blended scores or the
        last head)."""
        L = _lib.lib()
        tiles = getattr(x, "tiles", None)        # preprocess.PatchTiles: the input already as the first conv's bf16 tiles
        if tiles is not None:
            if self.model.precision != "bf16":
                raise RuntimeError("PatchTiles inputs need a bf16-mode network")
        else:
            x = H._check_input(x)
        self._tiles = tiles                      # the backward reads them again (first conv's weight gradient)
        self._prepare(x)
        table = _lib.ScoreTable()
        joint = _lib.ptr(self.logits)
        if self.three_head:
            for g in range(self.head_scores.shape[0]):
                for hd in range(3):
                    table[g][hd] = self.head_scores[g, hd].data_ptr()
            if not self.hang:
                joint = None
        elif not self.single_score:
            if not self.last_head_only:
                raise RuntimeError("the fused step needs a single-score model (Hang2020, vanilla_CNN), a spectral/"
                                   "spatial network built with last_head_only=True, or three_head_loss=True")
            table[0][2] = self.logits.data_ptr()
            joint = None
        if self.fused_loss and self.hang:
            joint = None                         # DTA_SKIP_BLEND: the blend happens inside the loss launch
        fwd = L.dta_net_forward if tiles is None else L.dta_net_forward_tiles
        _lib.check(fwd(C.byref(self.desc), self.nets, _lib.ptr(self.alpha) if self.hang else None,
                       _lib.ptr(x if tiles is None else tiles), _lib.ptr(self._ws), C.byref(table), joint,
                       _lib.current_stream_ptr()), "dta_net_forward")
        if self.three_head and not self.hang:
            self.logits = self.head_scores[0, 2]
        return self.logits

    def _forward_and_loss(self, x, y, want_grad):
        """Forward + class-weighted cross-entropy of a single-score network in ONE C-ABI call (dta_net_forward_loss: the
        reference's `y_hat = model.forward(images); loss = F.cross_entropy(y_hat, y, weight)`, src/main.py:77-78).  For a
        training-mode Hang2020 on 11x11 patches the third stage of both branches, the last heads, the blend and the loss
        are one launch.  Scores land in self.logits, the loss in self.loss, d(loss)/d(scores) in self.dlogits."""
        L = _lib.lib()
        tiles = getattr(x, "tiles", None)
        if tiles is not None:
            if self.model.precision != "bf16":
                raise RuntimeError("PatchTiles inputs need a bf16-mode network")
        else:
            x = H._check_input(x)
        self._tiles = tiles
        self._prepare(x)
        self.loss = torch.empty((), dtype=torch.float32, device=self.device)      # fresh per step (see _loss)
        _lib.check(L.dta_net_forward_loss(C.byref(self.desc), self.nets, _lib.ptr(self.alpha) if self.hang else None,
                                          None if tiles is not None else _lib.ptr(x), _lib.ptr(tiles), _lib.ptr(self._ws),
                                          _lib.ptr(y), _lib.ptr(self.loss_weight), _lib.ptr(self.logits), _lib.ptr(self.loss),
                                          _lib.ptr(self.dlogits) if want_grad else None, _lib.ptr(self.ce_scratch),
                                          _lib.current_stream_ptr()), "dta_net_forward_loss")
        return self.logits

    def _loss_heads(self, y, want_grad):
        """Sum over all heads of the class-weighted cross-entropy; d(loss)/d(head scores) in self.head_dscores."""
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        hs = self.head_scores
        k = 0
        for g in range(hs.shape[0]):
            for hd in range(3):
                _lib.check(L.dta_weighted_ce(_lib.ptr(hs[g, hd]), _lib.ptr(y), _lib.ptr(self.loss_weight), hs.shape[2],
                                             hs.shape[3], C.c_void_p(self.head_losses.data_ptr() + 4 * k),
                                             _lib.ptr(self.head_dscores[g, hd]) if want_grad else None,
                                             _lib.ptr(self.ce_scratch), st), "dta_weighted_ce")
                k += 1
        self.loss = torch.sum(self.head_losses, dim=0)
        return self.loss

    def _loss(self, logits, y, want_grad):
        L = _lib.lib()
        # a fresh 0-d tensor per call (caching-allocator bookkeeping only, no device work): callers may keep every
        # step's loss (Lightning collects them per epoch) without an extra copy kernel on the stream
        self.loss = torch.empty((), dtype=torch.float32, device=self.device)
        if self.fused_loss:
            # blend (Hang2020) + cross-entropy + loss in one launch; self.logits receives / holds the scores
            _lib.check(L.dta_net_loss(C.byref(self.desc), _lib.ptr(self.alpha) if self.hang else None, _lib.ptr(self._ws),
                                      _lib.ptr(y), _lib.ptr(self.loss_weight), _lib.ptr(self.logits), _lib.ptr(self.loss),
                                      _lib.ptr(self.dlogits) if want_grad else None, _lib.ptr(self.ce_scratch),
                                      _lib.current_stream_ptr()), "dta_net_loss")
            return self.loss
        _lib.check(L.dta_weighted_ce(_lib.ptr(logits), _lib.ptr(y), _lib.ptr(self.loss_weight), logits.shape[0],
                                     logits.shape[1], _lib.ptr(self.loss),
                                     _lib.ptr(self.dlogits) if want_grad else None, _lib.ptr(self.ce_scratch),
                                     _lib.current_stream_ptr()), "dta_weighted_ce")
        return self.loss

    def _backward(self, dlogits):
        """Enqueue the backward of the last forward from d(loss)/d(scores); gradients land in the flat buffer
        (summed over ranks when data-parallel).  Three-head mode: dlogits is ignored, self.head_dscores is used."""
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        d = C.byref(self.desc)
        alpha = _lib.ptr(self.alpha) if self.hang else None
        dalpha = _lib.ptr(self.alpha_g) if self.alpha_on_graph else None
        table = _lib.ScoreTable()
        djoint = _lib.ptr(dlogits)
        if self.three_head:
            for g in range(self.head_dscores.shape[0]):
                for hd in range(3):
                    table[g][hd] = self.head_dscores[g, hd].data_ptr()
            djoint = None
        elif not self.single_score:
            table[0][2] = dlogits.data_ptr()
            djoint = None
        self._zero_grads()             # C-ABI contract: gradient buffers (and dalpha) arrive zero-filled

        # data-parallel: alpha's finished float64 gradient is also rounded (once) into its fp32 slot of the first bucket by the backward's last launch
        # (peer exchange: the exchange launch itself converts the float64 d(alpha) into the slot)
        slot = _lib.ptr(self.alpha_slot) if (self.comm and self.alpha_on_graph and self.ex is None) else None

        def run(phases):
            tiles = None if self._tiles is None else _lib.ptr(self._tiles)
            _lib.check(L.dta_net_backward_dp(d, self.nets, alpha, tiles, _lib.ptr(self._ws), C.byref(table), djoint,
                                             self.grads, dalpha, slot, phases, st), "dta_net_backward_dp")
        ag, slot_t = None, None        # (GradSync's copy-in / copy-out of alpha is for callers without the slot kernels)
        if self.ex is not None and self.ex.split:
            # peer exchange, overlapped: the head bucket's sum over the ranks rides in the first conv's weight-gradient
            # launch (side workgroups); the optimizer launch that follows sums only the tail
            tiles = None if self._tiles is None else _lib.ptr(self._tiles)
            _lib.check(L.dta_net_backward_xchg(d, self.nets, alpha, tiles, _lib.ptr(self._ws), C.byref(table), djoint,
                                               self.grads, dalpha, self.ex._h, self.alpha_slot_off if self.alpha_on_graph else -1,
                                               st), "dta_net_backward_xchg")
        elif not self.comm or self.ex is not None:
            run(3)                     # peer exchange: the sum over ranks happens inside the optimizer launch
        elif self.overlap:
            # phase 1: everything but the first conv's weight gradient; its all-reduce (backend stream) runs while
            # phase 2, the first conv's weight gradient, is computed: two collectives per step
            run(1)
            self.sync.reduce_early(self.g_head, ag, slot_t)
            run(2)
            self.sync.reduce_late(self.g_tail)
            self.sync.finish()
        elif self.flat_g is not None:
            run(3)                     # no overlap: ONE collective over the whole flat gradient
            self.sync.reduce_all(self.flat_g, ag, slot_t)
            self.sync.finish()
        else:
            run(3)
            self.sync.reduce_early(self.g_head, ag, slot_t)
            self.sync.reduce_late(self.g_tail)
            self.sync.finish()
        if self.comm and self.alpha_on_graph and self.keep_grads and self.ex is None:
            self.alpha_g.copy_(self.alpha_slot[0])      # readable summed gradient (grad_of); the optimizer reads the slot
        self._grads_clear = False

    def _zero_grads(self):
        if not self._grads_clear:
            if self.flat_g is not None:
                self.flat_g.zero_()
            else:
                self.g_head.zero_()
                self.g_tail.zero_()
            if self.hang:
                self.alpha_g.zero_()
            self._grads_clear = True

    def _adam(self):
        L = _lib.lib()
        self.step_count += 1
        # default: step + zero_grad in one pass, so the next backward finds its gradient buffers already cleared
        al = self.alpha_on_graph
        if self.ex is not None and not self._reduced:
            # sum over ranks + Adam (+ zero_grad) in one launch; with keep_grads the buffer holds the summed gradient after
            self.ex.adam_step(self.flat_p, self.flat_m, self.flat_v, self.alpha if al else None,
                              self.alpha_g if al else None, self.alpha_slot_off, self.alpha_m if al else None,
                              self.alpha_v if al else None, self.step_count, self.lr, self.betas, self.eps,
                              zero_grad=not self.keep_grads)
            self._grads_clear = not self.keep_grads
            return
        self._reduced = False
        slot = _lib.ptr(self.alpha_slot) if (self.comm and al) else None
        if self.flat_p is not None:
            segs = [(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.n)]
        elif all(h.data_ptr() + 4 * self.split == t.data_ptr() for h, t in
                 ((self.p_head, self.p_tail), (self.g_head, self.g_tail), (self.m_head, self.m_tail),
                  (self.v_head, self.v_tail))):
            segs = [(self.p_head, self.g_head, self.m_head, self.v_head, self.n)]   # adjacent segments: one pass
        else:
            segs = [(self.p_head, self.g_head, self.m_head, self.v_head, self.split),
                    (self.p_tail, self.g_tail, self.m_tail, self.v_tail, self.n_first)]
        for i, (p, g, m, v, n) in enumerate(segs):
            a = al and i == 0
            _lib.check(L.dta_adam_step_dp(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), n,
                                          _lib.ptr(self.alpha) if a else None, _lib.ptr(self.alpha_g) if a else None,
                                          slot if a else None,
                                          _lib.ptr(self.alpha_m) if a else None, _lib.ptr(self.alpha_v) if a else None,
                                          self.step_count, self.lr, self.betas[0], self.betas[1], self.eps,
                                          self.sync.grad_scale, 0 if self.keep_grads else 1, _lib.current_stream_ptr()),
                       "dta_adam_step_dp")
        self._grads_clear = not self.keep_grads

    def _labels(self, y):
        if y.dtype != torch.int64 or not y.is_cuda:
            y = y.to(self.device, torch.int64)
        return y

    def train_step(self, x, y):
        """One optimisation step on the batch (x: float32 NCHW on the device, y: int64 labels).  Returns the loss
        as a fresh 0-d device tensor (no host sync; safe to collect across steps, as Lightning does)."""
        if not (self.single_score or self.three_head or self.last_head_only):
            raise RuntimeError("train_step supports Hang2020 / vanilla_CNN (single score), any network with "
                               "three_head_loss=True, or a spectral/spatial network with last_head_only=True; a year "
                               "ensemble of spectral networks trains through EnsembleTrainer")
        if self.ex is not None:
            self.ex.check()                # a timed-out exchange is fatal (its later launches apply nothing)
        y = self._labels(y)
        if self.broadcast_buffers and self.world > 1:
            self.sync_buffers()            # DDP's per-forward buffer broadcast from rank 0
        self._describe(x)
        if self.fused_loss:
            self._forward_and_loss(x, y, True)
            self._backward(self.dlogits)
        else:
            logits = self._forward_scores(x)
            if self.three_head:
                self._loss_heads(y, True)
                self._backward(None)
            else:
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
        """Forward + loss only (validation_step, reference src/main.py:82-94); returns fresh (logits, loss) tensors."""
        self._describe(x)
        if self.fused_loss:
            logits = self._forward_and_loss(x, self._labels(y), False)
            return logits.clone(), self.loss
        logits = self._forward_scores(x)
        loss = self._loss_heads(self._labels(y), False) if self.three_head else self._loss(logits, self._labels(y), False)
        return logits.clone(), loss


class EnsembleTrainer:
    """Fused train step of the year ensemble (reference src/models/year.py:9-33) as the reference's MultiStage loop
    drives one level of it (src/models/multi_stage.py:277-288 training_step, :258-275 one Adam per level):
    scores = mean over the kept years of each year's last-head scores, loss = weighted CE, backward, Adam.

    All years share ONE set of flat parameter / gradient / moment buffers laid out
    [year 0..Y-1 first buckets | Y year flags | year 0..Y-1 first-conv weights]; each year's spectral_network is a
    FusedTrainer (restricted to the last head) on its two segments.  A year whose whole batch tensor sums to zero is
    skipped exactly as the reference skips it: no forward (BatchNorm running statistics and num_batches_tracked
    untouched), no gradient, and -- since torch's Adam passes over parameters whose grad is None -- no moment decay
    and no step-count advance for that year.  The zero-year test costs ONE host transfer per step (the reference: one
    blocking comparison per year); callers that know which years are present (the reference's dataset zero-fills
    missing years, src/data.py) pass `present` and the step enqueues without any host synchronisation.

    Data-parallel: a year is stepped when ANY rank kept it; ranks that skipped it contribute the zeros their gradient
    segment holds.  Which years were kept anywhere is decided ON THE DEVICE: every rank writes its 0/1 year flags into
    the flag slots of the gradient buffer, the SUM all-reduce of the first bucket carries them along, and the
    optimizer launches are gated by the reduced flags and a device-side step counter (`dta_adam_step_gated`) -- no
    extra collective and no host round trip.  Two collectives per step when overlapping (the first runs beside the
    years' first-conv weight gradients), one otherwise."""

    def __init__(self, model, lr, loss_weight=None, betas=(0.9, 0.999), eps=1e-8, process_group=None,
                 overlap_comm=True, keep_grads=False, exchange=None, exchange_opts=None):
        """exchange / exchange_opts: as for FusedTrainer ("peer": one all-reduce launch through peer memory before the
        gated optimizer passes; "rccl": one ncclAllReduce on the compute stream; "torch": two overlapped buckets)."""
        from .year import learned_ensemble
        if not isinstance(model, learned_ensemble):
            raise TypeError("EnsembleTrainer needs a deeptreeattention_amd.year.learned_ensemble")
        self.model = model
        mods = list(model.year_models)
        Y = len(mods)
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("EnsembleTrainer needs the model on a ROCm device (model.cuda()); there is no CPU path")
        sizes = [FusedTrainer.bucket_sizes(m) for m in mods]
        n_head, n_tail = sum(h for h, _ in sizes), sum(t for _, t in sizes)
        world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            world = torch.distributed.get_world_size(process_group)
        self.flat = [torch.zeros(_round4(n_head + Y) + n_tail, dtype=torch.float32, device=dev) for _ in range(4)]   # p g m v
        self.ex, self.exchange = None, None
        # DTA_FORCE_COLLECTIVES=1 (development measurement switch, as for FusedTrainer): a one-rank process group still takes the
        # data-parallel path -- exchange launch, device-gated optimizer passes -- so that its fixed cost can be measured
        self.comm = world > 1 or (os.environ.get("DTA_FORCE_COLLECTIVES") == "1" and torch.distributed.is_available()
                                  and torch.distributed.is_initialized())
        if self.comm:
            self.exchange = choose_exchange(exchange, process_group) if world > 1 else (exchange or "torch")
            if self.exchange == "peer":
                # overlap_comm: the buffer is exchanged as head [all years' first segments | year flags] + tail [the years'
                # first-conv weights]; the head's sum over the ranks rides in the years' grouped first-conv weight-gradient
                # launch (dta_ensemble_backward_xchg), as FusedTrainer's does in dta_net_backward_xchg
                self.ex = PeerExchange(self.flat[1].numel(), process_group, split=(_round4(n_head + Y) if overlap_comm else 0),
                                       **{k: v for k, v in (exchange_opts or {}).items() if k != "side_stream"})
                assert self.ex.capacity == self.flat[1].numel()
                self.flat[1] = self.ex.grad           # the gradient buffer every peer has mapped
        self.years = []
        if not self.comm:
            # no exchange: each year's two segments adjacent, so its optimizer pass is one launch; flags unused
            off = 0
            for m, (h, t) in zip(mods, sizes):
                storage = (tuple(b[off:off + h] for b in self.flat), tuple(b[off + h:off + h + t] for b in self.flat))
                self.years.append(FusedTrainer(m, lr, loss_weight, betas, eps, None, overlap_comm, keep_grads,
                                               last_head_only=True, storage=storage))
                off += h + t
            self.g_head = self.g_tail = self.flags = None
        else:
            self.n_head = _round4(n_head + Y)          # first bucket: all years' first segments + the year flags
            self.g_head, self.g_tail = self.flat[1][:self.n_head], self.flat[1][self.n_head:]
            self.flags = self.g_head[n_head:n_head + Y]  # this rank's 0/1 flags -> after the reduce: ranks that kept it
            ho, to = 0, self.n_head
            for m, (h, t) in zip(mods, sizes):
                storage = (tuple(b[ho:ho + h] for b in self.flat), tuple(b[to:to + t] for b in self.flat))
                self.years.append(FusedTrainer(m, lr, loss_weight, betas, eps, process_group, overlap_comm, keep_grads,
                                               last_head_only=True, storage=storage))
                ho += h
                to += t
        first = self.years[0]
        self.device, self.world, self.pg = first.device, first.world, first.pg
        side_stream = bool((exchange_opts or {}).get("side_stream", False))      # (rccl: see FusedTrainer)
        self.overlap = bool(overlap_comm) and (self.exchange == "torch" or (self.exchange == "rccl" and side_stream))
        self.keep_grads = bool(keep_grads)
        self.overlap_comm = bool(self.overlap or (self.ex is not None and self.ex.split))      # what bench.py reports
        self.betas, self.eps = betas, float(eps)
        self.sync = first.sync if self.exchange != "rccl" else GradSync(world, process_group, rccl=RcclDirect(process_group))
        self.world = world
        self.loss_weight = first.loss_weight
        self.loss = torch.zeros((), dtype=torch.float32, device=self.device)
        self._shape = None
        self._ws = None
        self._ws_key = None
        # per-year optimizer step counts on the device, two banks used alternately (the gated launches read one, write the
        # other); single process: only the device-decided steps (present=None) use them, see _counters_to
        self.dev_steps = torch.zeros(2, Y, dtype=torch.int32, device=dev)
        self._bank = 0
        self._counters_on = "device" if self.comm else "host"
        # this rank's 0/1 year flags (dta_year_flags): two banks used alternately, each call clears the other one
        self._flag_banks = torch.zeros(2, Y, dtype=torch.float32, device=dev)
        self._flag_bank = 0
        self.local_flags = self._flag_banks[0]
        self.kept_dev = torch.zeros(2, dtype=torch.float32, device=dev)        # {years kept, 1 / years kept} of the last forward
        if self.comm:
            if self.world > 1:
                self.sync.broadcast([self.flat[0]] + list(model.buffers()), 0)
            # the 2^Y possible flag vectors, resident on the device: setting the flags is a device-to-device copy
            self._flag_table = torch.tensor([[(mask >> i) & 1 for i in range(Y)] for mask in range(1 << Y)],
                                            dtype=torch.float32, device=dev)

    @property
    def lr(self):
        return self.years[0].lr

    @lr.setter
    def lr(self, value):       # ReduceLROnPlateau-style schedulers set one rate per level
        for t in self.years:
            t.lr = float(value)

    def step_counts(self):
        """Optimizer steps taken per year (data-parallel: read back from the device counters, one host sync)."""
        if self._counters_on == "device":
            return [int(v) for v in self.dev_steps[self._bank].tolist()]
        return [t.step_count for t in self.years]

    def _counters_to(self, where):
        """Single process: steps with `present=` count on the host (one optimizer launch per kept year), device-decided
        steps (present=None) on the device.  Mixing the two moves the counts across once per switch (one small transfer,
        device -> host being a synchronisation); a loop that sticks to one form never pays it."""
        if self.comm or where == self._counters_on:
            return
        if where == "device":
            self.dev_steps[self._bank].copy_(torch.tensor([t.step_count for t in self.years], dtype=torch.int32))
        else:
            for t, n in zip(self.years, self.dev_steps[self._bank].tolist()):
                t.step_count = int(n)
        self._counters_on = where

    def grad_of(self, param):
        for t in self.years:
            if id(param) in t._gview:
                return t._gview[id(param)]
        raise KeyError("not a parameter of this ensemble")

    # ---- optimizer state in torch.optim.Adam's layout (one Adam per level over the ensemble's parameters,
    #      reference multi_stage.py:258-275); a year torch never stepped has no state there, step 0 and zero moments here ----
    def _year_of(self):
        owner = {}
        for i, t in enumerate(self.years):
            for p in t.model.parameters():
                owner[id(p)] = i
        return owner

    def optimizer_state_dict(self):
        params = list(self.model.parameters())
        owner, counts = self._year_of(), self.step_counts()
        state = {}
        for k, p in enumerate(params):
            i = owner[id(p)]
            if counts[i] == 0:
                continue
            m, v = self.years[i]._moment_views(p)
            state[k] = {"step": torch.tensor(float(counts[i])), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    @torch.no_grad()
    def load_optimizer_state_dict(self, sd):
        params = list(self.model.parameters())
        g = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or len(g["params"]) != len(params):
            raise ValueError("expected one parameter group over the ensemble's {} parameters".format(len(params)))
        if g.get("weight_decay") or g.get("amsgrad") or g.get("maximize"):
            raise ValueError("the fused step is Adam without weight decay / amsgrad / maximize (the reference's setting)")
        owner = self._year_of()
        counts = [set() for _ in self.years]
        for key, p in zip(g["params"], params):
            i = owner[id(p)]
            m, v = self.years[i]._moment_views(p)
            st = sd["state"].get(key)
            if st is None:
                m.zero_(); v.zero_()
                continue
            m.copy_(st["exp_avg"].to(m)); v.copy_(st["exp_avg_sq"].to(v))
            if float(st["step"]) > 0:
                counts[i].add(int(float(st["step"])))
        if any(len(c) > 1 for c in counts):
            raise ValueError("parameters of one year carry different step counts")
        per_year = [c.pop() if c else 0 for c in counts]
        for t, n in zip(self.years, per_year):
            t.step_count = n
        self.dev_steps[:] = torch.tensor(per_year, dtype=torch.int32, device=self.dev_steps.device)
        self.lr = float(g["lr"])
        self.betas, self.eps = tuple(g["betas"]), float(g["eps"])
        for t in self.years:
            t.betas, t.eps = self.betas, self.eps

    def _kept(self, images, present):
        if len(images) != len(self.years):
            raise ValueError("expected one image tensor per year ({}), got {}".format(len(self.years), len(images)))
        local = [bool(k) for k in present]      # (present=None never comes here: that decision is taken on the device)
        if not any(local) and not self.comm:
            raise RuntimeError("every year of the batch is all-zero: the reference has nothing to average (year.py:33)")
        return local

    def _buffers(self, B, classes):
        if self._shape != (B, classes):
            self.scores = torch.empty(B, classes, dtype=torch.float32, device=self.device)
            self.dscores = torch.empty_like(self.scores)
            self.ce_scratch = torch.zeros(B + 2, dtype=torch.float32, device=self.device)   # (last word: block counter)
            self._shape = (B, classes)

    def _forward(self, images, local, loss_y=None, want_grad=True):
        """All kept years as the groups of one set of launches (dta_ensemble_forward); self.scores = their mean.
        loss_y (labels): also the level's weighted cross-entropy -> self.loss, d(loss)/d(one year's scores) -> self.dscores,
        in the same C-ABI call (dta_ensemble_forward_loss)."""
        L = _lib.lib()
        kept = [i for i, k in enumerate(local) if k]
        if not kept:
            raise RuntimeError("every year of this rank's batch is all-zero: nothing to average (reference year.py:33)")
        if len(kept) > _lib.MAX_YEARS:
            raise RuntimeError("at most {} years per grouped launch (train_step / forward_loss chunk beyond that)".format(_lib.MAX_YEARS))
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
        if loss_y is not None:
            # forward + the level's loss in one call: the mean over the years is formed inside the loss launch
            self.loss = torch.empty((), dtype=torch.float32, device=self.device)      # fresh per step (see FusedTrainer._loss)
            _lib.check(L.dta_ensemble_forward_loss(C.byref(self._desc), n, self._nets, self._xptr, None, _lib.ptr(self._ws),
                                                   _lib.ptr(loss_y), _lib.ptr(self.loss_weight), _lib.ptr(self.scores), None,
                                                   _lib.ptr(self.loss), _lib.ptr(self.dscores) if want_grad else None,
                                                   _lib.ptr(self.ce_scratch), _lib.current_stream_ptr()), "dta_ensemble_forward_loss")
        else:
            _lib.check(L.dta_ensemble_forward(C.byref(self._desc), n, self._nets, self._xptr, _lib.ptr(self._ws),
                                              _lib.ptr(self.scores), _lib.current_stream_ptr()), "dta_ensemble_forward")
        self._live = xs     # inputs stay referenced until the step's launches are enqueued
        return kept

    def _forward_gated(self, images, loss_y=None, want_grad=True):
        """All years as the groups of one set of launches, the missing-year decision (reference year.py:27) taken ON THE
        DEVICE: dta_year_flags -> self.local_flags, dta_ensemble_forward_gated (a flagged-off year is left out of the mean
        and keeps its BatchNorm statistics), 1 / kept years left in self.kept_dev[1] for the loss launch.  No host
        synchronisation.  Returns the list of all year indices (what the backward launches)."""
        L = _lib.lib()
        Y = len(self.years)
        if len(images) != Y:
            raise ValueError("expected one image tensor per year ({}), got {}".format(Y, len(images)))
        if Y > _lib.MAX_YEARS:
            raise RuntimeError("at most {} years per grouped launch".format(_lib.MAX_YEARS))
        xs = [H._check_input(x) for x in images]
        if any(x.shape != xs[0].shape for x in xs):
            raise ValueError("all years of a batch must have the same shape")
        keys = [t._describe(x) for t, x in zip(self.years, xs)]
        B, classes = xs[0].shape[0], self.model.year_models[0]._classes
        self._buffers(B, classes)
        self._nets = (_lib.SubnetParams * Y)(*[t.nets[0] for t in self.years])
        self._grads = (_lib.SubnetGrads * Y)(*[t.grads[0] for t in self.years])
        self._xptr = (C.c_void_p * Y)(*[x.data_ptr() for x in xs])
        self._desc = self.years[0].desc
        ws_key = (Y,) + keys[0]
        if ws_key != self._ws_key:
            nbytes = L.dta_ensemble_workspace_bytes(C.byref(self._desc), Y)
            if nbytes == 0:
                raise RuntimeError("dta_ensemble_workspace_bytes: " + L.dta_last_error().decode())
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_key = ws_key
        st = _lib.current_stream_ptr()
        self._flag_bank ^= 1
        self.local_flags = self._flag_banks[self._flag_bank]
        _lib.check(L.dta_year_flags(self._xptr, Y, xs[0].numel(), _lib.ptr(self.local_flags),
                                    _lib.ptr(self._flag_banks[self._flag_bank ^ 1]), st), "dta_year_flags")
        if loss_y is not None:
            self.loss = torch.empty((), dtype=torch.float32, device=self.device)
            _lib.check(L.dta_ensemble_forward_loss(C.byref(self._desc), Y, self._nets, self._xptr, _lib.ptr(self.local_flags),
                                                   _lib.ptr(self._ws), _lib.ptr(loss_y), _lib.ptr(self.loss_weight),
                                                   _lib.ptr(self.scores), _lib.ptr(self.kept_dev), _lib.ptr(self.loss),
                                                   _lib.ptr(self.dscores) if want_grad else None, _lib.ptr(self.ce_scratch), st),
                       "dta_ensemble_forward_loss")
        else:
            _lib.check(L.dta_ensemble_forward_gated(C.byref(self._desc), Y, self._nets, self._xptr, _lib.ptr(self.local_flags),
                                                    _lib.ptr(self._ws), _lib.ptr(self.scores), _lib.ptr(self.kept_dev), st),
                       "dta_ensemble_forward_gated")
        self._live = xs
        return list(range(Y))

    # ---- more kept years than one grouped launch takes (DTA_MAX_YEARS = 16): chunks of 16, single process ----
    def _chunked_forward(self, images, local, y, want_grad):
        """The kept years in chunks of DTA_MAX_YEARS networks: one grouped forward per chunk (its mean over ITS years,
        dta_ensemble_forward, own workspace), the ensemble's scores = the chunk means weighted by their year counts
        (reference year.py:33: the mean over all kept years), then the level's weighted CE with d(mean)/d(year score) =
        1 / kept folded into the gradient (dta_weighted_ce_scaled).  Leaves what _chunked_backward needs in self._chunks."""
        L = _lib.lib()
        kept = [i for i, k in enumerate(local) if k]
        st = _lib.current_stream_ptr()
        xs_all = {i: H._check_input(images[i]) for i in kept}
        shape = xs_all[kept[0]].shape
        if any(x.shape != shape for x in xs_all.values()):
            raise ValueError("all years of a batch must have the same shape")
        B, classes = shape[0], self.model.year_models[0]._classes
        self._buffers(B, classes)
        cache = self.__dict__.setdefault("_chunk_ws", {})
        chunks = []
        self.scores.zero_()
        for ci, lo in enumerate(range(0, len(kept), _lib.MAX_YEARS)):
            part = kept[lo:lo + _lib.MAX_YEARS]
            xs = [xs_all[i] for i in part]
            keys = [self.years[i]._describe(x) for i, x in zip(part, xs)]
            n = len(part)
            nets = (_lib.SubnetParams * n)(*[self.years[i].nets[0] for i in part])
            grads = (_lib.SubnetGrads * n)(*[self.years[i].grads[0] for i in part])
            xptr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
            desc = self.years[part[0]].desc
            wkey = (ci, n) + keys[0]
            hit = cache.get(ci)
            if hit is None or hit[0] != wkey:
                nbytes = L.dta_ensemble_workspace_bytes(C.byref(desc), n)
                if nbytes == 0:
                    raise RuntimeError("dta_ensemble_workspace_bytes: " + L.dta_last_error().decode())
                hit = cache[ci] = (wkey, torch.empty(nbytes, dtype=torch.uint8, device=self.device),
                                   torch.empty(B, classes, dtype=torch.float32, device=self.device))
            ws, mean_c = hit[1], hit[2]
            _lib.check(L.dta_ensemble_forward(C.byref(desc), n, nets, xptr, _lib.ptr(ws), _lib.ptr(mean_c), st), "dta_ensemble_forward")
            self.scores.add_(mean_c, alpha=n / len(kept))
            chunks.append((part, desc, nets, grads, ws, xs))
        self.loss = torch.empty((), dtype=torch.float32, device=self.device)
        _lib.check(L.dta_weighted_ce_scaled(_lib.ptr(self.scores), _lib.ptr(y), _lib.ptr(self.loss_weight), B, classes,
                                            1.0 / len(kept), _lib.ptr(self.loss), _lib.ptr(self.dscores) if want_grad else None,
                                            _lib.ptr(self.ce_scratch), st), "dta_weighted_ce_scaled")
        self._chunks = chunks
        return kept

    def _chunked_backward(self):
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        for part, desc, nets, grads, ws, xs in self._chunks:
            for i in part:
                self.years[i]._zero_grads()
            _lib.check(L.dta_ensemble_backward_gated(C.byref(desc), len(part), nets, _lib.ptr(ws), _lib.ptr(self.dscores), grads,
                                                     None, 3, st), "dta_ensemble_backward")
            for i in part:
                self.years[i]._grads_clear = False

    def _host_flags(self, images):
        """year.py:27 for every year in ONE transfer (the chunked path only: beyond DTA_MAX_YEARS networks the missing-year
        decision is the host's, as on the module path)."""
        return [bool(v) for v in torch.stack([x.ne(0).any() for x in images]).tolist()]

    def _backward_xchg(self, kept, gate=None):
        """The whole backward with the exchange's head segment (every gradient but the years' first-conv weights, and the
        year flags) summed over the ranks by spare workgroups of the years' first-conv weight-gradient launch."""
        L = _lib.lib()
        for i in kept:
            self.years[i]._zero_grads()
        _lib.check(L.dta_ensemble_backward_xchg(C.byref(self._desc), len(kept), self._nets, _lib.ptr(self._ws),
                                                _lib.ptr(self.dscores), self._grads, _lib.ptr(gate), self.ex._h,
                                                _lib.current_stream_ptr()), "dta_ensemble_backward_xchg")
        for i in kept:
            self.years[i]._grads_clear = False

    def _backward(self, kept, phases=3, gate=None):
        """gate (device, float[len(kept)]): this rank's year flags of a device-decided step -- a year whose flag is 0 gets
        EXACT-ZERO gradients (reference year.py:27-28: a skipped year has no gradient), so a data-parallel rank whose
        batch lacks a year that another rank kept contributes nothing to that year's gradient sum."""
        L = _lib.lib()
        if phases & 1:
            for i in kept:
                self.years[i]._zero_grads()      # C-ABI contract: gradient buffers arrive zero-filled
        _lib.check(L.dta_ensemble_backward_gated(C.byref(self._desc), len(kept), self._nets, _lib.ptr(self._ws),
                                                 _lib.ptr(self.dscores), self._grads, _lib.ptr(gate), phases,
                                                 _lib.current_stream_ptr()), "dta_ensemble_backward")
        for i in kept:
            self.years[i]._grads_clear = False

    @staticmethod
    def _year_segments(t):
        """(p, g, m, v, n) of one year's flat segments: one when head and tail are adjacent (single process), else two."""
        if all(h.data_ptr() + 4 * t.split == tl.data_ptr() for h, tl in
               ((t.p_head, t.p_tail), (t.g_head, t.g_tail), (t.m_head, t.m_tail), (t.v_head, t.v_tail))):
            return ((t.p_head, t.g_head, t.m_head, t.v_head, t.n),)
        return ((t.p_head, t.g_head, t.m_head, t.v_head, t.split), (t.p_tail, t.g_tail, t.m_tail, t.v_tail, t.n_first))

    def _adam_multi(self, segs):
        """All of `segs` (_lib.AdamSegment entries) in launches of up to DTA_ADAM_MAX_SEGMENTS parameter groups: the years of an
        ensemble are stepped by ONE launch instead of one per year and segment (3 x 369 x 24x24, B = 256: three k_adam
        launches of 5.6 us + two boundaries -> one)."""
        L = _lib.lib()
        lr = self.years[0].lr
        for lo in range(0, len(segs), _lib.ADAM_MAX_SEGMENTS):
            part = segs[lo:lo + _lib.ADAM_MAX_SEGMENTS]
            arr = (_lib.AdamSegment * len(part))(*part)
            _lib.check(L.dta_adam_step_multi(len(part), arr, lr, self.betas[0], self.betas[1], self.eps, self.sync.grad_scale,
                                             0 if self.keep_grads else 1, _lib.current_stream_ptr()), "dta_adam_step_multi")

    def _adam_gated(self, flags=None):
        """The gated optimizer passes of all years and segments in one launch, driven by `flags` (data-parallel: the reduced
        year flags in the gradient buffer; single process: this rank's dta_year_flags) and the device step counters."""
        flags = self.flags if flags is None else flags
        Y = len(self.years)
        cur, nxt = self._bank, 1 - self._bank
        segs = []
        for i, t in enumerate(self.years):
            active = flags.data_ptr() + 4 * i
            step = self.dev_steps.data_ptr() + 4 * (cur * Y + i)
            step_next = self.dev_steps.data_ptr() + 4 * (nxt * Y + i)
            for k, (p, g, m, v, n) in enumerate(self._year_segments(t)):
                segs.append(_lib.AdamSegment(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, active, step,
                                             step_next if k == 0 else None, 0))
            t._grads_clear = not self.keep_grads
        self._adam_multi(segs)
        self._bank = nxt

    def _adam_present(self, local):
        """Host-counted steps (present=[...], single process): the kept years' passes in one launch."""
        segs = []
        for i, t in enumerate(self.years):
            if not local[i]:
                continue
            t.step_count += 1
            for p, g, m, v, n in self._year_segments(t):
                segs.append(_lib.AdamSegment(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, None, None, None, t.step_count))
            t._grads_clear = not self.keep_grads
        if segs:
            self._adam_multi(segs)

    def train_step(self, images, y, present=None):
        """images: list of (B, bands, H, W) float32 device tensors, one per year; y: int64 labels.  Returns the loss
        as a fresh 0-d device tensor.
        present=None: which years are missing (reference year.py:27: the whole tensor sums to zero) is decided ON THE
        DEVICE -- all years are launched, a missing one is left out of the mean, keeps its BatchNorm statistics and is
        not stepped -- and the call returns without any host synchronisation.  (No year present: the loss is NaN; the
        reference raises.)  present=[...]: the caller knows (the reference's loader zero-fills missing years on the host):
        only the kept years are launched at all."""
        self.check_exchange()
        y = self.years[0]._labels(y)
        gate = None
        if len(self.years) > _lib.MAX_YEARS:
            # more years than one grouped launch takes: chunks of DTA_MAX_YEARS (the reference takes the year count from the
            # data, multi_stage.py:39, :61-66); missing years decided on the host there (one transfer), single process only
            if self.comm:
                raise RuntimeError("a data-parallel year ensemble takes at most {} years".format(_lib.MAX_YEARS))
            local = self._kept(images, present if present is not None else self._host_flags(images))
            if sum(local) > _lib.MAX_YEARS:
                self._counters_to("host")
                self._chunked_forward(images, local, y, True)
                self._chunked_backward()
                self._adam_present(local)
                return self.loss
            present = local
        if present is not None and self.ex is not None and self.ex.split:
            # overlapped peer exchange: the head's sum rides in the years' first-conv weight-gradient launch, and whether that
            # combined kernel exists depends on how many years a rank launches.  Host flags differ from rank to rank (and a
            # rank may call with present=None while another passes flags), so the launch plan would too, one rank waiting for
            # head flags another never posts.  Every rank therefore takes the device-decided form -- all Y years
            # launched, the missing ones gated by dta_year_flags, which computes exactly what truthful host flags say.
            present = None
        if present is None:
            self._counters_to("device")
            kept = self._forward_gated(images, y, True)
            if not self.comm:
                self._backward(kept)
                self._adam_gated(self.local_flags)
                return self.loss
            local_flags = gate = self.local_flags      # all years are launched: the flags zero the missing ones' gradients
        else:
            local = self._kept(images, present)
            kept = self._forward(images, local, y, True)
            if not self.comm:
                self._counters_to("host")
                self._backward(kept)
                self._adam_present(local)
                return self.loss
            local_flags = self._flag_table[sum(1 << i for i, k in enumerate(local) if k)]
        # data-parallel: every rank issues the same collectives whatever it kept; skipped years send their zeros
        if self.ex is not None:
            self.flags.copy_(local_flags)       # (before the backward: the flags ride in the head segment)
            # The head's sum rides in the years' first-conv weight-gradient launch only when EVERY rank launches the same
            # plan: with host-side present= flags each rank launches its own kept years, and whether the combined kernel
            # exists for a plan depends on the group count (capi.hip: launch_conv_wgrad_bf16_xchg) -- ranks that kept
            # different year counts could disagree, one waiting for head flags another never posts.  The
            # device-decided step launches all Y years on every rank (train_step turns present= calls into it when ex.split).
            if self.ex.split:
                self._backward_xchg(kept, gate) # head summed beside the first convs' weight gradients
            else:
                self._backward(kept, 3, gate)
            self.ex.allreduce()                 # gradients and year flags summed over the ranks in one launch (head done: the tail)
        elif self.overlap:
            self._backward(kept, 1, gate)
            self.flags.copy_(local_flags)
            self.sync.reduce_early(self.g_head)
            self._backward(kept, 2, gate)
            self.sync.reduce_late(self.g_tail)
        else:
            self._backward(kept, 3, gate)
            self.flags.copy_(local_flags)
            self.sync.reduce_all(self.flat[1])
        self.sync.finish()
        self._adam_gated()
        return self.loss

    def check_exchange(self):
        """Raise if a peer-exchange launch timed out waiting for a rank (see FusedTrainer.check_exchange)."""
        if self.ex is not None:
            self.ex.check()

    def close(self):
        """Collective: release the peer exchange / RCCL communicator (every rank calls it)."""
        if self.ex is not None:
            for t in self.years:
                t._gview = {}
                t.g_head = t.g_tail = None
            self.g_head = self.g_tail = self.flags = None
            self.flat[1] = None
            self.ex.close()
            self.ex = None
        if self.sync.rccl is not None:
            self.sync.rccl.close()
            self.sync.rccl = None

    def forward_loss(self, images, y, present=None):
        """validation_step of the level (multi_stage.py:290-304): ensemble scores + weighted CE, no update."""
        y = self.years[0]._labels(y)
        if len(self.years) > _lib.MAX_YEARS:
            local = self._kept(images, present if present is not None else self._host_flags(images))
            if sum(local) > _lib.MAX_YEARS:
                self._chunked_forward(images, local, y, False)
            else:
                self._forward(images, local, y, False)
        elif present is None:
            self._forward_gated(images, y, False)
        else:
            self._forward(images, self._kept(images, present), y, False)
        return self.scores.clone(), self.loss


class MultiStageTrainer:
    """Step driver of the reference's hierarchical model (src/models/multi_stage.py): one year ensemble, one class
    weight vector, one Adam and one learning rate per level, selected by Lightning's optimizer_idx /
    dataloader_idx.  Batches keep the reference's structure: (individual, {"HSI": [year tensors]}, labels)."""

    def __init__(self, models, lrs, loss_weights=None, **kwargs):
        loss_weights = loss_weights or [None] * len(models)
        self.levels = [EnsembleTrainer(m, lr, w, **kwargs) for m, lr, w in zip(models, lrs, loss_weights)]
        self._ws = None
        self._ws_key = None
        self._gate_banks = None
        self._gate_bank = 0
        self.batched_last = False        # did the last training_step_all run as one launch chain?

    # ---- every level of a batch in ONE launch chain (reference train.py:75-100 + multi_stage.py:277-288: Lightning calls
    #      training_step once per optimizer -- i.e. per level -- on every batch; the levels are independent networks) ----
    def _batchable(self, xs_levels):
        """The levels' steps can share one set of launches when nothing is exchanged between ranks, every level brings the
        same batch size / crop shape / precision / mode, and the networks fit one grouped launch."""
        if len(self.levels) > _lib.MAX_LEVELS or any(t.comm for t in self.levels):
            return False
        shapes = {tuple(x.shape) for xs in xs_levels for x in xs}
        if len(shapes) != 1:
            return False
        m0 = self.levels[0].model.year_models[0]
        for t in self.levels:
            for m in t.model.year_models:
                if m.precision != m0.precision or m.training != m0.training or not m.training:
                    return False
        return True

    def training_step_all(self, batch, batch_idx=0, present=None):
        """One optimisation step of EVERY level on its batch: batch[l] = (individual, {"HSI": [year tensors]}, labels) as
        Lightning hands them to MultiStage.training_step (multi_stage.py:277-288, once per optimizer_idx).  Returns the
        list of per-level losses (fresh 0-d device tensors).  The levels x kept-years networks run as the groups of ONE
        launch chain -- one forward, ONE loss launch over the levels, one backward, one optimizer launch per 16 parameter
        segments, each level with its own class weights and learning rate (dta_multistage_*); results are those of
        training_step(batch, ., l) for l = 0, 1, ... in turn.  present: None (missing years decided on the device, no host
        synchronisation), or one list of booleans per level.  Steps the levels one after the other when they cannot share a
        launch chain (different batch sizes or shapes, more than 16 kept networks, data-parallel trainers)."""
        nl = len(self.levels)
        if len(batch) != nl:
            raise ValueError("expected one batch per level ({}), got {}".format(nl, len(batch)))
        if present is not None and len(present) != nl:
            raise ValueError("present: one list of year flags per level")
        xs_levels = [[H._check_input(x) for x in b[1]["HSI"]] for b in batch]
        for t, xs in zip(self.levels, xs_levels):
            if len(xs) != len(t.years):
                raise ValueError("expected one image tensor per year ({}), got {}".format(len(t.years), len(xs)))
        if present is None:
            kept = [list(range(len(t.years))) for t in self.levels]
        else:
            kept = [[i for i, k in enumerate(t._kept(xs, pr)) if k] for t, xs, pr in zip(self.levels, xs_levels, present)]
        total = sum(len(k) for k in kept)
        self.batched_last = self._batchable(xs_levels) and total <= _lib.MAX_YEARS and all(kept)
        if not self.batched_last:
            return [self.training_step(batch, batch_idx, l, None if present is None else present[l]) for l in range(nl)]
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        dev = self.levels[0].device
        nets, grads, xptr, live, lv = [], [], [], [], []
        B = xs_levels[0][0].shape[0]
        key = None
        for l, (t, xs, b) in enumerate(zip(self.levels, xs_levels, batch)):
            t.check_exchange()
            y = t.years[0]._labels(b[2])
            classes = t.model.year_models[0]._classes
            t._buffers(B, classes)
            t._counters_to("device" if present is None else "host")
            t.loss = torch.empty((), dtype=torch.float32, device=dev)      # fresh per step (see FusedTrainer._loss)
            first = len(nets)
            for i in kept[l]:
                k = t.years[i]._describe(xs[i])
                key = key or k
                nets.append(t.years[i].nets[0]); grads.append(t.years[i].grads[0]); xptr.append(xs[i].data_ptr())
            live.append((xs, y))
            lv.append(_lib.Level(classes, first, len(kept[l]), y.data_ptr(), _lib.ptr(t.loss_weight), t.scores.data_ptr(),
                                 t.kept_dev.data_ptr(), t.loss.data_ptr(), t.dscores.data_ptr(), t.ce_scratch.data_ptr()))
        n = len(nets)
        nets_a = (_lib.SubnetParams * n)(*nets)
        grads_a = (_lib.SubnetGrads * n)(*grads)
        x_a = (C.c_void_p * n)(*xptr)
        lv_a = (_lib.Level * nl)(*lv)
        desc = self.levels[0].years[kept[0][0]].desc
        ws_key = (tuple((v.classes, v.count) for v in lv),) + key
        if ws_key != self._ws_key:
            nbytes = L.dta_multistage_workspace_bytes(C.byref(desc), nl, lv_a)
            if nbytes == 0:
                raise RuntimeError("dta_multistage_workspace_bytes: " + L.dta_last_error().decode())
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._ws_key = ws_key
        gate = None
        if present is None:
            # the reference's missing-year test (year.py:27) for every network of the step in one launch, on the device
            if self._gate_banks is None or self._gate_banks.shape[1] != n:
                self._gate_banks = torch.zeros(2, n, dtype=torch.float32, device=dev)
                self._gate_bank = 0
            self._gate_bank ^= 1
            gate = self._gate_banks[self._gate_bank]
            _lib.check(L.dta_year_flags(x_a, n, xs_levels[0][0].numel(), _lib.ptr(gate),
                                        _lib.ptr(self._gate_banks[self._gate_bank ^ 1]), st), "dta_year_flags")
        _lib.check(L.dta_multistage_forward_loss(C.byref(desc), nl, lv_a, nets_a, x_a, _lib.ptr(gate), _lib.ptr(self._ws), st),
                   "dta_multistage_forward_loss")
        for t, ks in zip(self.levels, kept):
            for i in ks:
                t.years[i]._zero_grads()      # C-ABI contract: gradient buffers arrive zero-filled
        _lib.check(L.dta_multistage_backward(C.byref(desc), nl, lv_a, nets_a, _lib.ptr(self._ws), grads_a, _lib.ptr(gate), st),
                   "dta_multistage_backward")
        # ---- one Adam per level (multi_stage.py:258-262: its own learning rate), all of them in launches of 16 segments ----
        segs, g = [], 0
        for l, (t, ks) in enumerate(zip(self.levels, kept)):
            cur, nxt = t._bank, 1 - t._bank
            Y = len(t.years)
            for i in ks:
                yr = t.years[i]
                yr._grads_clear = False
                if present is None:
                    active = gate.data_ptr() + 4 * g
                    step = t.dev_steps.data_ptr() + 4 * (cur * Y + i)
                    step_next = t.dev_steps.data_ptr() + 4 * (nxt * Y + i)
                    for k, (p, gr, m, v, cnt) in enumerate(t._year_segments(yr)):
                        segs.append(_lib.AdamSegment(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), cnt, active, step,
                                                     step_next if k == 0 else None, 0, yr.lr))
                else:
                    yr.step_count += 1
                    for p, gr, m, v, cnt in t._year_segments(yr):
                        segs.append(_lib.AdamSegment(p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), cnt, None, None, None,
                                                     yr.step_count, yr.lr))
                yr._grads_clear = not t.keep_grads
                g += 1
            if present is None:
                t._bank = nxt
                t.local_flags = gate[lv[l].first:lv[l].first + lv[l].count]
        t0 = self.levels[0]
        if any(t.betas != t0.betas or t.eps != t0.eps or t.keep_grads != t0.keep_grads for t in self.levels):
            raise RuntimeError("training_step_all: the levels' Adam settings (betas, eps, keep_grads) must agree")
        for lo in range(0, len(segs), _lib.ADAM_MAX_SEGMENTS):
            part = segs[lo:lo + _lib.ADAM_MAX_SEGMENTS]
            arr = (_lib.AdamSegment * len(part))(*part)
            _lib.check(L.dta_adam_step_multi(len(part), arr, t0.lr, t0.betas[0], t0.betas[1], t0.eps, t0.sync.grad_scale,
                                             0 if t0.keep_grads else 1, st), "dta_adam_step_multi")
        self._live = live      # inputs and labels stay referenced until the step's launches are enqueued
        return [t.loss for t in self.levels]

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
        """multi_stage.py:306-318: every level's softmax scores for the same crops: ONE eval-mode launch chain over the
        levels x years networks (MultiStagePredictor; a cached Predictor per level when they cannot share a chain); the year
        ensembles' zero years are skipped as in training."""
        individual, inputs = batch
        if not hasattr(self, "_predictors"):
            self._predictors = [Predictor(t.model) for t in self.levels]
            self._ms_predictor = MultiStagePredictor([t.model for t in self.levels])
        images = inputs["HSI"]
        if self._ms_predictor.supported(len(images)):
            # every level sees the same crops: all levels x years networks in ONE eval-mode launch chain
            return individual, [o[0].clone() for o in self._ms_predictor(images, True, present)]
        return individual, [pr(images, True, present)[0].clone() for pr in self._predictors]


class MetadataTrainer:
    """Fused train step of the site-metadata fusion model (reference src/models/metadata.py): the step
    MetadataModel.training_step defines (:52-63: unweighted F.cross_entropy(model(images, site), y)) with Adam.
    The HSI branch (Hang2020, >99.9 % of the work) runs through the fused C-ABI pieces on flat buffers; the 16-wide
    site MLP and the 2*classes -> classes fusion layer (<0.2 MFLOP per sample) stay a small torch
    autograd graph joined to the HSI branch at its (B, classes) scores (graph_head=True replays it as ONE hipGraph per
    step: for hosts that cannot issue its ~35 small launches in the 0.2 ms they take on the GPU).  Their parameters LIVE in spare slots of the HSI branch's flat buffers (values, gradients -- the .grad of each is a
    persistent view autograd accumulates into --, Adam moments): the one Adam launch steps them, the one gradient exchange
    sums them, nothing is copied around either."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, process_group=None, overlap_comm=True,
                 keep_grads=False, exchange=None, exchange_opts=None, graph_head=False, native_head=True):
        from .metadata import metadata_sensor_fusion
        if not isinstance(model, metadata_sensor_fusion):
            raise TypeError("MetadataTrainer needs a deeptreeattention_amd.metadata.metadata_sensor_fusion")
        self.model = model
        self.small = list(model.metadata_model.parameters()) + list(model.fc1.parameters())
        self.small_sizes = [p.numel() for p in self.small]
        small_off, small_n = _aligned_offsets(self.small)       # every tensor on a 16-byte boundary (the head GEMMs' vector loads)
        self.sensor = FusedTrainer(model.sensor_model, lr, None, betas, eps, process_group, overlap_comm, keep_grads,
                                   extra_grad_slots=small_n, exchange=exchange, exchange_opts=exchange_opts)
        # native_head: the site MLP / fusion layer / loss run as csrc/meta.hip (dta_meta_head_*: ~12 launches); False: the same
        # graph as stock torch ops (~35 launches), which is also what the parity tests compare the native form with
        self.native_head = bool(native_head)
        self._mh = None                           # native head: (key, workspace, out, dlogits, dscores, loss scratch)
        self.sensor.external_loss = True         # the loss is taken on the fused (HSI + site) scores, by torch
        self.world, self.pg = self.sensor.world, self.sensor.pg
        self.graph_head = bool(graph_head)
        self._graph = None                       # (key, CUDAGraph, static scores leaf, site, y, loss)
        if any(p.dtype != torch.float32 or p.device != self.sensor.device for p in self.small):
            raise RuntimeError("MetadataTrainer needs the whole model in float32 on the sensor model's device")
        self._gviews = []
        with torch.no_grad():
            for p, k, o in zip(self.small, self.small_sizes, small_off):
                off = self.sensor.extra_off + o
                self.sensor.p_head[off:off + k].copy_(p.reshape(-1))
                p.data = self.sensor.p_head[off:off + k].view(p.shape)
                self._gviews.append(self.sensor.g_head[off:off + k].view(p.shape))
        self._attach_grads()
        if self.world > 1:
            for t in self.small + list(model.metadata_model.buffers()):
                torch.distributed.broadcast(t.data, 0, group=self.pg)

    def _attach_grads(self):
        if not self._gviews:
            raise RuntimeError("MetadataTrainer was closed")
        for p, g in zip(self.small, self._gviews):
            if p.grad is not g:
                p.grad = g

    @property
    def lr(self):
        return self.sensor.lr

    @lr.setter
    def lr(self, value):
        self.sensor.lr = float(value)

    def _head(self, scores, site):
        meta = self.model.metadata_model(site)
        return torch.relu(self.model.fc1(torch.cat([meta, scores], dim=1)))

    @staticmethod
    def _ce(scores, y):
        from .optim import cross_entropy
        return cross_entropy(scores, y)          # loss + d(loss)/d(scores) in one launch

    # ---- the small torch graph (site MLP, fusion layer, loss and their backward) as ONE hipGraph ----
    # torch's semantics are untouched (same kernels, same Philox stream for the dropout): the eager path below is the same
    # code issued launch by launch, and what a platform without graph capture falls back to.
    def _capture_head(self, site, y):
        import torch.nn.functional as F
        dev = self.sensor.device
        leaf = self.sensor.logits.detach().requires_grad_(True)        # the HSI scores buffer itself: written by the C ABI each step
        g_site, g_y = site.clone(), y.clone()
        bufs = [b for m in (self.model.metadata_model, self.model.fc1) for b in m.buffers()]
        saved = ([b.clone() for b in bufs], torch.cuda.get_rng_state(dev))

        def one():
            loss = self._ce(self._head(leaf, g_site), g_y)
            loss.backward()
            return loss

        def restore():
            with torch.no_grad():
                for b_, v in zip(bufs, saved[0]):
                    b_.copy_(v)
                self.sensor.extra_g.zero_()                             # what the trial backwards accumulated
            torch.cuda.set_rng_state(saved[1], dev)

        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                              # torch's capture recipe: warm up on a side stream
                for _ in range(2):
                    leaf.grad = None
                    one()
            torch.cuda.current_stream(dev).wait_stream(side)
            restore()
            leaf.grad = None
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                loss = one()
        finally:
            restore()                                                   # (capturing runs nothing; the RNG bookkeeping is reset anyway)
        return graph, leaf, g_site, g_y, loss

    def _graph_step(self, site, y):
        key = (self.sensor.logits.data_ptr(), tuple(self.sensor.logits.shape), float(self.model.metadata_model.dropout.p),
               self.model.training, self.sensor.g_head.data_ptr())
        if self._graph is None or self._graph[0] != key:
            self._graph = (key,) + self._capture_head(site, y)
        _, graph, leaf, g_site, g_y, loss = self._graph
        g_site.copy_(site)
        g_y.copy_(y)
        graph.replay()
        return leaf.grad, loss

    # ---- native head (csrc/meta.hip) ----
    def _native_tables(self):
        mm, fc = self.model.metadata_model, self.model.fc1
        bn = mm.batch_norm
        P = _lib.MetaParams(mm.embedding.weight.data_ptr(), bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                            bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(), mm.mlp.weight.data_ptr(),
                            mm.mlp.bias.data_ptr(), fc.weight.data_ptr(), fc.bias.data_ptr())
        gv = dict(zip([id(p) for p in self.small], self._gviews))
        g = lambda p: gv[id(p)].data_ptr()
        G = _lib.MetaGrads(g(mm.embedding.weight), g(bn.weight), g(bn.bias), g(mm.mlp.weight), g(mm.mlp.bias), g(fc.weight), g(fc.bias))
        return P, G

    def _native_buffers(self, B, classes, sites):
        L = _lib.lib()
        key = (B, classes, sites)
        if self._mh is None or self._mh[0] != key:
            nbytes = L.dta_meta_head_workspace_bytes(B, classes, sites)
            if nbytes == 0:
                raise RuntimeError("dta_meta_head_workspace_bytes: " + L.dta_last_error().decode())
            dev = self.sensor.device
            f = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
            self._mh = (key, torch.empty(nbytes, dtype=torch.uint8, device=dev), f(B, classes), f(B, classes), f(B, classes), f(B + 2))
        return self._mh[1:]

    def _native_forward(self, scores, site, training):
        """The fused scores (B, classes) of the whole model for HSI scores `scores`; returns (out, drop factors or None)."""
        L = _lib.lib()
        mm = self.model.metadata_model
        B, classes = scores.shape
        sites = mm.embedding.num_embeddings
        ws, out, _, _, _ = self._native_buffers(B, classes, sites)
        bn = mm.batch_norm
        if bn.momentum is None or not bn.track_running_stats or not bn.affine:
            raise RuntimeError("MetadataTrainer(native_head=True) supports the reference's BatchNorm1d settings (affine, running statistics, fixed momentum)")
        drop = None
        if training and mm.dropout.p > 0:
            # torch's generator draws the mask (the same Philox stream as the module's own dropout call); the kernels apply it
            drop = torch.nn.functional.dropout(torch.ones(B, mm.embedding.embedding_dim, dtype=torch.float32, device=scores.device),
                                               p=mm.dropout.p, training=True)
        P, _ = self._native_tables()
        _lib.check(L.dta_meta_head_forward(B, classes, sites, 1 if training else 0, float(bn.momentum), float(bn.eps), C.byref(P),
                                           _lib.ptr(site), _lib.ptr(scores), _lib.ptr(drop), _lib.ptr(ws), _lib.ptr(out),
                                           _lib.current_stream_ptr()), "dta_meta_head_forward")
        return out, drop

    def _native_step(self, scores, site, y):
        L = _lib.lib()
        mm = self.model.metadata_model
        B, classes = scores.shape
        sites = mm.embedding.num_embeddings
        ws, out, dlogits, dscores, scratch = self._native_buffers(B, classes, sites)
        training = bool(mm.training)              # the site branch's own mode: eval() -> running statistics, no dropout
        out, drop = self._native_forward(scores, site, training)
        loss = torch.empty((), dtype=torch.float32, device=scores.device)
        st = _lib.current_stream_ptr()
        _lib.check(L.dta_meta_head_loss(B, classes, _lib.ptr(out), _lib.ptr(y), _lib.ptr(loss), _lib.ptr(dlogits),
                                        _lib.ptr(scratch), st), "dta_meta_head_loss")
        P, G = self._native_tables()
        _lib.check(L.dta_meta_head_backward(B, classes, sites, 1 if training else 0, C.byref(P), _lib.ptr(site), _lib.ptr(drop), _lib.ptr(ws),
                                            _lib.ptr(out), _lib.ptr(dlogits), C.byref(G), _lib.ptr(dscores), st),
                   "dta_meta_head_backward")
        return dscores, loss

    def train_step(self, images, site, y):
        """images (B, bands, 11, 11) float32, site (B,) int64 site indices, y (B,) int64 labels -> loss (0-d tensor)."""
        y = self.sensor._labels(y)
        site = site if (site.dtype == torch.int64 and site.is_cuda) else site.to(self.sensor.device, torch.int64)
        scores = self.sensor._forward_scores(images)
        self.sensor._zero_grads()                 # the small backward accumulates into its slots of the flat gradient buffer
        self._attach_grads()
        if self.native_head and _lib.lib().dta_meta_head_workspace_bytes(
                scores.shape[0], scores.shape[1], self.model.metadata_model.embedding.num_embeddings) == 0:
            import warnings
            warnings.warn("MetadataTrainer: " + _lib.lib().dta_last_error().decode() + "; running the site / fusion head as torch ops")
            self.native_head = False
        if self.native_head:
            dscores, loss = self._native_step(scores, site.contiguous(), y)
            self.sensor._backward(dscores)
            self.sensor._adam()
            return loss
        dscores = loss = None
        if self.graph_head:
            try:
                dscores, loss = self._graph_step(site, y)
                loss = loss.detach().clone()
            except RuntimeError as e:             # a platform / torch build without graph capture keeps the eager path
                import warnings
                warnings.warn("MetadataTrainer: hipGraph capture of the site / fusion head failed ({}); running it eagerly".format(e))
                self.graph_head, self._graph = False, None
        if dscores is None:
            leaf = scores.detach().requires_grad_(True)
            loss = self._ce(self._head(leaf, site), y)
            loss.backward()                       # the small graph: MLP / fusion grads (into their slots) and d(loss)/d(scores)
            dscores, loss = leaf.grad.contiguous(), loss.detach()
        self.sensor._backward(dscores)            # + the gradient exchange (the slots ride in the first bucket)
        self.sensor._adam()                       # steps the slots' parameters too, and clears their gradients
        return loss

    def check_exchange(self):
        self.sensor.check_exchange()

    # ---- optimizer state of the WHOLE fusion model in torch.optim.Adam's layout (metadata.py:85-87: one Adam over
    #      self.model.parameters()); the small parameters' moments sit in the spare slots of the sensor's buffers ----
    def optimizer_state_dict(self):
        params = list(self.model.parameters())
        n = self.sensor.step_count
        state = {}
        for k, p in enumerate(params):
            m, v = self.sensor._moment_views(p)
            state[k] = {"step": torch.tensor(float(n)), "exp_avg": m.detach().clone(), "exp_avg_sq": v.detach().clone()}
        group = {"lr": self.lr, "betas": tuple(self.sensor.betas), "eps": self.sensor.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(params)))}
        return {"state": state if n > 0 else {}, "param_groups": [group]}

    @torch.no_grad()
    def load_optimizer_state_dict(self, sd):
        params = list(self.model.parameters())
        g = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or len(g["params"]) != len(params):
            raise ValueError("expected one parameter group over the model's {} parameters".format(len(params)))
        if g.get("weight_decay") or g.get("amsgrad") or g.get("maximize"):
            raise ValueError("the fused step is Adam without weight decay / amsgrad / maximize (the reference's setting)")
        steps = set()
        for key, p in zip(g["params"], params):
            m, v = self.sensor._moment_views(p)
            st = sd["state"].get(key)
            if st is None:
                m.zero_(); v.zero_()
                continue
            m.copy_(st["exp_avg"].to(m)); v.copy_(st["exp_avg_sq"].to(v))
            if float(st["step"]) > 0:
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("parameters carry different step counts: {}".format(sorted(steps)))
        self.sensor.step_count = steps.pop() if steps else 0
        self.lr = float(g["lr"])
        self.sensor.betas, self.sensor.eps = tuple(g["betas"]), float(g["eps"])

    def close(self):
        """Collective (data-parallel): drop the captured graph and the gradient views, release the sensor's exchange."""
        self._graph = None
        for p, g in zip(self.small, self._gviews):
            if p.grad is g:
                p.grad = None
        self._gviews = []
        self.sensor.close()

    def training_step(self, batch, batch_idx=0):
        """metadata.py:52-63 unpacking: batch = (individual, {"HSI": images, "site": site}, y)."""
        individual, inputs, y = batch
        return self.train_step(inputs["HSI"], inputs["site"], y)

    def validation_step(self, batch, batch_idx=0):
        """metadata.py:65-83: forward + unweighted CE, no update."""
        individual, inputs, y = batch
        with torch.no_grad():
            scores = self.sensor._forward_scores(inputs["HSI"])
            if self.native_head and not self.model.training:
                site = inputs["site"]
                site = site if (site.dtype == torch.int64 and site.is_cuda) else site.to(self.sensor.device, torch.int64)
                out, _ = self._native_forward(scores, site.contiguous(), False)
                return torch.nn.functional.cross_entropy(out, self.sensor._labels(y))
            return torch.nn.functional.cross_entropy(self._head(scores, inputs["site"]), self.sensor._labels(y))


class Predictor:
    """Inference step of the reference (`MultiStage.predict_step` src/models/multi_stage.py:306-318,
    `TreeModel.predict_dataloader` src/main.py:165-205): eval-mode forward, softmax over the classes and the top-2
    labels / scores, all on the device, with the descriptor, pointer tables, workspace and output buffers cached
    across calls (two C-ABI calls per batch, no per-call Python walk over the module tree).

    model: a network of this package (Hang2020, vanilla_CNN; spectral/spatial_network -> last head) or a
    year.learned_ensemble (zero years are skipped as in training: one host transfer per call, or pass `present`).
    The pointer tables are rebuilt when the batch shape changes; call refresh() after replacing parameter tensors
    (in-place updates, e.g. by FusedTrainer or load_state_dict, need nothing).
    frozen=True (tile prediction with a trained model, reference predict.py:140-151): the conv / attention weight
    re-layouts of the first call stay in the workspace and later calls skip them (DTA_REUSE_PACKED), and the per-call walk over
    the model's tensors shrinks to one sentinel per network (_fingerprint): whole-model changes made through torch (.to(),
    load_state_dict, an optimizer step) are still noticed; single edited tensors and this package's fused trainers need
    refresh()."""

    def __init__(self, model, frozen=False):
        import weakref
        self.frozen = bool(frozen)
        self._packed = False
        from .year import learned_ensemble
        self._model_ref = weakref.ref(model)        # no strong reference: the cache must not keep the model alive
        self.ensemble = isinstance(model, learned_ensemble)
        nets = list(model.year_models) if self.ensemble else [model]
        if not all(isinstance(n, H._Net) for n in nets):
            raise TypeError("Predictor needs a deeptreeattention_amd network or learned_ensemble")
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("Predictor needs the model on a ROCm device (model.cuda()); there is no CPU path")
        self._key = None
        self._tensors = None

    @property
    def model(self):
        return self._model_ref()

    @property
    def nets_mod(self):
        m = self.model
        return list(m.year_models) if self.ensemble else [m]

    def refresh(self):
        self._key = None
        self._tensors = None

    def _desc(self):
        """The descriptor of this call: with frozen weights every call after the first on a workspace reuses its re-layouts."""
        return self.desc_reuse if (self.frozen and self._packed) else self.desc

    def _fingerprint(self, versions=False):
        """Storage addresses of every parameter and BatchNorm buffer the pointer tables were built from.  The tables
        hold raw device pointers, so anything that re-homes a tensor (FusedTrainer moving the parameters into its flat
        buffer, model.to(), load_state_dict(assign=True)) must rebuild them: ~8 us of data_ptr() calls per batch buy
        that.  The tensor objects themselves are re-collected when a load_state_dict ran (it may replace Parameter
        objects; see _Net._dta_epoch)."""
        nets = self.nets_mod
        epoch = tuple(n.__dict__.get("_dta_epoch", 0) for n in nets)
        if self._tensors is None or self._tensors[0] != epoch:
            ts = []
            for n in nets:
                ts += list(n.parameters()) + list(n.buffers())
            self._tensors = (epoch, ts)
            self._sentinels = [next(t for t in n.parameters() if t.dim() == 4) for n in nets]
        if versions or self.frozen:
            # frozen weights: nothing but one sentinel per network -- its first conv weight's address and torch version
            # counter -- is looked at per call (the full walk is 65 us of host time for a 15-network MultiStage, more than
            # the re-layouts it guards cost on the device).  That catches what touches a whole model at once: .to(),
            # load_state_dict, a torch optimizer step, a trainer moving the parameters into its flat buffer.  A single edited
            # tensor, or updates by this package's fused trainers (raw pointers: no version counter moves), need refresh().
            return (epoch,) + tuple(v for t in self._sentinels for v in (t.data_ptr(), t._version))
        return (epoch,) + tuple(map(torch.Tensor.data_ptr, self._tensors[1]))

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
        nets_mod = self.nets_mod
        m0 = nets_mod[0]
        key = (tuple(shape), m0.precision, tuple(kept), self._fingerprint())
        if key == self._key:
            return
        B, bands, Hh, Ww = shape
        single = m0._net_code in (_lib.NET_HANG2020, _lib.NET_VANILLA)
        self.desc = _lib.NetDesc(B, bands, Hh, Ww, m0._classes, m0._net_code, _lib.dtype_code(m0.precision), 0,
                                 4 | _lib.FORWARD_ONLY, H.BN_MOMENTUM, H.BN_EPS)
        self.desc_reuse = _lib.NetDesc(B, bands, Hh, Ww, m0._classes, m0._net_code, _lib.dtype_code(m0.precision), 0,
                                       4 | _lib.FORWARD_ONLY | _lib.REUSE_PACKED, H.BN_MOMENTUM, H.BN_EPS)
        self._packed = False
        tables = self._tables([nets_mod[i] for i in kept])
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
        if self.ensemble and present is None:
            # missing years decided on the device (dta_year_flags), all years launched, no host round trip
            xs = [H._check_input(x) for x in images]
            kept = list(range(len(xs)))
            self._prepare(xs[0].shape, kept)
            if getattr(self, "_banks", None) is None or self._banks[0].numel() != len(xs):
                # two flag banks: a call writes one and zeroes the other for the next call (no clearing launch)
                self._banks = [torch.zeros(len(xs), dtype=torch.float32, device=self.device) for _ in range(2)]
                self._bank = 0
            flags, nxt = self._banks[self._bank], self._banks[self._bank ^ 1]
            self._bank ^= 1
            self._flags = flags
            xptr = (C.c_void_p * len(kept))(*[x.data_ptr() for x in xs])
            _lib.check(L.dta_year_flags(xptr, len(xs), xs[0].numel(), _lib.ptr(flags), _lib.ptr(nxt), st), "dta_year_flags")
            _lib.check(L.dta_ensemble_forward_gated(C.byref(self._desc()), len(kept), self.nets, xptr, _lib.ptr(flags),
                                                    _lib.ptr(self.ws), _lib.ptr(self.logits), None, st),
                       "dta_ensemble_forward_gated")
            self._packed = True
            return self.logits
        if self.ensemble:
            kept = [i for i, k in enumerate(present) if k]
            if not kept:
                raise RuntimeError("every year of the batch is all-zero: nothing to average (reference year.py:33)")
            xs = [H._check_input(images[i]) for i in kept]
            self._prepare(xs[0].shape, kept)
            xptr = (C.c_void_p * len(kept))(*[x.data_ptr() for x in xs])
            _lib.check(L.dta_ensemble_forward(C.byref(self._desc()), len(kept), self.nets, xptr, _lib.ptr(self.ws),
                                              _lib.ptr(self.logits), st), "dta_ensemble_forward")
            self._packed = True
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
        _lib.check(L.dta_net_forward(C.byref(self._desc()), self.nets, alpha, _lib.ptr(x), _lib.ptr(self.ws),
                                     C.byref(table), joint, st), "dta_net_forward")
        self._packed = True
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


class MultiStagePredictor:
    """`MultiStage.predict_step` (reference src/models/multi_stage.py:306-318: `for model in self.models: softmax(model(images))`
    on the SAME crops) as ONE launch chain: the levels x years networks are the groups of one eval-mode forward
    (dta_multistage_forward; every level's year-y network reads the same input tensor), each level's mean over its kept years,
    then softmax + top-2 per level (dta_softmax_top2).  Tile prediction is the reference's largest wall-clock consumer
    (SLURM/predict.sh), at `predict_batch_size: 64` -- where one chain per level is pure launch-latency floor.
    models: the levels' learned_ensembles (same year count, bands and precision).  Missing years are decided on the device
    unless `present` (one list of booleans, shared by the levels: they see the same crops) is passed."""

    def __init__(self, models, frozen=False):
        self.frozen = bool(frozen)      # as Predictor: keep the weight re-layouts of the first call (refresh() after updates)
        self._packed = False
        self.preds = [Predictor(m) for m in models]
        if not all(p.ensemble for p in self.preds):
            raise TypeError("MultiStagePredictor needs year.learned_ensemble levels")
        self.device = self.preds[0].device
        self._key = None

    def refresh(self):
        """After in-place weight updates of a frozen predictor (or replaced parameter tensors): rebuild everything."""
        self._key = None
        for p in self.preds:
            p.refresh()

    def supported(self, n_years):
        return (len(self.preds) <= _lib.MAX_LEVELS and len(self.preds) * n_years <= _lib.MAX_YEARS
                and all(len(p.nets_mod) == n_years for p in self.preds))

    def _prepare(self, shape, kept):
        L = _lib.lib()
        mods = [p.nets_mod for p in self.preds]
        m0 = mods[0][0]
        key = (tuple(shape), m0.precision, tuple(kept)) + tuple(p._fingerprint(self.frozen) for p in self.preds)
        if key == self._key:
            return
        if any(m.precision != m0.precision for ms in mods for m in ms):
            raise ValueError("all levels must run in the same precision")
        B, bands, Hh, Ww = shape
        self.desc = _lib.NetDesc(B, bands, Hh, Ww, m0._classes, _lib.NET_SPECTRAL, _lib.dtype_code(m0.precision), 0,
                                 4 | _lib.FORWARD_ONLY, H.BN_MOMENTUM, H.BN_EPS)
        self.desc_reuse = _lib.NetDesc(B, bands, Hh, Ww, m0._classes, _lib.NET_SPECTRAL, _lib.dtype_code(m0.precision), 0,
                                       4 | _lib.FORWARD_ONLY | _lib.REUSE_PACKED, H.BN_MOMENTUM, H.BN_EPS)
        self._packed = False
        nets, lv = [], []
        self.logits, self.probs, self.top_idx, self.top_score = [], [], [], []
        for p, ms in zip(self.preds, mods):
            classes = ms[0]._classes
            tables = p._tables([ms[i] for i in kept])
            first = len(nets)
            nets += [t[0] for t in tables]
            self.logits.append(torch.empty(B, classes, dtype=torch.float32, device=self.device))
            self.probs.append(torch.empty(B, classes, dtype=torch.float32, device=self.device))
            self.top_idx.append(torch.empty(B, 2, dtype=torch.int64, device=self.device))
            self.top_score.append(torch.empty(B, 2, dtype=torch.float32, device=self.device))
            lv.append(_lib.Level(classes, first, len(kept), None, None, self.logits[-1].data_ptr(), None, None, None, None))
        self.nets = (_lib.SubnetParams * len(nets))(*nets)
        self.lv = (_lib.Level * len(lv))(*lv)
        self._pp = (C.c_void_p * len(lv))(*[t.data_ptr() for t in self.probs])
        self._pi = (C.c_void_p * len(lv))(*[t.data_ptr() for t in self.top_idx])
        self._ps = (C.c_void_p * len(lv))(*[t.data_ptr() for t in self.top_score])
        nbytes = L.dta_multistage_workspace_bytes(C.byref(self.desc), len(lv), self.lv)
        if nbytes == 0:
            raise RuntimeError("dta_multistage_workspace_bytes: " + L.dta_last_error().decode())
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        # two flag banks: a call writes one and zeroes the other for the next call (no clearing launch)
        self._banks = [torch.zeros(len(self.preds[0].nets_mod), dtype=torch.float32, device=self.device) for _ in range(2)]
        self._bank = 0
        self._key = key

    def __call__(self, images, return_probs=True, present=None):
        """Returns one (probs or None, top_idx [B,2], top_score [B,2]) triple per level; buffers are reused across calls."""
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        nl = len(self.preds)
        Y = len(images)
        if present is None:
            kept = list(range(Y))
        else:
            kept = [i for i, k in enumerate(present) if k]
            if not kept:
                raise RuntimeError("every year of the batch is all-zero: nothing to average (reference year.py:33)")
        xs = [H._check_input(images[i]) for i in kept]
        if any(x.shape != xs[0].shape for x in xs):
            raise ValueError("all years of a batch must have the same shape")
        self._prepare(xs[0].shape, kept)
        xptr = (C.c_void_p * (nl * len(kept)))(*([x.data_ptr() for x in xs] * nl))      # every level reads the same crops
        gate = None
        if present is None:
            yptr = (C.c_void_p * Y)(*[x.data_ptr() for x in xs])
            flags, nxt = self._banks[self._bank], self._banks[self._bank ^ 1]
            self._bank ^= 1
            _lib.check(L.dta_year_flags(yptr, Y, xs[0].numel(), _lib.ptr(flags), _lib.ptr(nxt), st), "dta_year_flags")
            gate = flags.repeat(nl)                 # the (level, year) groups' flags: the years' flags once per level
        # forward of all levels x years + ONE launch for every level's mean over its years, softmax and top-2
        desc = self.desc_reuse if (self.frozen and self._packed) else self.desc
        _lib.check(L.dta_multistage_predict(C.byref(desc), nl, self.lv, self.nets, xptr, _lib.ptr(gate), _lib.ptr(self.ws),
                                            self._pp if return_probs else None, self._pi, self._ps, st), "dta_multistage_predict")
        self._packed = True
        self._live = (xs, gate)
        return [(self.probs[l] if return_probs else None, self.top_idx[l], self.top_score[l]) for l in range(nl)]


_PREDICTORS = None      # model -> Predictor, weakly keyed: nothing is stored on the module (deepcopy / torch.save stay clean)


def predict(model, images, return_probs=True):
    """Inference step of the reference (`MultiStage.predict_step` / `TreeModel.predict_dataloader`): eval-mode forward
    (whatever the module's current train/eval flag), softmax over classes and the top-2 labels/scores, all on the
    device, through a Predictor cached per module (weakly; its pointer tables follow the parameters wherever a trainer
    or `.to()` moves them).  Returns (probs or None, top_idx [B,2] int64, top_score [B,2] float32); the tensors are
    reused by the next call on the same module."""
    global _PREDICTORS
    if _PREDICTORS is None:
        import weakref
        _PREDICTORS = weakref.WeakKeyDictionary()
    pr = _PREDICTORS.get(model)
    if pr is None:
        pr = Predictor(model)
        _PREDICTORS[model] = pr
    return pr(images, return_probs)
