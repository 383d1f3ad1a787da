"""Drop-in optimizer and loss for the UNCHANGED reference step (TreeModel.training_step + configure_optimizers,
/root/reference/src/main.py:71-80, :135-149; MultiStage src/models/multi_stage.py:258-288):

    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    optimizer = DtaAdam(self.model.parameters(), lr=self.config["lr"])       # was optim.Adam(...)
    loss = cross_entropy(y_hat, y, weight=self.loss_weight)                  # was F.cross_entropy(...)

`DtaAdam` is a torch.optim.Optimizer (schedulers such as ReduceLROnPlateau attach to its param_groups): it moves the
float32 device parameters into ONE flat buffer (the Parameters become views, state_dict keys / shapes unchanged), gives
every parameter a persistent `.grad` that is a view of ONE flat gradient buffer, and `step()` is one launch of the HIP
Adam kernel over the flat buffers (`dta_adam_step_dp`; data-parallel: the sum over ranks + Adam in one launch through
peer memory, `dta_xchg_adam_step`, or one all-reduce of the flat buffer).  The networks' autograd node writes its
gradients straight into that buffer (Hang2020._NetFn.backward: no per-parameter copies, no torch foreach kernels), so the
module-level step costs three launches more than engine.FusedTrainer's.

Year ensembles (year.learned_ensemble): the reference skips a year whose batch tensor sums to zero, its parameters then
have grad None and torch's Adam passes over them (no moment decay, no step count).  Here that decision is taken on the
device: each year's parameters are one segment of the flat buffer, stepped by `dta_adam_step_gated` under the year's
"kept" flag of the last training forward, with per-year step counters on the device -- no host synchronisation.

`cross_entropy` is F.cross_entropy(input, target, weight) for (B, classes) float32 device scores: loss and
d(loss)/d(scores) in ONE launch (`dta_weighted_ce_scaled`)."""
import ctypes as C
import weakref

import torch

from . import Hang2020 as H
from . import _lib


def _round4(n):
    return (int(n) + 3) & ~3


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight):
        L = _lib.lib()
        B, classes = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits) if logits.requires_grad else None
        scratch = torch.zeros(B + 2, dtype=torch.float32, device=logits.device)
        _lib.check(L.dta_weighted_ce_scaled(_lib.ptr(logits), _lib.ptr(target), _lib.ptr(weight), B, classes, 1.0,
                                            _lib.ptr(loss), _lib.ptr(dl), _lib.ptr(scratch), _lib.current_stream_ptr()),
                   "dta_weighted_ce_scaled")
        ctx.dl = dl
        return loss

    @staticmethod
    def backward(ctx, gout):
        if ctx.dl is None or gout is None:
            return None, None, None
        return ctx.dl * gout, None, None      # (not in place: a second backward with retain_graph=True gets the same values)


def cross_entropy(input, target, weight=None):
    """F.cross_entropy(input, target, weight=weight) (mean reduction, ignore_index -100) of the reference's steps
    (src/main.py:78,89; multi_stage.py:285,297; metadata.py:61,74) for (B, classes) float32 scores on the device."""
    if input.dim() != 2 or not input.is_cuda:
        raise RuntimeError("deeptreeattention_amd.optim.cross_entropy takes (B, classes) scores on a ROCm device")
    x = input if (input.dtype == torch.float32 and input.is_contiguous()) else input.float().contiguous()
    t = target if (target.dtype == torch.int64 and target.is_cuda) else target.to(input.device, torch.int64)
    w = None if weight is None else weight.to(input.device, torch.float32).contiguous()
    return _CrossEntropyFn.apply(x, t.contiguous(), w)


class DtaAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) (no weight decay / amsgrad, as the reference uses it) on flat buffers.

    fuse_zero_grad (default True): step() also clears the gradient buffer in the same pass, i.e. it behaves like
    `optimizer.step(); optimizer.zero_grad()` -- the state the next backward's in-place write needs -- and zero_grad()
    becomes free.  With False the gradients stay readable after step() and zero_grad() is one fill launch.
    process_group / exchange / exchange_opts: data-parallel (one process per GPU): gradients are summed over the ranks
    inside step() -- "peer": sum + Adam in one launch through IPC-mapped peer memory, "rccl" / "torch": one all-reduce
    of the flat buffer, then Adam with grad_scale = 1 / world (DDP's mean); None: decided collectively (dist.py)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, fuse_zero_grad=True, process_group=None,
                 exchange=None, exchange_opts=None):
        # (torch.optim.Adam's remaining group keys ride along with the values this optimizer implements, so that a state
        #  dict saved here loads into torch.optim.Adam and steps there)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None))
        if len(self.param_groups) != 1:
            raise ValueError("DtaAdam keeps one parameter group (one learning rate), as the reference's optimizers do")
        plist = [p for p in self.param_groups[0]["params"] if p.requires_grad]
        flat = [p for p in plist if p.is_cuda and p.dtype == torch.float32]
        if not flat:
            raise RuntimeError("DtaAdam needs parameters on a ROCm device (model.cuda() first); there is no CPU path")
        dev = flat[0].device
        f64 = [p for p in plist if p.is_cuda and p.dtype == torch.float64 and p.numel() == 1]
        self._alpha = f64[0] if f64 else None          # Hang2020.alpha: stepped in float64 inside the same launch
        known = {id(p) for p in flat} | ({id(self._alpha)} if self._alpha is not None else set())
        self._other = [p for p in plist if id(p) not in known]      # anything else: plain torch math (tiny, rare)
        self.fuse_zero_grad = bool(fuse_zero_grad)
        self.device = dev
        self.pg = process_group
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
        # segments: parameters without a gate first, then one segment per (ensemble, year)
        groups = {}
        for p in flat:
            gate = H._PARAM_GATE.get(id(p))
            ens = gate[0]() if gate is not None else None
            key = None if ens is None else (id(ens), gate[1])
            groups.setdefault(key, (ens, gate[1] if gate else None, []))[2].append(p)
        order = ([None] if (None in groups or self._alpha is not None) else []) + sorted(k for k in groups if k is not None)
        self._segs = []                 # (ensemble weakref or None, year, offset, length)
        self._offs = {}
        self._alpha_slot = -1
        off = 0
        for key in order:
            ens, year, ps = groups.get(key, (None, None, []))
            start = off
            for p in ps:
                self._offs[id(p)] = off
                off = _round4(off + p.numel())
            if key is None and self._alpha is not None:
                self._alpha_slot = off  # data-parallel: alpha's float64 gradient crosses the ranks in this fp32 slot
                off = _round4(off + 1)
            self._segs.append((None if ens is None else weakref.ref(ens), year, start, off - start))
        n_gated = sum(1 for s in self._segs if s[0] is not None)
        self._flag_off = off            # data-parallel: the gated segments' 0/1 flags ride in the exchange
        n = _round4(off + n_gated)
        self.n = n
        self.exchange, self.ex, self.rccl = None, None, None
        if self.world > 1:
            from .dist import PeerExchange, RcclDirect, choose_exchange
            self.exchange = choose_exchange(exchange, process_group)
            if self.exchange == "peer":
                self.ex = PeerExchange(n, process_group, **(exchange_opts or {}))
            elif self.exchange == "rccl":
                self.rccl = RcclDirect(process_group)
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = self.ex.grad if self.ex is not None else torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self._gview = {}
        self._me = weakref.ref(self)
        with torch.no_grad():
            for p in flat:
                o, k = self._offs[id(p)], p.numel()
                self.flat_p[o:o + k].copy_(p.reshape(-1))
                p.data = self.flat_p[o:o + k].view(p.shape)
                self._gview[id(p)] = self.flat_g[o:o + k].view(p.shape)
                H._GRAD_SINKS[id(p)] = self._me
                self.state[p] = {"step": torch.zeros((), dtype=torch.float32), "exp_avg": self.flat_m[o:o + k].view(p.shape),
                                 "exp_avg_sq": self.flat_v[o:o + k].view(p.shape)}
        if self._alpha is not None:
            a = self._alpha
            self._alpha_g = torch.zeros((), dtype=torch.float64, device=dev)
            self._alpha_m = torch.zeros((), dtype=torch.float64, device=dev)
            self._alpha_v = torch.zeros((), dtype=torch.float64, device=dev)
            self._gview[id(a)] = self._alpha_g
            H._GRAD_SINKS[id(a)] = self._me
            self.state[a] = {"step": torch.zeros((), dtype=torch.float32), "exp_avg": self._alpha_m, "exp_avg_sq": self._alpha_v}
        self._flat_params = flat
        self._written = set()           # ids of parameters whose gradient was written in place since the last clear
        self._accum = set()             # ids of parameters autograd accumulated into (their .grad view) since the last clear
        self._sets = {}                 # take_inplace: caller key -> (parameters, ids, views)
        self._loose = None
        self._clean = True              # the whole gradient buffer is KNOWN to hold zeros
        # gradients that reach a .grad view through autograd's own accumulation (small torch modules sharing the optimizer,
        # a year ensemble on its host-decided path, a second backward before step()) leave no trace in take_inplace: a
        # post-accumulate hook per parameter records them, so zero_grad() / step() know the buffer is dirty
        me = self._me
        self._hooks = []

        def _dirty(param, me=me):
            o = me()
            if o is not None:
                o._accum.add(id(param))
                o._clean = False
        for p in flat + ([self._alpha] if self._alpha is not None else []):
            self._hooks.append(p.register_post_accumulate_grad_hook(_dirty))
        self.layout_epoch = 0
        self._steps = 0
        self.dev_steps = torch.zeros(2, max(1, n_gated), dtype=torch.int32, device=dev)      # per gated segment, two banks
        self._bank = 0
        self._zero_flags = torch.zeros(max(1, n_gated), dtype=torch.float32, device=dev)
        self._attach()
        H._SINK_EPOCH[0] += 1
        if self.world > 1:
            for t in [self.flat_p] + ([self._alpha.data] if self._alpha is not None else []):
                torch.distributed.broadcast(t, 0, group=process_group)      # DDP start-up: rank 0's parameters

    # ---- gradient views ------------------------------------------------------------------------------------
    def _attach(self, everything=True):
        """(Re-)install the .grad views.  Parameters whose gradients arrive through a network's in-place backward are
        re-attached lazily by take_inplace; the others (small torch modules sharing the optimizer, e.g. the site MLP of the
        metadata model: autograd accumulates INTO a .grad that exists, but would install its own tensor over a None)
        are checked on every zero_grad."""
        if everything or self._loose is None or self._loose[0] != len(self._sets):
            taken = set()
            for params, ids, views in self._sets.values():
                taken.update(ids)
            allp = self._flat_params + ([self._alpha] if self._alpha is not None else [])
            self._loose = (len(self._sets), [p for p in allp if id(p) not in taken])
            todo = allp if everything else self._loose[1]
        else:
            todo = self._loose[1]
        for p in todo:
            if p.grad is not self._gview[id(p)]:
                p.grad = self._gview[id(p)]

    def take_inplace(self, key, build):
        """Called by a network's backward: may it write its parameters' gradients straight into their .grad views?
        key: identifies the caller's parameter set (cached here); build(): the list of those Parameters.
        Yes when each still IS this optimizer's view and none has been written since the buffer was cleared (a second
        backward before step() -- gradient accumulation -- goes through accumulation instead)."""
        hit = self._sets.get(key)
        if hit is None:
            params = build()
            hit = self._sets[key] = (params, [id(p) for p in params], [self._gview.get(id(p)) for p in params])
        params, ids, views = hit
        written = self._written
        if written and not written.isdisjoint(ids):
            return False
        for p, v in zip(params, views):
            g = p.grad
            if g is not v:
                if g is not None or v is None:
                    return False            # somebody installed their own gradient tensor: leave it to autograd
                p.grad = v                  # set to None by a module.zero_grad(): same as cleared
        written.update(ids)
        self._clean = False
        return True

    def zero_grad(self, set_to_none=False):
        """Clears the flat gradient buffer (free when it is known to be clean: right after a step() with fuse_zero_grad, or
        after a zero_grad()) and keeps the .grad views in place whatever `set_to_none` says -- they are what lets a backward
        write gradients without copies.  Anything written since -- in place by a network's backward (take_inplace) or by
        autograd's accumulation (the post-accumulate hooks) -- makes the buffer dirty, and a dirty buffer is cleared."""
        self._attach(everything=False)
        if not self._clean or self._written or self._accum:
            self.flat_g.zero_()
            if self._alpha is not None:
                self._alpha_g.zero_()
        for p in self._other:
            p.grad = None
        self._written.clear()
        self._accum.clear()
        self._clean = True

    # ---- the step ------------------------------------------------------------------------------------------
    def _gate_flags(self):
        """Device pointers (as ints) of each gated segment's 0/1 'kept' flag for THIS step: the flags of the ensemble's
        training forward(s) since the last step (year.learned_ensemble.step_flags; several forwards -- gradient
        accumulation -- count as kept when any of them kept the year).  An ensemble that ran no training forward since the
        last step is inactive (its segments hold zeros) -- unless gradients WERE written for it, which means some path
        produced gradients without publishing its years: an error, never a silently skipped update."""
        out, self._flag_keep = [], []
        k = 0
        per_ens = {}
        for ens_ref, year, off, n in self._segs:
            if ens_ref is None:
                continue
            ens = ens_ref()
            if ens is not None and id(ens) not in per_ens:
                per_ens[id(ens)] = ens.step_flags()
            fl = per_ens.get(id(ens)) if ens is not None else None
            if fl is None:
                dirty = self._written | self._accum
                if dirty and any(off <= self._offs[id(p)] < off + n for p in self._flat_params if id(p) in dirty):
                    raise RuntimeError("DtaAdam.step: gradients were written for year {} of a learned_ensemble that published no "
                                       "'year kept' flags for this step (call the ensemble's forward, not its year models, "
                                       "when its parameters belong to a DtaAdam)".format(year))
                out.append(self._zero_flags.data_ptr() + 4 * k)
            else:
                self._flag_keep.append(fl)      # (alive until the launches that read it are enqueued)
                out.append(fl.data_ptr() + 4 * year)
            k += 1
        return out

    def _flags_consumed(self):
        seen = set()
        for ens_ref, year, off, n in self._segs:
            ens = ens_ref() if ens_ref is not None else None
            if ens is not None and id(ens) not in seen:
                seen.add(id(ens))
                ens.flags_consumed()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        g = self.param_groups[0]
        if g.get("weight_decay") or g.get("amsgrad") or g.get("maximize"):
            raise ValueError("DtaAdam implements torch.optim.Adam without weight decay / amsgrad / maximize (the reference's setting)")
        lr, (b1, b2), eps = float(g["lr"]), g["betas"], float(g["eps"])
        self._steps += 1
        zero = 1 if self.fuse_zero_grad else 0
        al = self._alpha is not None
        gated = [s for s in self._segs if s[0] is not None]
        flags = self._gate_flags()
        scale = 1.0 / self.world
        if self.ex is not None:
            self.ex.check()
        if self.world > 1 and gated:
            # this rank's year flags travel with the gradients: after the sum, > 0 means "some rank kept the year"
            fl = self.flat_g[self._flag_off:self._flag_off + len(gated)]
            for k, ptr in enumerate(flags):
                src = torch.as_tensor(_FlagWord(ptr), device=self.device)
                fl[k:k + 1].copy_(src)
            flags = [fl.data_ptr() + 4 * k for k in range(len(gated))]
        if self.world > 1:
            if self.ex is not None and not gated:
                # sum over ranks + Adam (+ zero_grad) in ONE launch
                self.ex.adam_step(self.flat_p, self.flat_m, self.flat_v, self._alpha if al else None,
                                  self._alpha_g if al else None, self._alpha_slot, self._alpha_m if al else None,
                                  self._alpha_v if al else None, self._steps, lr, (b1, b2), eps, zero_grad=bool(zero))
                self._step_others(lr, b1, b2, eps)
                self._after_step(zero)
                return loss
            if self.ex is not None:
                self.ex.allreduce(self._alpha_g if al else None, self._alpha_slot)
            else:
                if al:
                    self.flat_g[self._alpha_slot:self._alpha_slot + 1].copy_(self._alpha_g.reshape(1))
                if self.rccl is not None:
                    self.rccl.all_reduce(self.flat_g)
                else:
                    torch.distributed.all_reduce(self.flat_g, group=self.pg)
        slot = C.c_void_p(self.flat_g.data_ptr() + 4 * self._alpha_slot) if (al and self.world > 1) else None
        cur, nxt = self._bank, 1 - self._bank
        k = 0
        multi = []                  # the gated segments (one per year of an ensemble): ONE launch for all of them
        for ens_ref, year, off, n in self._segs:
            base = [t.data_ptr() + 4 * off for t in (self.flat_p, self.flat_g, self.flat_m, self.flat_v)]
            if ens_ref is None:
                # the ungated segment (alpha's exchange slot is its last element)
                ptrs = [C.c_void_p(b) for b in base]
                _lib.check(L.dta_adam_step_dp(*ptrs, n, _lib.ptr(self._alpha) if al else None,
                                              _lib.ptr(self._alpha_g) if al else None, slot,
                                              _lib.ptr(self._alpha_m) if al else None, _lib.ptr(self._alpha_v) if al else None,
                                              self._steps, lr, b1, b2, eps, scale, zero, st), "dta_adam_step_dp")
            else:
                step_ptr = self.dev_steps.data_ptr() + 4 * (cur * self.dev_steps.shape[1] + k)
                next_ptr = self.dev_steps.data_ptr() + 4 * (nxt * self.dev_steps.shape[1] + k)
                multi.append(_lib.AdamSegment(base[0], base[1], base[2], base[3], n, flags[k], step_ptr, next_ptr, 0))
                k += 1
        for lo in range(0, len(multi), _lib.ADAM_MAX_SEGMENTS):
            part = multi[lo:lo + _lib.ADAM_MAX_SEGMENTS]
            arr = (_lib.AdamSegment * len(part))(*part)
            _lib.check(L.dta_adam_step_multi(len(part), arr, lr, b1, b2, eps, scale, zero, st), "dta_adam_step_multi")
        self._bank = nxt
        if self.world > 1 and gated:
            self.flat_g[self._flag_off:].zero_()          # the flag slots (outside every segment)
        self._step_others(lr, b1, b2, eps)
        self._flags_consumed()
        self._after_step(zero)
        return loss

    def _after_step(self, zero):
        if zero:            # the step's launches cleared every gradient they read
            self._written.clear()
            self._accum.clear()
            self._clean = True
        else:               # gradients stay readable: the buffer is dirty until zero_grad()
            self._clean = False

    def _step_others(self, lr, b1, b2, eps):
        for p in self._other:
            if p.grad is None and self.world == 1:
                continue
            gr = p.grad
            if self.world > 1:
                # every rank joins the collective whatever its own gradient is (a rank without one sends zeros): ranks that
                # disagreed about `grad is None` would otherwise wait for each other forever.  One extra element carries
                # "this rank produced a gradient": when NO rank did, the parameter is passed over exactly as torch.optim.Adam
                # -- and the reference under DDP -- pass over a grad-None parameter (no state, no moment decay, no move)
                buf = torch.zeros(p.numel() + 1, dtype=p.dtype, device=p.device)
                if gr is not None:
                    buf[:-1].copy_(gr.reshape(-1))
                    buf[-1] = 1
                torch.distributed.all_reduce(buf, group=self.pg)
                if float(buf[-1]) == 0.0:      # (these are the few small parameters outside the flat buffers: one scalar read each)
                    continue
                gr = (buf[:-1] / self.world).view_as(p)
            s = self.state.setdefault(p, {})
            if not s:
                s["step"], s["exp_avg"], s["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
            s["step"] += 1
            s["exp_avg"].mul_(b1).add_(gr, alpha=1 - b1)
            s["exp_avg_sq"].mul_(b2).addcmul_(gr, gr, value=1 - b2)
            bc1, bc2 = 1 - b1 ** s["step"], 1 - b2 ** s["step"]
            p.addcdiv_(s["exp_avg"], (s["exp_avg_sq"] / bc2).sqrt_().add_(eps), value=-lr / bc1)

    def step_counts(self):
        """Optimizer steps taken per segment (gated segments: read back from the device counters, one host sync)."""
        dev = self.dev_steps[self._bank].tolist()
        out, k = [], 0
        for ens_ref, year, off, n in self._segs:
            if ens_ref is None:
                out.append(self._steps)
            else:
                out.append(int(dev[k]))
                k += 1
        return out

    def state_dict(self):
        counts = self.step_counts()
        for (ens_ref, year, off, n), c in zip(self._segs, counts):
            for p in self._flat_params:
                if off <= self._offs[id(p)] < off + n:
                    self.state[p]["step"] = torch.tensor(float(c))
        if self._alpha is not None:
            self.state[self._alpha]["step"] = torch.tensor(float(self._steps))
        return super().state_dict()

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Resume (Lightning restores optimizer states from its checkpoints; a state_dict of torch.optim.Adam over the same
        parameter list has the same layout and loads too).  torch's loader replaces the state tensors by copies: their
        values go back into the flat moment buffers the kernel reads, the views are re-installed, and the step counts
        return to the host counter (ungated parameters, alpha) and the per-year device counters."""
        super().load_state_dict(state_dict)
        steps = []
        for p in self._flat_params:
            o, k = self._offs[id(p)], p.numel()
            st = self.state.get(p)
            m, v = self.flat_m[o:o + k].view(p.shape), self.flat_v[o:o + k].view(p.shape)
            if st and "exp_avg" in st:
                m.copy_(st["exp_avg"].to(m))
                v.copy_(st["exp_avg_sq"].to(v))
                steps.append((o, int(float(st.get("step", 0)))))
            else:       # torch leaves no entry for a parameter that was never stepped (an always-missing year)
                m.zero_(); v.zero_()
                steps.append((o, 0))
                st = self.state[p] = {}
            st["step"] = torch.tensor(float(steps[-1][1]))
            st["exp_avg"], st["exp_avg_sq"] = m, v
        if self._alpha is not None:
            st = self.state.get(self._alpha) or {}
            if "exp_avg" in st:
                self._alpha_m.copy_(st["exp_avg"].to(self._alpha_m))
                self._alpha_v.copy_(st["exp_avg_sq"].to(self._alpha_v))
            else:
                self._alpha_m.zero_(); self._alpha_v.zero_()
            self.state[self._alpha] = {"step": torch.tensor(float(st.get("step", 0))), "exp_avg": self._alpha_m, "exp_avg_sq": self._alpha_v}
        # one count per segment (all parameters of a segment were stepped together)
        seg_steps = []
        for ens_ref, year, off, n in self._segs:
            # (a parameter torch never stepped -- grad None every time, e.g. the unused heads -- has count 0 and zero
            #  moments: stepping it with the segment's count and its all-zero gradient leaves it where it is)
            c = sorted({s for o, s in steps if off <= o < off + n and s > 0})
            if len(c) > 1:
                raise ValueError("DtaAdam.load_state_dict: parameters of one segment carry different step counts")
            seg_steps.append(c[0] if c else 0)
        k = 0
        for (ens_ref, year, off, n), c in zip(self._segs, seg_steps):
            if ens_ref is None:
                self._steps = c
                if self._alpha is not None and not any(off <= o < off + n for o, _ in steps):
                    self._steps = int(float(self.state[self._alpha]["step"]))
            else:
                self.dev_steps[:, k] = c
                k += 1

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):
            raise ValueError("DtaAdam keeps one parameter group: build it over all parameters (as the reference's optimizers are)")
        super().add_param_group(param_group)

    def close(self):
        """Collective (data-parallel): release the peer exchange / RCCL communicator; every rank calls it."""
        for p in self._flat_params + ([self._alpha] if self._alpha is not None else []):
            if p.grad is self._gview.get(id(p)):
                p.grad = None
            if H._GRAD_SINKS.get(id(p)) is self._me:
                del H._GRAD_SINKS[id(p)]
        H._SINK_EPOCH[0] += 1
        self._sets = {}
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if self.ex is not None:
            self._gview = {}
            self.flat_g = None
            self.ex.close()
            self.ex = None
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None


class _FlagWord:
    """One float32 word of device memory as a tensor (year flags of an ensemble's forward)."""

    def __init__(self, ptr):
        self.__cuda_array_interface__ = {"shape": (1,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
