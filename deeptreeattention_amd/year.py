"""MI355X-native drop-in for the reference's `src/models/year.py` (year ensemble).

One `spectral_network` per year; a year is skipped only when its WHOLE batch tensor sums to zero
(reference year.py:27); the kept years' last-head scores are averaged (:30, :33).  All kept years run as the groups of
ONE set of launches through the HIP library (`dta_ensemble_forward/backward`: a third of the launches of three
separate networks; only head 3 is evaluated, as the reference discards heads 1-2), inside one autograd node."""
import ctypes as C

import torch
from torch import nn

from . import Hang2020
from . import _lib


class _EnsembleFn(torch.autograd.Function):
    """One autograd node for all kept years: forward/backward are single C-ABI calls."""

    @staticmethod
    def forward(ctx, owner, kept, *args):
        L = _lib.lib()
        n = len(kept)
        xs, params = args[:n], args[n:]
        if any(x.requires_grad for x in xs):
            raise RuntimeError("the year ensemble does not produce a gradient for its input crops: pass detached tensors")
        ctx.set_materialize_grads(False)
        nets_mod = [owner.year_models[i] for i in kept]
        names = Hang2020._subnet_param_names("spectral")
        B, bands, H, W = xs[0].shape
        m0 = nets_mod[0]
        desc = _lib.NetDesc(B, bands, H, W, m0._classes, _lib.NET_SPECTRAL, _lib.dtype_code(m0.precision),
                            1 if m0.training else 0, 4, Hang2020.BN_MOMENTUM, Hang2020.BN_EPS)
        nbytes = L.dta_ensemble_workspace_bytes(C.byref(desc), n)
        if nbytes == 0:
            raise RuntimeError("dta_ensemble_workspace_bytes: " + L.dta_last_error().decode())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xs[0].device)
        nets = (_lib.SubnetParams * n)()
        for k, mod in enumerate(nets_mod):
            tensors = {nm: params[k * len(names) + j] for j, nm in enumerate(names)}
            for Lv in (1, 2, 3):
                bn = Hang2020._get(mod, f"conv{Lv}.bn1")
                tensors[f"conv{Lv}.bn1.running_mean"] = bn.running_mean
                tensors[f"conv{Lv}.bn1.running_var"] = bn.running_var
                tensors[f"conv{Lv}.bn1.num_batches_tracked"] = bn.num_batches_tracked
            Hang2020._fill_struct(nets[k], "spectral", tensors, False)
        xptr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        out = torch.empty(B, m0._classes, dtype=torch.float32, device=xs[0].device)
        _lib.check(L.dta_ensemble_forward(C.byref(desc), n, nets, xptr, _lib.ptr(ws), _lib.ptr(out),
                                          _lib.current_stream_ptr()), "dta_ensemble_forward")
        ctx.desc, ctx.n, ctx.nets, ctx.names = desc, n, nets, names
        ctx.save_for_backward(ws, *params)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        ws, *params = ctx.saved_tensors
        n, names = ctx.n, ctx.names
        if gout is None:
            return (None, None) + (None,) * n + (None,) * len(params)
        dscore = (gout.contiguous().float() / n).contiguous()      # d(mean over years)/d(one year's scores)
        wanted = [i for i in range(len(params)) if "classifier1" not in names[i % len(names)]
                  and "classifier2" not in names[i % len(names)]]  # heads 1-2 never reach the loss: grad None
        flat = torch.zeros(sum(params[i].numel() for i in wanted), dtype=torch.float32, device=ws.device)
        grads, off = [None] * len(params), 0
        for i in wanted:
            k = params[i].numel()
            grads[i] = flat[off:off + k].view(params[i].shape)
            off += k
        gstructs = (_lib.SubnetGrads * n)()
        for k in range(n):
            gt = {nm: grads[k * len(names) + j] for j, nm in enumerate(names) if grads[k * len(names) + j] is not None}
            Hang2020._fill_struct(gstructs[k], "spectral", gt, True)
        _lib.check(L.dta_ensemble_backward(C.byref(ctx.desc), n, ctx.nets, _lib.ptr(ws), _lib.ptr(dscore), gstructs,
                                           _lib.current_stream_ptr()), "dta_ensemble_backward")
        return (None, None) + (None,) * n + tuple(grads)


class _EnsembleGatedFn(torch.autograd.Function):
    """All years as the groups of one set of launches with the missing-year decision (reference year.py:27) taken ON THE
    DEVICE: dta_year_flags -> dta_ensemble_forward_gated -> (backward) dta_ensemble_backward_gated.  No host round trip.
    A flagged-off year is left out of the mean, keeps its BatchNorm statistics and gets exact-zero gradients; the owning
    optim.DtaAdam steps each year under the same flag (a skipped year's parameters are passed over, as torch's Adam
    passes over grad None)."""

    @staticmethod
    def forward(ctx, owner, flags, clear_next, *args):
        L = _lib.lib()
        n = len(owner.year_models)
        xs, params = args[:n], args[n:]
        if any(x.requires_grad for x in xs):
            raise RuntimeError("the year ensemble does not produce a gradient for its input crops: pass detached tensors")
        ctx.set_materialize_grads(False)
        B = xs[0].shape[0]
        ctx.anchor = len(params) == 1          # anchor mode: one autograd input stands for all parameters (Hang2020._NetFn)
        if ctx.anchor:
            params = owner._plist()
        desc, nets, nbytes = owner._tables(xs[0].shape, params)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xs[0].device)
        xptr = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        out = torch.empty(B, owner.year_models[0]._classes, dtype=torch.float32, device=xs[0].device)
        kept = torch.empty(2, dtype=torch.float32, device=xs[0].device)      # {years kept, 1 / years kept}
        st = _lib.current_stream_ptr()
        _lib.check(L.dta_year_flags(xptr, n, xs[0].numel(), _lib.ptr(flags), _lib.ptr(clear_next), st), "dta_year_flags")
        _lib.check(L.dta_ensemble_forward_gated(C.byref(desc), n, nets, xptr, _lib.ptr(flags), _lib.ptr(ws), _lib.ptr(out),
                                                _lib.ptr(kept), st), "dta_ensemble_forward_gated")
        ctx.owner, ctx.desc, ctx.nets, ctx.n, ctx.flags, ctx.kept = owner, desc, nets, n, flags, kept
        if ctx.anchor:
            ctx.save_for_backward(ws)
        else:
            ctx.save_for_backward(ws, *params)
        return out

    @staticmethod
    def backward(ctx, gout):
        L = _lib.lib()
        ws, *params = ctx.saved_tensors
        owner, n = ctx.owner, ctx.n
        if ctx.anchor:
            params = owner._plist()
        nret = 1 if ctx.anchor else len(params)
        if gout is None:
            return (None, None, None) + (None,) * n + (None,) * nret
        names = Hang2020._subnet_param_names("spectral")
        dscore = gout.contiguous().float() * ctx.kept[1]        # d(mean over kept years)/d(one year's scores), on the device
        wanted = [i for i in range(len(params)) if "classifier1" not in names[i % len(names)]
                  and "classifier2" not in names[i % len(names)]]  # heads 1-2 never reach the loss: grad None
        plist = owner._plist()
        sink = Hang2020._cached_sink(owner, plist)
        if sink is not None and sink.take_inplace((id(owner), 4), lambda: [plist[i] for i in wanted]):
            cache = Hang2020._table_cache(owner)
            gkey = ("grads", id(sink), sink.layout_epoch)
            gstructs = cache.get(gkey)
            if gstructs is None:
                gstructs = (_lib.SubnetGrads * n)()
                wset = set(wanted)
                for k in range(n):
                    gt = {nm: plist[k * len(names) + j].grad for j, nm in enumerate(names) if (k * len(names) + j) in wset}
                    Hang2020._fill_struct(gstructs[k], "spectral", gt, True)
                cache[gkey] = gstructs
            _lib.check(L.dta_ensemble_backward_gated(C.byref(ctx.desc), n, ctx.nets, _lib.ptr(ws), _lib.ptr(dscore), gstructs,
                                                     _lib.ptr(ctx.flags), 3, _lib.current_stream_ptr()), "dta_ensemble_backward")
            return (None, None, None) + (None,) * n + (None,) * nret
        flat = torch.zeros(sum(params[i].numel() for i in wanted), dtype=torch.float32, device=ws.device)
        grads, off = [None] * len(params), 0
        for i in wanted:
            k = params[i].numel()
            grads[i] = flat[off:off + k].view(params[i].shape)
            off += k
        gstructs = (_lib.SubnetGrads * n)()
        for k in range(n):
            gt = {nm: grads[k * len(names) + j] for j, nm in enumerate(names) if grads[k * len(names) + j] is not None}
            Hang2020._fill_struct(gstructs[k], "spectral", gt, True)
        _lib.check(L.dta_ensemble_backward_gated(C.byref(ctx.desc), n, ctx.nets, _lib.ptr(ws), _lib.ptr(dscore), gstructs,
                                                 _lib.ptr(ctx.flags), 3, _lib.current_stream_ptr()), "dta_ensemble_backward")
        if ctx.anchor:      # a second backward before step(): accumulate here, as autograd would
            with torch.no_grad():
                for p, g in zip(plist, grads):
                    if g is None:
                        continue
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.add_(g)
            return (None, None, None) + (None,) * n + (None,)
        return (None, None, None) + (None,) * n + tuple(grads)


class learned_ensemble(nn.Module):
    def __init__(self, years, classes, config):
        super().__init__()
        self.year_models = nn.ModuleList()
        self.years = years
        for _ in range(years):
            if config["pretrain_state_dict"]:
                base_model = Hang2020.load_from_backbone(state_dict=config["pretrain_state_dict"], classes=classes,
                                                         bands=config["bands"])
            else:
                base_model = Hang2020.spectral_network(bands=config["bands"], classes=classes)
            self.year_models.append(base_model)
        self._register_gates()
        # device-side "year kept" flags (each a (Y,) float32 tensor of 0 / 1):
        #   _pending_flags: one tensor per TRAINING forward since the owning optimizer's last step -- the forward's own
        #     tensor (its backward reads exactly that one, whatever other forwards run in between); optim.DtaAdam steps a year
        #     when ANY of them kept it (gradient accumulation) and then empties the list;
        #   local_flags: the newest of them (None before the first training forward);
        #   _scratch_flags: two banks used alternately by forwards that build no graph (validation, inference) -- never
        #     published, so a no_grad forward between a training forward and its backward / step disturbs nothing.
        self.__dict__["local_flags"] = None
        self.__dict__["_pending_flags"] = []
        self.__dict__["_scratch_flags"] = None

    def _register_gates(self):
        """An optimizer built from .parameters() (optim.DtaAdam) finds each year's parameters through this registry and
        steps them under that year's device-side "kept" flag.  Keyed by Parameter OBJECT: repeated wherever the objects
        can change -- construction, copy.deepcopy / unpickling (__setstate__), load_state_dict(assign=True) (_plist)."""
        import weakref
        me = weakref.ref(self)
        for y, m in enumerate(self.year_models):
            for p in m.parameters():
                Hang2020._PARAM_GATE[id(p)] = (me, y)

    def __setstate__(self, state):
        super().__setstate__(state)
        # a copy starts with no training forward pending, its own scratch banks, and ITS parameters in the registry
        self.__dict__["local_flags"] = None
        self.__dict__["_pending_flags"] = []
        self.__dict__["_scratch_flags"] = None
        self._register_gates()

    def _tables(self, shape, params):
        """Cached (descriptor, parameter pointer tables, workspace bytes) of the grouped launch over ALL years."""
        L = _lib.lib()
        mods = list(self.year_models)
        cache = Hang2020._table_cache(self)
        bufs = [b for m in mods for b in Hang2020._bn_buffers(m, Hang2020._table_cache(m))]
        fp = tuple(t.data_ptr() for t in params) + tuple(t.data_ptr() for t in bufs)
        m0 = mods[0]
        key = (tuple(shape), m0.precision, m0.training)
        hit = cache.get(key)
        if hit is None or hit[0] != fp:
            n = len(mods)
            names = Hang2020._subnet_param_names("spectral")
            B, bands, H, W = shape
            desc = _lib.NetDesc(B, bands, H, W, m0._classes, _lib.NET_SPECTRAL, _lib.dtype_code(m0.precision),
                                1 if m0.training else 0, 4, Hang2020.BN_MOMENTUM, Hang2020.BN_EPS)
            nets = (_lib.SubnetParams * n)()
            for k in range(n):
                tensors = {nm: params[k * len(names) + j] for j, nm in enumerate(names)}
                for Lv in (1, 2, 3):
                    tensors[f"conv{Lv}.bn1.running_mean"] = bufs[9 * k + 3 * (Lv - 1)]
                    tensors[f"conv{Lv}.bn1.running_var"] = bufs[9 * k + 3 * (Lv - 1) + 1]
                    tensors[f"conv{Lv}.bn1.num_batches_tracked"] = bufs[9 * k + 3 * (Lv - 1) + 2]
                Hang2020._fill_struct(nets[k], "spectral", tensors, False)
            if len(cache) > 8:
                for k in [k for k in cache if k != "bufs"][:4]:
                    cache.pop(k)
            hit = cache[key] = (fp, desc, nets)
        nbytes = L.dta_ensemble_workspace_bytes(C.byref(hit[1]), len(mods))
        if nbytes == 0:
            raise RuntimeError("dta_ensemble_workspace_bytes: " + L.dta_last_error().decode())
        return hit[1], hit[2], nbytes

    def _plist(self):
        """All years' parameters in struct order (cached until a load_state_dict may have replaced Parameter objects)."""
        cache = Hang2020._table_cache(self)
        epoch = tuple(m.__dict__.get("_dta_epoch", 0) for m in self.year_models)
        hit = cache.get("plist")
        if hit is None or hit[0] != epoch:
            hit = cache["plist"] = (epoch, [p for m in self.year_models for p in m._param_list()])
            self._register_gates()
        return hit[1]

    def _next_flags(self, dev, publish, own=None):
        """(flags, clear_next) for one dta_year_flags call.  A training forward (publish) gets a fresh tensor of its own
        (clear_next None: the library zero-fills it in the same call) and queues it for the optimizer; a forward without a
        graph alternates two private banks (each call clears the other one: no clearing launch)."""
        Y = len(self.year_models)
        if publish or own:      # own: a graph-building forward keeps a tensor of its own for its backward, published or not
            flags = torch.empty(Y, dtype=torch.float32, device=dev)
            if publish:
                self._publish_flags(flags)
            return flags, None
        fs = self.__dict__["_scratch_flags"]
        if fs is None or fs[0].device != dev:
            fs = self.__dict__["_scratch_flags"] = [torch.zeros(2, Y, dtype=torch.float32, device=dev), 0]
        fs[1] ^= 1
        return fs[0][fs[1]], fs[0][fs[1] ^ 1]

    def _publish_flags(self, flags):
        pend = self.__dict__["_pending_flags"]
        if len(pend) >= 64:       # a long accumulation, or nobody consumes them (a stock torch optimizer): fold them
            pend[:] = [torch.stack(pend).amax(0)]
        pend.append(flags)
        self.__dict__["local_flags"] = flags

    def step_flags(self):
        """The flags the owning optimizer gates THIS step by: a year is stepped when any training forward since the last
        step kept it (one forward: its own tensor, no launch; several -- gradient accumulation --: their maximum).  None:
        no training forward since the last step."""
        pend = self.__dict__["_pending_flags"]
        if not pend:
            return None
        if len(pend) == 1:
            return pend[0]
        return torch.stack(pend).amax(0)

    def flags_consumed(self):
        """Called by the optimizer after its step: the next step starts a new set of training forwards."""
        self.__dict__["_pending_flags"] = []

    def forward(self, images):
        if len(images) != len(self.year_models):
            raise ValueError("expected one image tensor per year ({}), got {}".format(len(self.year_models), len(images)))
        params = self._plist()
        train_graph = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        gated = len(self.year_models) <= _lib.MAX_YEARS and (not train_graph or Hang2020._cached_sink(self, params) is not None)
        if gated:
            # missing years decided on the device (no host round trip): inference / validation always; training when
            # the parameters belong to an optim.DtaAdam, which steps each year under the same device-side flag
            xs = [Hang2020._check_input(x) for x in images]
            if any(x.shape != xs[0].shape for x in xs):
                raise ValueError("all years of a batch must have the same shape")
            # flags are published for the optimizer only by TRAINING-mode forwards that build a graph: an eval-mode forward
            # with grad enabled (a saliency map, a validation loop that forgot no_grad) must not mark its years as kept for
            # the next DtaAdam.step() -- that year would be stepped on zero gradients (moment decay + a momentum move),
            # where the reference passes over it
            flags, other = self._next_flags(xs[0].device, publish=train_graph and self.training, own=train_graph)
            if train_graph:      # (the parameters belong to a DtaAdam: one anchor input instead of 123 parameter inputs)
                anchor = next(p for p in params if p.requires_grad)
                return _EnsembleGatedFn.apply(self, flags, other, *xs, anchor)
            return _EnsembleGatedFn.apply(self, flags, other, *xs, *params)
        # stock torch optimizers: same test as the reference (year.py:27: a year is skipped iff its whole batch tensor
        # sums to zero) with all years' sums travelling to the host in ONE transfer, so that a skipped year's parameters
        # really have grad None (torch's Adam passes over them)
        keep = (torch.stack([x.sum() for x in images]) != 0).tolist()
        kept = [i for i, k in enumerate(keep) if k]
        if train_graph and Hang2020._any_sink(params):
            # an optim.DtaAdam owns (some of) these parameters but the device-gated launch cannot be used (more than
            # DTA_MAX_YEARS years, or parameters outside the optimizer): its per-year gated passes still need this step's
            # decision -- the host's, sent to the device (this path has already synchronised)
            self._publish_flags(torch.tensor([1.0 if k else 0.0 for k in keep], dtype=torch.float32, device=images[0].device))
        if not kept:
            raise RuntimeError("every year of the batch is all-zero: nothing to average (reference year.py:33)")
        out = None
        for lo in range(0, len(kept), _lib.MAX_YEARS):      # DTA_MAX_YEARS networks per grouped launch
            part = kept[lo:lo + _lib.MAX_YEARS]
            xs = [Hang2020._check_input(images[i]) for i in part]
            params = [p for i in part for p in self.year_models[i]._param_list()]
            s = _EnsembleFn.apply(self, part, *xs, *params) * (len(part) / len(kept))
            out = s if out is None else out + s
        return out
