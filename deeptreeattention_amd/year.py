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

    def forward(self, images):
        # same test as the reference (year.py:27: a year is skipped iff its whole batch tensor sums to zero), but all
        # years' sums travel to the host in ONE transfer instead of one blocking comparison per year
        keep = (torch.stack([x.sum() for x in images]) != 0).tolist()
        kept = [i for i, k in enumerate(keep) if k]
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
