"""Stand-alone forwards of the reference's building blocks (conv_module, spectral_attention, spatial_attention,
Classifier) through the module-level C-ABI entry points (include/dta_hip.h, capi_modules.hip).  Same kernels as the
network-level path; torch is used only for the NCHW<->NHWC views the attention entry points take."""
import ctypes as C

import torch

from . import _lib
from . import Hang2020 as H


def _zeros_like_flat(tensors):
    flat = torch.zeros(sum(t.numel() for t in tensors), dtype=torch.float32, device=tensors[0].device)
    out, off = [], 0
    for t in tensors:
        out.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    return out


class _ConvModuleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, pool, precision, x, conv_w, conv_b, bn_w, bn_b):
        L = _lib.lib()
        B, Cin, Hh, Ww = x.shape
        N = conv_w.shape[0]
        desc = _lib.ConvModuleDesc(B, Cin, N, Hh, Ww, 1 if pool else 0, 1 if mod.training else 0,
                                   _lib.dtype_code(precision), H.BN_MOMENTUM, H.BN_EPS)
        nbytes = L.dta_conv_module_workspace_bytes(C.byref(desc))
        if nbytes == 0:
            raise RuntimeError("conv_module: " + L.dta_last_error().decode())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        Ho, Wo = (Hh // 2, Ww // 2) if pool else (Hh, Ww)
        out = torch.empty(B, N, Ho, Wo, dtype=torch.float32, device=x.device)
        bn = mod.bn1
        _lib.check(L.dta_conv_module_forward(C.byref(desc), _lib.ptr(conv_w), _lib.ptr(conv_b), _lib.ptr(bn_w),
                                             _lib.ptr(bn_b), _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var),
                                             _lib.ptr(bn.num_batches_tracked), _lib.ptr(x), _lib.ptr(ws), _lib.ptr(out),
                                             _lib.current_stream_ptr()), "dta_conv_module_forward")
        ctx.desc, ctx.need_dx = desc, x.requires_grad
        ctx.save_for_backward(ws, conv_w, conv_b, bn_w, bn_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        ws, conv_w, conv_b, bn_w, bn_b = ctx.saved_tensors
        d = ctx.desc
        g_w, g_b, g_bw, g_bb = _zeros_like_flat([conv_w, conv_b, bn_w, bn_b])
        dx_nhwc = None
        if ctx.need_dx:
            dx_nhwc = torch.empty(d.batch, d.height * d.width, d.in_channels, dtype=torch.float32, device=ws.device)
        _lib.check(L.dta_conv_module_backward(C.byref(d), _lib.ptr(conv_w), _lib.ptr(bn_w), _lib.ptr(ws),
                                              _lib.ptr(dout.contiguous().float()), _lib.ptr(dx_nhwc), _lib.ptr(g_w),
                                              _lib.ptr(g_b), _lib.ptr(g_bw), _lib.ptr(g_bb), _lib.current_stream_ptr()),
                   "dta_conv_module_backward")
        dx = None
        if dx_nhwc is not None:
            dx = dx_nhwc.view(d.batch, d.height, d.width, d.in_channels).permute(0, 3, 1, 2).contiguous()
        return None, None, None, dx, g_w, g_b, g_bw, g_bb


def conv_module_forward(mod, x, pool):
    x = H._check_input(x)
    if pool and not mod.maxpool_kernal:
        raise AttributeError("conv_module has no max_pool (constructed without maxpool_kernel)")   # as the reference
    if pool and tuple(mod.maxpool_kernal) != (2, 2):
        raise NotImplementedError("only the reference's 2x2 max-pool is implemented")
    precision = getattr(mod, "precision", None) or H.get_default_precision()
    return _ConvModuleFn.apply(mod, bool(pool), precision, x, mod.conv_layer.weight, mod.conv_layer.bias,
                               mod.bn1.weight, mod.bn1.bias)


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, x, *params):
        L = _lib.lib()
        B, Cc, Hh, Ww = x.shape
        desc = _lib.AttentionDesc(B, Cc, Hh, Ww, 0 if kind == "spectral" else 1)
        nbytes = L.dta_attention_workspace_bytes(C.byref(desc))
        if nbytes == 0:
            raise RuntimeError(f"{kind}_attention: " + L.dta_last_error().decode())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        x_nhwc = x.permute(0, 2, 3, 1).contiguous()
        arr = _lib.PtrArray6()
        for i, p in enumerate(params):
            arr[i] = p.data_ptr()
        if kind == "spectral":
            F = Cc
        else:
            ps = {32: 4, 64: 2, 128: 1}[Cc]
            F = Cc * (Hh // ps) * (Ww // ps)
        out = torch.empty(B, Cc, Hh, Ww, dtype=torch.float32, device=x.device)
        feat = torch.empty(B, F, dtype=torch.float32, device=x.device)
        _lib.check(L.dta_attention_forward(C.byref(desc), C.byref(arr), _lib.ptr(x_nhwc), _lib.ptr(ws), _lib.ptr(out),
                                           _lib.ptr(feat), _lib.current_stream_ptr()), "dta_attention_forward")
        ctx.desc, ctx.kind = desc, kind
        ctx.save_for_backward(ws, x_nhwc, *params)
        return out, feat

    @staticmethod
    def backward(ctx, dout, dfeat):
        L = _lib.lib()
        ws, x_nhwc, *params = ctx.saved_tensors
        d = ctx.desc
        grads = _zeros_like_flat(list(params))
        arr, garr = _lib.PtrArray6(), _lib.PtrArray6()
        for i, p in enumerate(params):
            arr[i] = p.data_ptr()
            garr[i] = grads[i].data_ptr()
        dx_nhwc = torch.empty_like(x_nhwc)
        dout = None if dout is None else dout.contiguous().float()
        dfeat = None if dfeat is None else dfeat.contiguous().float()
        _lib.check(L.dta_attention_backward(C.byref(d), C.byref(arr), _lib.ptr(x_nhwc), _lib.ptr(ws), _lib.ptr(dout),
                                            _lib.ptr(dfeat), _lib.ptr(dx_nhwc), C.byref(garr), _lib.current_stream_ptr()),
                   "dta_attention_backward")
        dx = dx_nhwc.view(d.batch, d.height, d.width, d.filters).permute(0, 3, 1, 2).contiguous()
        return (None, dx, *grads)


def attention_forward(mod, x, kind):
    x = H._check_input(x)
    params = [H._get(mod, n) for n in H._ATT_NAMES[kind]]
    return _AttentionFn.apply(kind, x, *params)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        L = _lib.lib()
        out = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        _lib.check(L.dta_linear_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), x.shape[0], w.shape[1], w.shape[0],
                                        _lib.ptr(out), _lib.current_stream_ptr()), "dta_linear_forward")
        ctx.save_for_backward(x, w, b)
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        x, w, b = ctx.saved_tensors
        gw, gb = _zeros_like_flat([w, b])
        dx = torch.empty_like(x)
        _lib.check(L.dta_linear_backward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(dout.contiguous().float()), x.shape[0],
                                         w.shape[1], w.shape[0], _lib.ptr(dx), _lib.ptr(gw), _lib.ptr(gb),
                                         _lib.current_stream_ptr()), "dta_linear_backward")
        return dx, gw, gb


def classifier_forward(mod, features):
    if not features.is_cuda:
        raise RuntimeError("deeptreeattention_amd runs on a ROCm device only (no CPU fallback)")
    return _LinearFn.apply(features.contiguous().float(), mod.fc1.weight, mod.fc1.bias)
