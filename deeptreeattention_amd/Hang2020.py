"""MI355X-native drop-in for the reference's `src/models/Hang2020.py` (Hang et al. 2020 attention CNN).

Same plugin surface (README.md:98-114 of the reference): every class takes the reference's constructor
arguments, exposes the reference's parameter/buffer names and torch shapes (so `state_dict()` /
`load_state_dict()` / `load_from_backbone` interoperate), and `forward(x)` takes a
float32 NCHW `(B, bands, H, W)` tensor.  The arithmetic is NOT torch: forward and backward run as hand-written
HIP kernels for gfx950 behind the C ABI of include/dta_hip.h (libdta_hip.so), reached through one
`torch.autograd.Function`.  The torch sub-modules created below (`nn.Conv2d`, `nn.BatchNorm2d`, ...) are
parameter holders only: they give the reference's names, shapes, dtypes and default initialisation; their own
`forward` is never called.  There is no CPU fallback: inputs must live on a ROCm device.

Reference map: conv_module Hang2020.py:14-31, vanilla_CNN :33-53, Classifier :55-66, spatial_attention :68-124,
spectral_attention :126-168, spatial_network :170-204, spectral_network :206-240, Hang2020 :242-263,
load_from_backbone :266-278.
"""
import ctypes as C

import torch
from torch import nn

from . import _lib

_DEFAULT_PRECISION = "fp32"
BN_MOMENTUM, BN_EPS = 0.1, 1e-5   # nn.BatchNorm2d defaults used by the reference


def set_default_precision(name):
    """'fp32' (exact fp32 MFMA; parity-grade) or 'bf16' (bf16 MFMA inputs, fp32 accumulate/BN/loss)."""
    global _DEFAULT_PRECISION
    _lib.dtype_code(name)
    _DEFAULT_PRECISION = str(name).lower()


def get_default_precision():
    return _DEFAULT_PRECISION


# ----------------------------------------------------------------------------------------------------
# parameter holders with the reference's names
# ----------------------------------------------------------------------------------------------------
class conv_module(nn.Module):
    """3x3 'same' conv + BatchNorm2d + ReLU (+ optional 2x2 max-pool) -- reference Hang2020.py:14-31."""

    def __init__(self, in_channels, filters, maxpool_kernel=None):
        super().__init__()
        self.conv_layer = nn.Conv2d(in_channels, out_channels=filters, kernel_size=(3, 3), padding="same")
        self.bn1 = nn.BatchNorm2d(filters)
        self.maxpool_kernal = maxpool_kernel
        if maxpool_kernel:
            self.max_pool = nn.MaxPool2d(maxpool_kernel)

    def forward(self, x, pool=False):
        from .modules import conv_module_forward
        return conv_module_forward(self, x, pool)


class Classifier(nn.Module):
    """Linear head -- reference Hang2020.py:55-66."""

    def __init__(self, in_features, classes):
        super().__init__()
        self.fc1 = nn.Linear(in_features=in_features, out_features=classes)

    def forward(self, features):
        from .modules import classifier_forward
        return classifier_forward(self, features)


_SPATIAL_K = {32: 7, 64: 5, 128: 3}
_SPATIAL_POOL = {32: (4, 128), 64: (2, 256), 128: (1, 512)}
_SPECTRAL_K = {32: 3, 64: 5, 128: 7}


class spatial_attention(nn.Module):
    """Per-pixel sigmoid gate from a 1x1 channel pool and two kxk convs -- reference Hang2020.py:68-124."""

    def __init__(self, filters):
        super().__init__()
        if filters not in _SPATIAL_K:
            raise ValueError("Unknown incoming kernel size {} for attention layers".format(filters))
        k = _SPATIAL_K[filters]
        self.channel_pool = nn.Conv2d(in_channels=filters, out_channels=1, kernel_size=1)
        self.attention_conv1 = nn.Conv2d(1, 1, kernel_size=k, padding="same")
        self.attention_conv2 = nn.Conv2d(1, 1, kernel_size=k, padding="same")
        pool, _ = _SPATIAL_POOL[filters]
        self.class_pool = nn.MaxPool2d((pool, pool))

    def forward(self, x):
        from .modules import attention_forward
        return attention_forward(self, x, "spatial")


class spectral_attention(nn.Module):
    """Per-channel sigmoid gate from the pooled spectrum through two Conv1d -- reference Hang2020.py:126-168."""

    def __init__(self, filters):
        super().__init__()
        if filters not in _SPECTRAL_K:
            raise ValueError("Unknown incoming kernel size {} for attention layers".format(filters))
        k = _SPECTRAL_K[filters]
        self.attention_conv1 = nn.Conv1d(filters, filters, kernel_size=k, padding="same")
        self.attention_conv2 = nn.Conv1d(filters, filters, kernel_size=k, padding="same")

    def forward(self, x):
        from .modules import attention_forward
        return attention_forward(self, x, "spectral")


_ATT_NAMES = {
    "spectral": ("attention_conv1.weight", "attention_conv1.bias", "attention_conv2.weight", "attention_conv2.bias"),
    "spatial": ("channel_pool.weight", "channel_pool.bias", "attention_conv1.weight", "attention_conv1.bias",
                "attention_conv2.weight", "attention_conv2.bias"),
}


def _subnet_param_names(kind):
    """Canonical (struct) order of one sub-network's trainable tensors, relative to the sub-network module."""
    names = []
    for L in (1, 2, 3):
        names += [f"conv{L}.conv_layer.weight", f"conv{L}.conv_layer.bias", f"conv{L}.bn1.weight", f"conv{L}.bn1.bias"]
        if kind in _ATT_NAMES:
            names += [f"attention_{L}.{n}" for n in _ATT_NAMES[kind]]
            names += [f"classifier{L}.fc1.weight", f"classifier{L}.fc1.bias"]
    if kind == "vanilla":
        names += ["fc1.weight", "fc1.bias"]
    return names


def _get(module, dotted):
    obj = module
    for part in dotted.split("."):
        obj = getattr(obj, part)
    return obj


def _fill_struct(struct, kind, tensors, is_grad):
    """tensors: dict relative-name -> tensor (or None).  Fills a SubnetParams / SubnetGrads."""
    def p(name):
        t = tensors.get(name)
        return None if t is None else t.data_ptr()
    for i, L in enumerate((1, 2, 3)):
        struct.conv_w[i] = p(f"conv{L}.conv_layer.weight")
        struct.conv_b[i] = p(f"conv{L}.conv_layer.bias")
        struct.bn_w[i] = p(f"conv{L}.bn1.weight")
        struct.bn_b[i] = p(f"conv{L}.bn1.bias")
        if not is_grad:
            struct.bn_rm[i] = p(f"conv{L}.bn1.running_mean")
            struct.bn_rv[i] = p(f"conv{L}.bn1.running_var")
            struct.bn_nbt[i] = p(f"conv{L}.bn1.num_batches_tracked")
        if kind in _ATT_NAMES:
            for j, n in enumerate(_ATT_NAMES[kind]):
                struct.att[i][j] = p(f"attention_{L}.{n}")
            struct.fc_w[i] = p(f"classifier{L}.fc1.weight")
            struct.fc_b[i] = p(f"classifier{L}.fc1.bias")
    if kind == "vanilla":
        struct.fc_w[2] = p("fc1.weight")
        struct.fc_b[2] = p("fc1.bias")


_KIND_CODE = {"spectral": _lib.NET_SPECTRAL, "spatial": _lib.NET_SPATIAL, "vanilla": _lib.NET_VANILLA}


def _check_input(x):
    if not isinstance(x, torch.Tensor) or x.dim() != 4:
        raise ValueError("expected a (B, bands, H, W) tensor")
    if not x.is_cuda:
        raise RuntimeError("deeptreeattention_amd runs on a ROCm device only (no CPU fallback): move the input and "
                           "the module to 'cuda'")
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous()


# ----------------------------------------------------------------------------------------------------
# module-level (plugin) path: cached pointer tables and in-place gradient sinks
# ----------------------------------------------------------------------------------------------------
# optim.DtaAdam registers itself here for every parameter it owns (id(param) -> weakref of the optimizer): a backward
# whose parameters all belong to ONE such optimizer writes its gradients straight into the optimizer's flat gradient
# buffer (the parameters' .grad are views of it) instead of returning 59 fresh tensors for autograd to copy one by one.
_GRAD_SINKS = {}
# year.learned_ensemble registers (weakref of the ensemble, year index) for every parameter of its year models, so that
# the optimizer can step each year's parameters under that year's device-side "kept" flag (reference year.py:27-28: a
# skipped year's parameters have grad None and torch's Adam passes over them)
_PARAM_GATE = {}
_TABLES = None      # module -> {key: cached ctypes tables}; weakly keyed, nothing is stored on the module itself


_SINK_EPOCH = [0]      # bumped whenever an optimizer registers / unregisters parameters in _GRAD_SINKS


def _cached_plist(owner):
    """owner._param_list() (a walk over ~70 dotted names), cached per module until a load_state_dict may have replaced
    Parameter objects (_dta_epoch)."""
    cache = _table_cache(owner)
    epoch = owner.__dict__.get("_dta_epoch", 0)
    hit = cache.get("plist")
    if hit is None or hit[0] != epoch:
        hit = cache["plist"] = (epoch, owner._param_list())
    return hit[1]


def _cached_sink(owner, plist):
    cache = _table_cache(owner)
    hit = cache.get("sink")
    if hit is None or hit[0] != _SINK_EPOCH[0] or hit[1] is not plist:
        s = _sink_for(plist)
        import weakref
        hit = cache["sink"] = (_SINK_EPOCH[0], plist, None if s is None else weakref.ref(s))
    return None if hit[2] is None else hit[2]()


def _wanted(owner, used_heads):
    """Indices (into the parameter list) of the tensors a backward with these heads produces gradients for: heads unused by
    the loss keep grad None, as in torch."""
    cache = _table_cache(owner)
    key = ("wanted", used_heads)
    hit = cache.get(key)
    if hit is None:
        hit, q = [], (1 if owner._net_code == _lib.NET_HANG2020 else 0)
        for kind, mod, names in owner._subnets():
            for j, n in enumerate(names):
                head = int(n.split("classifier")[1][0]) - 1 if "classifier" in n else None
                if head is None or (used_heads & (1 << head)):
                    hit.append(q + j)
            q += len(names)
        cache[key] = hit
    return hit


def _sink_for(plist):
    """The optimizer (optim.DtaAdam) that owns ALL of `plist` and can take a backward's gradients in place, or None."""
    if not plist:
        return None
    ref = _GRAD_SINKS.get(id(plist[0]))
    if ref is None:
        return None
    for p in plist:
        if _GRAD_SINKS.get(id(p)) is not ref:
            return None
    return ref()


def _any_sink(plist):
    """True when at least one of `plist` belongs to an optim.DtaAdam (which then steps it from its flat buffers)."""
    return any(id(p) in _GRAD_SINKS for p in plist)


def _table_cache(module):
    global _TABLES
    if _TABLES is None:
        import weakref
        _TABLES = weakref.WeakKeyDictionary()
    c = _TABLES.get(module)
    if c is None:
        c = _TABLES[module] = {}
    return c


def _bn_buffers(owner, cache):
    """[running_mean, running_var, num_batches_tracked] x 3 layers x sub-networks, re-collected after a load_state_dict."""
    epoch = owner.__dict__.get("_dta_epoch", 0)
    hit = cache.get("bufs")
    if hit is None or hit[0] != epoch:
        lst = []
        for kind, mod, names in owner._subnets():
            for Lv in (1, 2, 3):
                bn = _get(mod, f"conv{Lv}.bn1")
                lst += [bn.running_mean, bn.running_var, bn.num_batches_tracked]
        hit = cache["bufs"] = (epoch, lst)
    return hit[1]


def _param_tables(owner, params, shape, heads_mask):
    """(desc, nets, workspace bytes) for this call, cached per module and (shape, precision, mode, heads); the raw
    pointers inside are revalidated against the tensors' current addresses on every call (~10 us)."""
    L = _lib.lib()
    cache = _table_cache(owner)
    bufs = _bn_buffers(owner, cache)
    fp = tuple(t.data_ptr() for t in params) + tuple(t.data_ptr() for t in bufs)
    key = (tuple(shape), owner.precision, owner.training, heads_mask)
    hit = cache.get(key)
    if hit is None or hit[0] != fp:
        subnets = owner._subnets()
        B, bands, H, W = shape
        desc = _lib.NetDesc(B, bands, H, W, owner._classes, owner._net_code, _lib.dtype_code(owner.precision),
                            1 if owner.training else 0, heads_mask, BN_MOMENTUM, BN_EPS)
        nets = (_lib.SubnetParams * len(subnets))()
        pos = 1 if owner._net_code == _lib.NET_HANG2020 else 0
        b = 0
        for i, (kind, mod, names) in enumerate(subnets):
            tensors = {n: params[pos + j] for j, n in enumerate(names)}
            pos += len(names)
            for Lv in (1, 2, 3):
                tensors[f"conv{Lv}.bn1.running_mean"] = bufs[b]
                tensors[f"conv{Lv}.bn1.running_var"] = bufs[b + 1]
                tensors[f"conv{Lv}.bn1.num_batches_tracked"] = bufs[b + 2]
                b += 3
            _fill_struct(nets[i], kind, tensors, False)
        if len(cache) > 8:
            for k in [k for k in cache if k != "bufs"][:4]:
                cache.pop(k)
        hit = cache[key] = (fp, desc, nets)
    # (the workspace size is asked for on every call: it belongs to the library's launch plan, not to this cache)
    nbytes = L.dta_net_workspace_bytes(C.byref(hit[1]))
    if nbytes == 0:
        raise RuntimeError("dta_net_workspace_bytes: " + L.dta_last_error().decode())
    return hit[1], hit[2], nbytes


class _NetFn(torch.autograd.Function):
    """One autograd node for a whole network: forward/backward are single C-ABI calls."""

    @staticmethod
    def forward(ctx, owner, x, heads_mask, *params):
        L = _lib.lib()
        if x.requires_grad:
            raise RuntimeError("deeptreeattention_amd networks do not produce a gradient for their input patches "
                               "(the reference's step never asks for one): pass x.detach()")
        ctx.set_materialize_grads(False)               # unused heads arrive as None, not as zero-filled tensors
        B = x.shape[0]
        # anchor mode (_Net._run): the parameters belong to an optim.DtaAdam that takes the gradients in place, so only ONE
        # of them is an autograd input (the graph node needs an input that requires grad; 70 inputs cost ~100 us of host
        # time per step in argument handling and saved-tensor bookkeeping)
        ctx.anchor = len(params) == 1
        if ctx.anchor:
            params = _cached_plist(owner)
        desc, nets, nbytes = _param_tables(owner, params, x.shape, heads_mask)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        alpha = params[0] if owner._net_code == _lib.NET_HANG2020 else None
        joint = None
        outs = []
        table = _lib.ScoreTable()
        if owner._net_code in (_lib.NET_HANG2020, _lib.NET_VANILLA):
            joint = torch.empty(B, owner._classes, dtype=torch.float32, device=x.device)
        else:
            for Lv in range(3):
                if heads_mask & (1 << Lv):
                    t = torch.empty(B, owner._classes, dtype=torch.float32, device=x.device)
                    table[0][Lv] = t.data_ptr()
                    outs.append(t)
        _lib.check(L.dta_net_forward(C.byref(desc), nets, _lib.ptr(alpha), _lib.ptr(x), _lib.ptr(ws), C.byref(table),
                                     _lib.ptr(joint), _lib.current_stream_ptr()), "dta_net_forward")
        ctx.owner, ctx.desc, ctx.nets, ctx.heads_mask = owner, desc, nets, heads_mask
        if ctx.anchor:
            ctx.save_for_backward(ws)
        else:
            ctx.save_for_backward(ws, *params)
        if joint is not None:
            return joint
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        L = _lib.lib()
        owner, desc, nets = ctx.owner, ctx.desc, ctx.nets      # (the tables of the forward: same tensors, same addresses)
        ws, *params = ctx.saved_tensors
        plist = _cached_plist(owner)
        if ctx.anchor:
            params = plist
        nret = 1 if ctx.anchor else len(params)
        subnets = owner._subnets()
        hang = owner._net_code == _lib.NET_HANG2020
        pos = 1 if hang else 0
        alpha = params[0] if hang else None
        table = _lib.ScoreTable()
        djoint = None
        used_heads = 0
        keep = []
        if owner._net_code in (_lib.NET_HANG2020, _lib.NET_VANILLA):
            if gouts[0] is None:
                return (None, None, None) + (None,) * nret
            djoint = gouts[0].contiguous().float()
            used_heads = 4
        else:
            k = 0
            for Lv in range(3):
                if ctx.heads_mask & (1 << Lv):
                    g = gouts[k]
                    k += 1
                    if g is not None:
                        g = g.contiguous().float()
                        keep.append(g)
                        table[0][Lv] = g.data_ptr()
                        used_heads |= 1 << Lv
        wanted = _wanted(owner, used_heads)
        # gradient destinations: the owning optimizer's flat buffer in place (optim.DtaAdam: the parameters' .grad are
        # views of it and arrive cleared), else one zero-filled flat buffer returned to autograd
        sink = _cached_sink(owner, plist)
        cache = _table_cache(owner)
        if sink is not None and sink.take_inplace((id(owner), used_heads),
                                                  lambda: [plist[i] for i in wanted] + ([plist[0]] if hang else [])):
            gkey = ("grads", used_heads, id(sink), sink.layout_epoch)
            gstructs = cache.get(gkey)
            if gstructs is None:
                gstructs = (_lib.SubnetGrads * len(subnets))()
                wset = set(wanted)
                for i, (kind, mod, names) in enumerate(subnets):
                    gt = {n: plist[pos + j].grad for j, n in enumerate(names) if (pos + j) in wset}
                    pos += len(names)
                    _fill_struct(gstructs[i], kind, gt, True)
                for k in [k for k in cache if isinstance(k, tuple) and k and k[0] == "grads"]:
                    cache.pop(k)
                cache[gkey] = gstructs
            dalpha = plist[0].grad if hang else None
            _lib.check(L.dta_net_backward(C.byref(desc), nets, _lib.ptr(alpha), _lib.ptr(ws), C.byref(table),
                                          _lib.ptr(djoint), gstructs, _lib.ptr(dalpha), 3, _lib.current_stream_ptr()),
                       "dta_net_backward")
            return (None, None, None) + (None,) * nret
        grads = [None] * len(params)
        dalpha = torch.zeros((), dtype=torch.float64, device=ws.device) if hang else None
        if hang:
            grads[0] = dalpha
        flat = torch.zeros(sum(params[i].numel() for i in wanted), dtype=torch.float32, device=ws.device)
        off = 0
        for i in wanted:
            k = params[i].numel()
            grads[i] = flat[off:off + k].view(params[i].shape)
            off += k
        gstructs = (_lib.SubnetGrads * len(subnets))()
        for i, (kind, mod, names) in enumerate(subnets):
            gt = {n: grads[pos + j] for j, n in enumerate(names) if grads[pos + j] is not None}
            pos += len(names)
            _fill_struct(gstructs[i], kind, gt, True)
        _lib.check(L.dta_net_backward(C.byref(desc), nets, _lib.ptr(alpha), _lib.ptr(ws), C.byref(table),
                                      _lib.ptr(djoint), gstructs, _lib.ptr(dalpha), 3, _lib.current_stream_ptr()),
                   "dta_net_backward")
        if ctx.anchor:
            # the optimizer could not take this backward in place (a second backward before step(): gradient accumulation):
            # add into the parameters' gradients here, as autograd's accumulation would
            with torch.no_grad():
                for i, g in enumerate(grads):
                    if g is None:
                        continue
                    p = plist[i]
                    if p.grad is None:
                        p.grad = g.clone()
                    else:
                        p.grad.add_(g)
            return (None, None, None, None)
        return (None, None, None, *grads)


def _bump_epoch(module, incompatible_keys):
    module.__dict__["_dta_epoch"] = module.__dict__.get("_dta_epoch", 0) + 1


class _Net(nn.Module):
    """Shared plumbing of the four network classes."""
    _net_code = None
    _kind = None

    def _init_net(self, classes, precision):
        self._classes = int(classes)
        self.precision = precision or _DEFAULT_PRECISION
        _lib.dtype_code(self.precision)
        # load_state_dict may replace Parameter objects (assign=True): cached raw-pointer tables (engine.Predictor) watch
        # this counter
        self.__dict__["_dta_epoch"] = 0
        self.register_load_state_dict_post_hook(_bump_epoch)

    def _subnets(self):
        return [(self._kind, self, _subnet_param_names(self._kind))]

    def _param_list(self):
        out = []
        for kind, mod, names in self._subnets():
            out += [_get(mod, n) for n in names]
        return out

    def _run(self, x, heads_mask):
        x = _check_input(x)
        plist = _cached_plist(self)
        if torch.is_grad_enabled() and _cached_sink(self, plist) is not None:
            anchor = next((p for p in plist if p.requires_grad), None)
            if anchor is not None:
                return _NetFn.apply(self, x, heads_mask, anchor)      # anchor mode: see _NetFn.forward
        return _NetFn.apply(self, x, heads_mask, *plist)


def _build_subnet(self, kind, bands, classes):
    att = spectral_attention if kind == "spectral" else spatial_attention
    feats = (32, 64, 128) if kind == "spectral" else (128, 256, 512)
    self.conv1 = conv_module(in_channels=bands, filters=32)
    self.attention_1 = att(filters=32)
    self.classifier1 = Classifier(classes=classes, in_features=feats[0])
    self.conv2 = conv_module(in_channels=32, filters=64, maxpool_kernel=(2, 2))
    self.attention_2 = att(filters=64)
    self.classifier2 = Classifier(classes=classes, in_features=feats[1])
    self.conv3 = conv_module(in_channels=64, filters=128, maxpool_kernel=(2, 2))
    self.attention_3 = att(filters=128)
    self.classifier3 = Classifier(classes=classes, in_features=feats[2])


class spatial_network(_Net):
    """conv/spatial-attention/classifier x3 -- reference Hang2020.py:170-204; returns [scores1, scores2, scores3].
    Like the reference it only accepts 11x11 patches (its head sizes are hard-coded, :91-99)."""
    _net_code, _kind = _lib.NET_SPATIAL, "spatial"

    def __init__(self, bands, classes, precision=None):
        super().__init__()
        _build_subnet(self, "spatial", bands, classes)
        self._init_net(classes, precision)

    def forward(self, x):
        if x.shape[-2:] != (11, 11):
            raise RuntimeError("spatial_network expects 11x11 patches (classifier sizes are fixed, as in the reference)")
        return list(self._run(x, 7))


class spectral_network(_Net):
    """conv/spectral-attention/classifier x3 -- reference Hang2020.py:206-240; size-agnostic in H, W."""
    _net_code, _kind = _lib.NET_SPECTRAL, "spectral"

    def __init__(self, bands, classes, precision=None):
        super().__init__()
        _build_subnet(self, "spectral", bands, classes)
        self._init_net(classes, precision)

    def forward(self, x):
        return list(self._run(x, 7))


class Hang2020(_Net):
    """Both branches on the same input, last heads blended by sigmoid(alpha) -- reference Hang2020.py:242-263."""
    _net_code = _lib.NET_HANG2020

    def __init__(self, bands, classes, precision=None):
        super().__init__()
        self.spectral_network = spectral_network(bands, classes, precision)
        self.spatial_network = spatial_network(bands, classes, precision)
        self.alpha = nn.Parameter(torch.tensor(0.5, dtype=float), requires_grad=True)
        self._init_net(classes, precision)

    def _subnets(self):
        return [("spectral", self.spectral_network, _subnet_param_names("spectral")),
                ("spatial", self.spatial_network, _subnet_param_names("spatial"))]

    def _param_list(self):
        return [self.alpha] + super()._param_list()

    def forward(self, x):
        if x.shape[-2:] != (11, 11):
            raise RuntimeError("Hang2020 expects 11x11 patches (the spatial branch's classifier sizes are fixed)")
        joint = self._run(x, 4)
        self.weighted_average = torch.sigmoid(self.alpha.detach())
        return joint


class vanilla_CNN(_Net):
    """Baseline without attention -- reference Hang2020.py:33-53."""
    _net_code, _kind = _lib.NET_VANILLA, "vanilla"

    def __init__(self, bands, classes, precision=None):
        super().__init__()
        self.conv1 = conv_module(in_channels=bands, filters=32)
        self.conv2 = conv_module(in_channels=32, filters=64, maxpool_kernel=(2, 2))
        self.conv3 = conv_module(in_channels=64, filters=128, maxpool_kernel=(2, 2))
        self.fc1 = nn.Linear(in_features=512, out_features=classes)
        self._init_net(classes, precision)

    def forward(self, x):
        # fc1 is Linear(512, classes) (reference Hang2020.py:43): only patches whose twice-pooled map flattens to 512
        # features fit, and the reference raises a shape error otherwise -- the C side sizes the head GEMM from the
        # patch, so a mismatch must never reach it
        feats = 128 * (x.shape[-2] // 4) * (x.shape[-1] // 4)
        if feats != self.fc1.in_features:
            raise RuntimeError("vanilla_CNN: a {}x{} patch flattens to {} features but fc1 expects {} (the reference "
                               "raises the same shape error)".format(x.shape[-2], x.shape[-1], feats, self.fc1.in_features))
        return self._run(x, 4)


def load_from_backbone(state_dict, classes, bands):
    """Reference Hang2020.py:266-278: copy every non-classifier tensor of a saved spectral_network into a fresh
    one (possibly with a different number of classes)."""
    train_state_dict = torch.load(state_dict, map_location="cpu")
    model = spectral_network(classes=classes, bands=bands)
    merged = model.state_dict()
    merged.update({k: v for k, v in train_state_dict.items() if "classifier" not in k})
    model.load_state_dict(merged)
    return model
