"""ORACLE (test infrastructure, never shipped, never the thing measured).

NumPy restatement, in this repo's own words, of the arithmetic of the reference's
Hang2020 hot path: forward, weighted cross-entropy, analytic backward and Adam.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
the product (deeptreeattention_amd) must never route through this file.

Parity status: PINNED.  tests/golden/*.npz hold outputs of the reference itself
(/root/reference/src/models/Hang2020.py imported in the build container by
tests/golden/make_golden.py) on inputs/weights from oracle/prng.py, and
tests/test_oracle_golden.py checks every function below against them.

Every function cites the reference lines it restates (paths relative to /root/reference).
Tensors are NCHW numpy arrays; parameters are a flat dict keyed by the reference's
state_dict names.
"""
import numpy as np

from . import prng

CH = (32, 64, 128)                       # src/models/Hang2020.py:174-188, 210-224
SPECTRAL_K = {32: 3, 64: 5, 128: 7}      # src/models/Hang2020.py:136-141
SPATIAL_K = {32: 7, 64: 5, 128: 3}       # src/models/Hang2020.py:77-85
SPATIAL_POOL = {32: 4, 64: 2, 128: 1}    # src/models/Hang2020.py:91-99
SPATIAL_FEAT = {32: 128, 64: 256, 128: 512}
BN_EPS = 1e-5                            # torch.nn.BatchNorm2d default (Hang2020.py:19)
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------------------
# parameter inventory (names/shapes = the drop-in state_dict contract)
# ----------------------------------------------------------------------------------------
def conv_module_spec(prefix, cin, cout):
    """src/models/Hang2020.py:14-22 (Conv2d 3x3 same + BatchNorm2d)."""
    return [
        (prefix + "conv_layer.weight", (cout, cin, 3, 3), "w", cin * 9),
        (prefix + "conv_layer.bias", (cout,), "w", cin * 9),
        (prefix + "bn1.weight", (cout,), "gamma", 0),
        (prefix + "bn1.bias", (cout,), "beta", 0),
        (prefix + "bn1.running_mean", (cout,), "rmean", 0),
        (prefix + "bn1.running_var", (cout,), "rvar", 0),
        (prefix + "bn1.num_batches_tracked", (), "nbt", 0),
    ]


def spectral_attention_spec(prefix, c):
    """src/models/Hang2020.py:133-147 (two Conv1d(C,C,K,same))."""
    k = SPECTRAL_K[c]
    return [
        (prefix + "attention_conv1.weight", (c, c, k), "w", c * k),
        (prefix + "attention_conv1.bias", (c,), "w", c * k),
        (prefix + "attention_conv2.weight", (c, c, k), "w", c * k),
        (prefix + "attention_conv2.bias", (c,), "w", c * k),
    ]


def spatial_attention_spec(prefix, c):
    """src/models/Hang2020.py:72-88 (1x1 channel pool + two kxk single-channel convs)."""
    k = SPATIAL_K[c]
    return [
        (prefix + "channel_pool.weight", (1, c, 1, 1), "w", c),
        (prefix + "channel_pool.bias", (1,), "w", c),
        (prefix + "attention_conv1.weight", (1, 1, k, k), "w", k * k),
        (prefix + "attention_conv1.bias", (1,), "w", k * k),
        (prefix + "attention_conv2.weight", (1, 1, k, k), "w", k * k),
        (prefix + "attention_conv2.bias", (1,), "w", k * k),
    ]


def classifier_spec(prefix, fin, classes):
    """src/models/Hang2020.py:59-61."""
    return [
        (prefix + "fc1.weight", (classes, fin), "w", fin),
        (prefix + "fc1.bias", (classes,), "w", fin),
    ]


def subnet_spec(kind, bands, classes, prefix=""):
    """src/models/Hang2020.py:170-188 (spatial) / 206-224 (spectral); registration order."""
    spec = []
    cin = bands
    for i, c in enumerate(CH):
        L = i + 1
        spec += conv_module_spec(f"{prefix}conv{L}.", cin, c)
        if kind == "spectral":
            spec += spectral_attention_spec(f"{prefix}attention_{L}.", c)
            fin = c
        else:
            spec += spatial_attention_spec(f"{prefix}attention_{L}.", c)
            fin = SPATIAL_FEAT[c]
        spec += classifier_spec(f"{prefix}classifier{L}.", fin, classes)
        cin = c
    return spec


def hang2020_spec(bands, classes, prefix=""):
    """src/models/Hang2020.py:243-249."""
    return ([(prefix + "alpha", (), "alpha", 0)]
            + subnet_spec("spectral", bands, classes, prefix + "spectral_network.")
            + subnet_spec("spatial", bands, classes, prefix + "spatial_network."))


def vanilla_spec(bands, classes, prefix=""):
    """src/models/Hang2020.py:37-43."""
    return (conv_module_spec(prefix + "conv1.", bands, 32)
            + conv_module_spec(prefix + "conv2.", 32, 64)
            + conv_module_spec(prefix + "conv3.", 64, 128)
            + classifier_spec(prefix, 512, classes))


def learned_ensemble_spec(years, bands, classes, prefix=""):
    """src/models/year.py:13-22."""
    spec = []
    for y in range(years):
        spec += subnet_spec("spectral", bands, classes, f"{prefix}year_models.{y}.")
    return spec


def init_params(spec, seed, randomize_bn=True):
    """Portable weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's default Conv/Linear
    reset_parameters; BN affine/running stats randomised (so that a wrong BN shows up) unless
    randomize_bn=False, which gives torch's defaults (gamma 1, beta 0, mean 0, var 1)."""
    p = {}
    for name, shape, kind, fan_in in spec:
        sid = prng.stream_id(name)
        if kind == "w":
            b = 1.0 / np.sqrt(fan_in)
            p[name] = prng.uniform(seed, sid, shape, -b, b)
        elif kind == "gamma":
            p[name] = prng.uniform(seed, sid, shape, 0.5, 1.5) if randomize_bn else np.ones(shape, np.float32)
        elif kind == "beta":
            p[name] = prng.uniform(seed, sid, shape, -0.3, 0.3) if randomize_bn else np.zeros(shape, np.float32)
        elif kind == "rmean":
            p[name] = prng.uniform(seed, sid, shape, -0.2, 0.2) if randomize_bn else np.zeros(shape, np.float32)
        elif kind == "rvar":
            p[name] = prng.uniform(seed, sid, shape, 0.5, 1.5) if randomize_bn else np.ones(shape, np.float32)
        elif kind == "nbt":
            p[name] = np.array(0, dtype=np.int64)
        elif kind == "alpha":
            p[name] = np.array(0.5 if not randomize_bn else 0.3, dtype=np.float64)  # Hang2020.py:249 (float64)
        else:
            raise ValueError(kind)
    return p


def is_buffer(name):
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


# ----------------------------------------------------------------------------------------
# primitive ops
# ----------------------------------------------------------------------------------------
def _windows(x, k):
    p = k // 2
    xp = np.pad(x, ((0, 0), (0, 0), (p, p), (p, p)))
    return np.lib.stride_tricks.sliding_window_view(xp, (k, k), axis=(2, 3))  # B,C,H,W,k,k


def conv2d_same(x, w, b=None):
    """Cross-correlation with zero 'same' padding (nn.Conv2d(padding="same"), Hang2020.py:18)."""
    y = np.einsum("bchwij,ncij->bnhw", _windows(x, w.shape[2]), w, optimize=True)
    if b is not None:
        y = y + b[None, :, None, None]
    return y


def conv2d_same_bwd(x, w, dy, need_dx=True):
    k = w.shape[2]
    dw = np.einsum("bchwij,bnhw->ncij", _windows(x, k), dy, optimize=True)
    db = dy.sum(axis=(0, 2, 3))
    dx = None
    if need_dx:
        wt = np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))
        dx = conv2d_same(dy, wt)
    return dx, dw, db


def maxpool_floor(x, k):
    """nn.MaxPool2d(k) (stride k, floor): Hang2020.py:22, :103.  Returns pooled + flat argmax
    (first maximum in row-major window order, as ATen's CPU/GPU kernels pick)."""
    B, C, H, W = x.shape
    ho, wo = H // k, W // k
    xc = x[:, :, :ho * k, :wo * k].reshape(B, C, ho, k, wo, k).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, ho, wo, k * k)
    arg = xc.argmax(axis=-1)
    return np.take_along_axis(xc, arg[..., None], -1)[..., 0], arg


def maxpool_floor_bwd(dy, arg, k, H, W):
    B, C, ho, wo = dy.shape
    d = np.zeros((B, C, ho, wo, k * k), dy.dtype)
    np.put_along_axis(d, arg[..., None], dy[..., None], -1)
    dx = np.zeros((B, C, H, W), dy.dtype)
    dx[:, :, :ho * k, :wo * k] = d.reshape(B, C, ho, wo, k, k).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, ho * k, wo * k)
    return dx


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def bf16_round(a):
    """Round to bfloat16 (nearest-even) and return in the input's dtype: emulates where the bf16 mode of the
    HIP path quantises conv operands (inputs, weights, output gradients); everything else stays wide."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(a.dtype if hasattr(a, "dtype") else np.float32)


_QUANT = None


def set_conv_operand_quantizer(fn):
    """None (default) = exact; bf16_round = emulate the bf16 compute mode's operand rounding."""
    global _QUANT
    _QUANT = fn


def _q(a):
    return a if _QUANT is None else _QUANT(a)


def f16_round(a):
    """Round to IEEE half (saturating at +-65504) and return in the input's dtype: how the bf16 mode of the HIP path
    stores the conv outputs between kernels."""
    return np.clip(a, -65504.0, 65504.0).astype(np.float16).astype(a.dtype)


_QUANT_Y = None
_QUANT_G = None


def set_storage_quantizers(conv_output=None, grad_map=None):
    """Emulate the 16-bit inter-kernel storage of the HIP path's bf16 mode: `conv_output` rounds the pre-BatchNorm conv
    outputs AFTER their batch statistics were taken from the exact values (f16_round), `grad_map` rounds the gradient
    maps handed from one backward kernel to the next (bf16_round).  None = keep wide (default)."""
    global _QUANT_Y, _QUANT_G
    _QUANT_Y, _QUANT_G = conv_output, grad_map


def bf16_mode(on=True):
    """All roundings of the HIP path's bf16 mode at once (operands bf16, conv outputs half, gradient maps bf16)."""
    set_conv_operand_quantizer(bf16_round if on else None)
    set_storage_quantizers(f16_round if on else None, bf16_round if on else None)


def _qy(a):
    return a if _QUANT_Y is None else _QUANT_Y(a)


def _qg(a):
    return a if _QUANT_G is None else _QUANT_G(a)


# ----------------------------------------------------------------------------------------
# conv_module  (src/models/Hang2020.py:14-31)
# ----------------------------------------------------------------------------------------
def conv_module_fwd(p, pre, x, pool, training, dt=np.float32):
    """conv3x3+bias (:25) -> BatchNorm2d (:26; batch stats when training, running stats otherwise)
    -> ReLU (:27) -> optional MaxPool2d(2) (:28-29).  Returns output, cache, and the running-stat
    updates torch would apply (momentum 0.1, unbiased variance)."""
    w = _q(p[pre + "conv_layer.weight"].astype(dt))
    x = _q(x.astype(dt))
    y = conv2d_same(x, w, p[pre + "conv_layer.bias"].astype(dt))
    g, be = p[pre + "bn1.weight"].astype(dt), p[pre + "bn1.bias"].astype(dt)
    upd = {}
    if training:
        n = y.shape[0] * y.shape[2] * y.shape[3]
        mu = y.mean(axis=(0, 2, 3))
        var = y.var(axis=(0, 2, 3))
        upd[pre + "bn1.running_mean"] = ((1 - BN_MOMENTUM) * p[pre + "bn1.running_mean"] + BN_MOMENTUM * mu).astype(np.float32)
        upd[pre + "bn1.running_var"] = ((1 - BN_MOMENTUM) * p[pre + "bn1.running_var"]
                                        + BN_MOMENTUM * var * n / max(n - 1, 1)).astype(np.float32)
        upd[pre + "bn1.num_batches_tracked"] = p[pre + "bn1.num_batches_tracked"] + 1
    else:
        mu = p[pre + "bn1.running_mean"].astype(dt)
        var = p[pre + "bn1.running_var"].astype(dt)
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    y = _qy(y)        # (statistics above come from the unrounded values, as in the conv kernel's epilogue)
    xhat = (y - mu[None, :, None, None]) * rstd[None, :, None, None]
    v = xhat * g[None, :, None, None] + be[None, :, None, None]
    r = np.maximum(v, 0)
    arg = None
    if pool:
        z, arg = maxpool_floor(r, 2)
    else:
        z = r
    cache = dict(x=x.astype(dt), w=w, xhat=xhat, rstd=rstd, g=g, v=v, arg=arg, pool=pool, hw=r.shape[2:], training=training)
    return z, cache, upd


def conv_module_bwd(cache, pre, dz, need_dx=True):
    c = cache
    if c["pool"]:
        dr = maxpool_floor_bwd(dz, c["arg"], 2, *c["hw"])
    else:
        dr = dz
    dv = _qg(dr * (c["v"] > 0))     # (bf16 mode of the HIP path: this map crosses HBM in bf16)
    n = dv.shape[0] * dv.shape[2] * dv.shape[3]
    dbeta = dv.sum(axis=(0, 2, 3))
    dgamma = (dv * c["xhat"]).sum(axis=(0, 2, 3))
    if c["training"]:
        dy = (c["g"] * c["rstd"])[None, :, None, None] * (
            dv - dbeta[None, :, None, None] / n - c["xhat"] * dgamma[None, :, None, None] / n)
    else:
        dy = (c["g"] * c["rstd"])[None, :, None, None] * dv
    dx, dw, db = conv2d_same_bwd(c["x"], c["w"], _q(dy), need_dx)
    if dx is not None:
        dx = _qg(dx)                # (likewise the input-gradient map handed to the previous stage)
    grads = {pre + "conv_layer.weight": dw, pre + "conv_layer.bias": db,
             pre + "bn1.weight": dgamma, pre + "bn1.bias": dbeta}
    return dx, grads


# ----------------------------------------------------------------------------------------
# spectral_attention  (src/models/Hang2020.py:126-168)
# ----------------------------------------------------------------------------------------
def spectral_attention_fwd(p, pre, z, dt=np.float32):
    """global mean over H,W (:7-12,:152) -> Conv1d (:155) -> ReLU -> Conv1d (:157) -> sigmoid (:158)
    -> channel gate (:161-162) -> pooled mean of the gated map (:165-166).  The pooled vector is a
    length-1 sequence, so with 'same' padding only tap K//2 of each Conv1d ever meets data."""
    z = z.astype(dt)
    k = p[pre + "attention_conv1.weight"].shape[2]
    a1 = p[pre + "attention_conv1.weight"][:, :, k // 2].astype(dt)
    a2 = p[pre + "attention_conv2.weight"][:, :, k // 2].astype(dt)
    c1 = p[pre + "attention_conv1.bias"].astype(dt)
    c2 = p[pre + "attention_conv2.bias"].astype(dt)
    pooled = z.mean(axis=(2, 3))                       # B,C
    h0 = pooled @ a1.T + c1
    h = np.maximum(h0, 0)
    g = sigmoid(h @ a2.T + c2)
    a = z * g[:, :, None, None]
    f = a.mean(axis=(2, 3))
    cache = dict(z=z, a1=a1, a2=a2, pooled=pooled, h0=h0, h=h, g=g, k=k)
    return a, f, cache


def spectral_attention_bwd(cache, pre, da, df):
    c = cache
    z = c["z"]
    hw = z.shape[2] * z.shape[3]
    dat = (np.zeros_like(z) if da is None else da) + df[:, :, None, None] / hw
    dg = (dat * z).sum(axis=(2, 3))
    d2 = dg * c["g"] * (1 - c["g"])
    da2 = d2.T @ c["h"]
    dh = d2 @ c["a2"]
    d1 = dh * (c["h0"] > 0)
    da1 = d1.T @ c["pooled"]
    dp = d1 @ c["a1"]
    dz = dat * c["g"][:, :, None, None] + dp[:, :, None, None] / hw
    C, k = z.shape[1], c["k"]
    w1 = np.zeros((C, C, k), z.dtype)
    w2 = np.zeros((C, C, k), z.dtype)
    w1[:, :, k // 2] = da1
    w2[:, :, k // 2] = da2
    grads = {pre + "attention_conv1.weight": w1, pre + "attention_conv1.bias": d1.sum(0),
             pre + "attention_conv2.weight": w2, pre + "attention_conv2.bias": d2.sum(0)}
    return dz, grads


# ----------------------------------------------------------------------------------------
# spatial_attention  (src/models/Hang2020.py:68-124)
# ----------------------------------------------------------------------------------------
def spatial_attention_fwd(p, pre, z, dt=np.float32):
    """1x1 conv C->1 (:108) -> ReLU -> kxk conv (:112) -> ReLU -> kxk conv (:114) -> sigmoid (:115)
    -> per-pixel gate (:118) -> MaxPool2d(pool) (:121) -> NCHW flatten (:122)."""
    z = z.astype(dt)
    C = z.shape[1]
    wc = p[pre + "channel_pool.weight"].astype(dt)
    m0 = conv2d_same(z, wc, p[pre + "channel_pool.bias"].astype(dt))
    m = np.maximum(m0, 0)
    k1 = p[pre + "attention_conv1.weight"].astype(dt)
    k2 = p[pre + "attention_conv2.weight"].astype(dt)
    t1p = conv2d_same(m, k1, p[pre + "attention_conv1.bias"].astype(dt))
    t1 = np.maximum(t1p, 0)
    s = sigmoid(conv2d_same(t1, k2, p[pre + "attention_conv2.bias"].astype(dt)))
    a = z * s
    ps = SPATIAL_POOL[C]
    pooled, arg = maxpool_floor(a, ps)
    f = pooled.reshape(z.shape[0], -1)
    cache = dict(z=z, wc=wc, m0=m0, m=m, k1=k1, k2=k2, t1p=t1p, t1=t1, s=s, arg=arg, ps=ps, pshape=pooled.shape)
    return a, f, cache


def spatial_attention_bwd(cache, pre, da, df):
    c = cache
    z = c["z"]
    dat = maxpool_floor_bwd(df.reshape(c["pshape"]), c["arg"], c["ps"], z.shape[2], z.shape[3])
    if da is not None:
        dat = dat + da
    ds = (dat * z).sum(axis=1, keepdims=True)
    d2 = ds * c["s"] * (1 - c["s"])
    dt1, dk2, db2 = conv2d_same_bwd(c["t1"], c["k2"], d2)
    d1 = dt1 * (c["t1p"] > 0)
    dm, dk1, db1 = conv2d_same_bwd(c["m"], c["k1"], d1)
    dm0 = dm * (c["m0"] > 0)
    dzc, dwc, dbc = conv2d_same_bwd(z, c["wc"], dm0)
    dz = dat * c["s"] + dzc
    grads = {pre + "channel_pool.weight": dwc, pre + "channel_pool.bias": dbc,
             pre + "attention_conv1.weight": dk1, pre + "attention_conv1.bias": db1,
             pre + "attention_conv2.weight": dk2, pre + "attention_conv2.bias": db2}
    return dz, grads


# ----------------------------------------------------------------------------------------
# Classifier (src/models/Hang2020.py:55-66)
# ----------------------------------------------------------------------------------------
def classifier_fwd(p, pre, f, dt=np.float32):
    return f.astype(dt) @ p[pre + "fc1.weight"].astype(dt).T + p[pre + "fc1.bias"].astype(dt)


def classifier_bwd(p, pre, f, ds, dt=np.float32):
    return ds @ p[pre + "fc1.weight"].astype(dt), {pre + "fc1.weight": ds.T @ f, pre + "fc1.bias": ds.sum(0)}


# ----------------------------------------------------------------------------------------
# spectral_network / spatial_network  (src/models/Hang2020.py:190-204, 226-240)
# ----------------------------------------------------------------------------------------
def subnet_fwd(p, pre, kind, x, training, dt=np.float32):
    """conv1 -> att1 -> cls1 -> conv2(pool) -> att2 -> cls2 -> conv3(pool) -> att3 -> cls3; the gated
    map (not the features) feeds the next conv (:229,:232).  Returns [scores1, scores2, scores3]."""
    att_fwd = spectral_attention_fwd if kind == "spectral" else spatial_attention_fwd
    scores, caches, upd = [], [], {}
    u = x
    for i in range(3):
        L = i + 1
        z, cc, up = conv_module_fwd(p, f"{pre}conv{L}.", u, pool=(L > 1), training=training, dt=dt)
        upd.update(up)
        a, f, ac = att_fwd(p, f"{pre}attention_{L}.", z, dt)
        scores.append(classifier_fwd(p, f"{pre}classifier{L}.", f, dt))
        caches.append((cc, ac, f))
        u = a
    return scores, dict(kind=kind, layers=caches, att=[c[1] for c in caches]), upd


def subnet_bwd(p, pre, cache, dscores, dt=np.float32):
    """dscores: list of 3 (None = head unused by the loss, its parameters get no gradient)."""
    att_bwd = spectral_attention_bwd if cache["kind"] == "spectral" else spatial_attention_bwd
    grads = {}
    da = None
    for i in (2, 1, 0):
        L = i + 1
        cc, ac, f = cache["layers"][i]
        if dscores[i] is not None:
            df, g = classifier_bwd(p, f"{pre}classifier{L}.", f, dscores[i].astype(dt), dt)
            grads.update(g)
        else:
            df = np.zeros_like(f)
        if da is None and dscores[i] is None:
            continue
        dz, g = att_bwd(ac, f"{pre}attention_{L}.", da, df)
        grads.update(g)
        da, g = conv_module_bwd(cc, f"{pre}conv{L}.", dz, need_dx=(i > 0))
        grads.update(g)
    return grads


# ----------------------------------------------------------------------------------------
# Hang2020  (src/models/Hang2020.py:242-263)
# ----------------------------------------------------------------------------------------
def hang2020_fwd(p, x, training, dt=np.float32, pre=""):
    """Both branches on the same x (:252-253), last heads only (:256-257), sigmoid(alpha) blend
    (:260-261); alpha is float64 (:249), the blended scores stay float32."""
    s_spec, c_spec, u1 = subnet_fwd(p, pre + "spectral_network.", "spectral", x, training, dt)
    s_spat, c_spat, u2 = subnet_fwd(p, pre + "spatial_network.", "spatial", x, training, dt)
    w = 1.0 / (1.0 + np.exp(-np.float64(p[pre + "alpha"])))
    joint = (s_spec[2] * dt(w) + s_spat[2] * dt(1 - w)).astype(dt)
    u1.update(u2)
    return joint, dict(spec=c_spec, spat=c_spat, w=w, s_spec=s_spec, s_spat=s_spat), u1


def hang2020_bwd(p, cache, djoint, dt=np.float32, pre=""):
    w = cache["w"]
    grads = {pre + "alpha": np.float64(
        (djoint.astype(np.float64) * (cache["s_spec"][2].astype(np.float64) - cache["s_spat"][2])).sum() * w * (1 - w))}
    grads.update(subnet_bwd(p, pre + "spectral_network.", cache["spec"], [None, None, djoint * dt(w)], dt))
    grads.update(subnet_bwd(p, pre + "spatial_network.", cache["spat"], [None, None, djoint * dt(1 - w)], dt))
    return grads


# ----------------------------------------------------------------------------------------
# vanilla_CNN  (src/models/Hang2020.py:33-53)
# ----------------------------------------------------------------------------------------
def vanilla_fwd(p, x, training, dt=np.float32, pre=""):
    caches, upd = [], {}
    u = x
    for L in (1, 2, 3):
        u, cc, up = conv_module_fwd(p, f"{pre}conv{L}.", u, pool=(L > 1), training=training, dt=dt)
        caches.append(cc)
        upd.update(up)
    f = u.reshape(u.shape[0], -1)
    return classifier_fwd(p, pre, f, dt), dict(layers=caches, f=f, shape=u.shape), upd


def vanilla_bwd(p, cache, dscores, dt=np.float32, pre=""):
    df, grads = classifier_bwd(p, pre, cache["f"], dscores.astype(dt), dt)
    d = df.reshape(cache["shape"])
    for L in (3, 2, 1):
        d, g = conv_module_bwd(cache["layers"][L - 1], f"{pre}conv{L}.", d, need_dx=(L > 1))
        grads.update(g)
    return grads


# ----------------------------------------------------------------------------------------
# learned_ensemble  (src/models/year.py:24-33)
# ----------------------------------------------------------------------------------------
def learned_ensemble_fwd(p, images, training, dt=np.float32, pre=""):
    """One spectral_network per year; a year is skipped only when its whole batch tensor sums to
    zero (:27); mean of the kept years' last heads (:30,:33)."""
    kept, caches, upd = [], [], {}
    for y, x in enumerate(images):
        if x.sum() == 0:
            caches.append(None)
            continue
        s, c, u = subnet_fwd(p, f"{pre}year_models.{y}.", "spectral", x, training, dt)
        kept.append(s[2])
        caches.append(c)
        upd.update(u)
    return np.stack(kept, axis=1).mean(axis=1), dict(years=caches, n=len(kept)), upd


def learned_ensemble_bwd(p, cache, dscore, dt=np.float32, pre=""):
    grads = {}
    for y, c in enumerate(cache["years"]):
        if c is None:
            continue
        grads.update(subnet_bwd(p, f"{pre}year_models.{y}.", c, [None, None, dscore / cache["n"]], dt))
    return grads


# ----------------------------------------------------------------------------------------
# loss + optimizer  (src/main.py:78, :136; src/models/multi_stage.py:285)
# ----------------------------------------------------------------------------------------
def weighted_cross_entropy(logits, y, w):
    """F.cross_entropy(y_hat, y, weight=w): sum_i w[y_i] * -log_softmax(z_i)[y_i] / sum_i w[y_i]."""
    z = logits.astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    wi = w.astype(np.float64)[y]
    den = wi.sum()
    loss = -(wi * logp[np.arange(len(y)), y]).sum() / den
    d = np.exp(logp)
    d[np.arange(len(y)), y] -= 1.0
    d *= (wi / den)[:, None]
    return loss, d.astype(logits.dtype)


def adam_step(p, g, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (src/main.py:136): no weight decay, no amsgrad; parameters whose
    gradient is missing are skipped."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    out = dict(p)
    for k, gk in g.items():
        if gk is None:
            continue
        dt = np.float64 if p[k].dtype == np.float64 else np.float32
        gk = np.asarray(gk, dtype=dt)
        m = state.setdefault("m", {}).get(k, np.zeros_like(gk))
        v = state.setdefault("v", {}).get(k, np.zeros_like(gk))
        m = dt(beta1) * m + dt(1 - beta1) * gk
        v = dt(beta2) * v + dt(1 - beta2) * gk * gk
        state["m"][k], state["v"][k] = m, v
        bc1 = 1 - beta1 ** t
        bc2 = 1 - beta2 ** t
        denom = np.sqrt(v) / dt(np.sqrt(bc2)) + dt(eps)
        out[k] = (p[k] - dt(lr / bc1) * (m / denom)).astype(p[k].dtype)
    return out
