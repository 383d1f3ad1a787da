"""GPU parity tests of the kernel variants bench.py actually times (round-1 verdict, weak #1-#3): the bf16 first conv that
reads the caller's fp32 NCHW tensor itself is only taken from ~100 first-conv workgroups (B >= ~400) and the
two-patches-per-workgroup stage kernels from B >= 512, i.e. at batch sizes the small oracle cases never reach.

Here the HIP path at B in {421, 530, 1024} is compared DIRECTLY with the NumPy oracle (seconds at 48 bands): against
the oracle run with the same bf16 operand rounding (implementation exactness: logits/loss 1e-3, whole gradient vector
1e-2 as BASELINE.json's north_star states for bf16), against the exact fp64 oracle (the bf16 budget itself), BatchNorm
buffers, and two FusedTrainer steps against the oracle's Adam loop.  A second group checks the full bench shape
(369 bands, 200 classes, B=1024) bf16 gradients against the fp32 HIP path (itself oracle-checked) and the train-mode
BatchNorm statistics against a split of the batch."""
import copy

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import hang2020_np as O
from oracle import prng

pytestmark = pytest.mark.gpu

BANDS, CLASSES = 48, 11


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _model(precision, seed=3, bands=BANDS, classes=CLASSES):
    from deeptreeattention_amd import Hang2020 as H
    p = O.init_params(O.hang2020_spec(bands, classes), seed=seed)
    m = H.Hang2020(bands, classes, precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
    return m.to(dev()).train(), p


def _whole(got, want, skip=("conv_layer.bias",)):
    """rel-L2 of the concatenated gradient vector (conv biases under batch-stat BN are analytically zero: skipped)."""
    num = den = 0.0
    worst = (0.0, None)
    for k, v in want.items():
        if any(k.endswith(s) for s in skip) or not np.any(v):
            continue
        g = np.asarray(got[k], np.float64)
        v = np.asarray(v, np.float64)
        num += float(((g - v) ** 2).sum())
        den += float((v ** 2).sum())
        worst = max(worst, (rel_l2(g, v), k))
    return float(np.sqrt(num / den)), worst


def _oracle(p, x, y, w, quantized):
    if quantized:
        O.bf16_mode(True)
    try:
        logits, cache, upd = O.hang2020_fwd(p, x, True, np.float64)
        loss, dl = O.weighted_cross_entropy(logits, y, w)
        g = O.hang2020_bwd(p, cache, dl, np.float64)
    finally:
        O.bf16_mode(False)
    return logits, loss, g, upd


@pytest.mark.parametrize("bands,classes,B", [(48, 11, 421), (48, 11, 530), (48, 11, 1024), (369, 200, 1024)])
def test_fused_input_bf16_step_vs_oracle(bands, classes, B, bf16_yardstick):
    """421: fused fp32-input conv1 + one-patch stage kernels (ragged: 421 = 105 x 4 + 1); 530: fused conv1 + two-patch
    stage kernels with a ragged tail; 1024: the bench batch; (369, 200, 1024): exactly the shape bench.py times (the
    NumPy oracle needs well under a minute for it)."""
    m, p = _model("bf16", bands=bands, classes=classes)
    x = prng.uniform01(40 + B, 1, (B, bands, 11, 11))
    y = prng.randint(40 + B, 2, (B,), classes)
    w = (0.1 + (np.arange(classes) % 7)).astype(np.float32)
    xt, yt, wt = torch.from_numpy(x).to(dev()), torch.from_numpy(y).to(dev()), torch.from_numpy(w).to(dev())
    logits = m(xt)
    loss = torch.nn.functional.cross_entropy(logits, yt, weight=wt)
    loss.backward()
    got = {k: (None if q.grad is None else q.grad.detach().cpu().numpy()) for k, q in m.named_parameters()}
    lg = logits.detach().cpu().numpy()

    # (1) same operand rounding in the oracle: implementation exactness of the benched kernels
    q_logits, q_loss, q_g, q_upd = _oracle(p, x, y, w, True)
    assert rel_l2(lg, q_logits) < 1e-3
    assert abs(loss.item() - q_loss) / q_loss < 1e-3
    whole, worst = _whole(got, q_g)
    print(f"B={B}: bf16 HIP vs bf16-operand oracle: whole-gradient rel-L2 {whole:.2e}, worst tensor {worst}")
    # Fixed tolerances.  north_star's bf16 budget names logits, loss and gradient NORMS: every tensor of >= 1000 elements
    # must keep its norm within 1e-2 (observed <= 3.1e-3 at 369 bands), and so must the total.  Element-wise, the whole
    # gradient vector is held to 1.5e-2: two float accumulations (float32 vs float64) of the SAME rounded step -- the
    # oracle against itself -- are 8.0e-3 apart at 369 bands (1.3e-3 at 48), because ~2 % of the half-precision conv
    # outputs round the other way and BatchNorm's backward cancels all but ~1 % of the weight-gradient sums
    # (tools/bf16diag2.py prints the three pairwise distances); 48 bands keep the 1e-2 of the budget.
    tag = "hang1024/" if bands == 369 else f"hang48_{B}/"      # the reference's own bf16-autocast run of this very step
    ref_whole = bf16_yardstick.ref(tag + "whole_elem_dev")
    print(f"B={B}: that is {whole / ref_whole:.2f} x the reference's own bf16 deviation from fp32 ({ref_whole:.2e})")
    assert whole < (min(1.5e-2, 0.25 * ref_whole) if bands > 100 else 1e-2)
    tot_q = tot_g = 0.0
    for k, v in q_g.items():        # per-tensor gradient norms (tensors that are more than a handful of scalars)
        if k.endswith("conv_layer.bias") or not np.any(v):
            continue
        n_q, n_g = np.linalg.norm(np.asarray(v, np.float64)), np.linalg.norm(np.asarray(got[k], np.float64))
        tot_q += n_q ** 2
        tot_g += n_g ** 2
        if np.asarray(v).size >= 1000:
            assert abs(n_g - n_q) <= 1e-2 * n_q, (k, n_g, n_q)
    assert abs(np.sqrt(tot_g) - np.sqrt(tot_q)) <= 1e-2 * np.sqrt(tot_q)
    # same-precision comparison: the oracle accumulating in float32 like the kernels (bf16-mode roundings in both)
    O.bf16_mode(True)
    try:
        l32, c32, _ = O.hang2020_fwd(p, x, True, np.float32)
        _, dl32 = O.weighted_cross_entropy(l32, y, w)
        g32 = O.hang2020_bwd(p, c32, dl32, np.float32)
    finally:
        O.bf16_mode(False)
    whole32, _ = _whole(got, g32)
    self32, _ = _whole(g32, q_g)
    print(f"B={B}: bf16 HIP vs float32-accumulating oracle {whole32:.2e}; that oracle vs its float64 run {self32:.2e}")
    assert whole32 < (min(1.5e-2, 0.25 * ref_whole) if bands > 100 else 1e-2)
    for k, v in g32.items():
        if np.asarray(v).size >= 1000 and not k.endswith("conv_layer.bias") and np.any(v):
            n_q, n_g = np.linalg.norm(np.asarray(v, np.float64)), np.linalg.norm(np.asarray(got[k], np.float64))
            assert abs(n_g - n_q) <= 1e-2 * n_q, (k, n_g, n_q)
    sd = m.state_dict()
    for k, v in q_upd.items():      # BatchNorm running statistics / counters after one train-mode forward
        assert rel_l2(sd[k].cpu().numpy(), v) < 1e-3, k

    # (2) the exact (unrounded, fp64) oracle: north_star's bf16 budget (1e-2) on logits, loss and gradient NORMS.
    # Element-wise the gradient of a bf16-operand step is 3-5 % away from the exact one in ANY implementation (the
    # oracle with rounded operands is exactly as far: measured 4.74e-2 vs 4.74e-2 at B=421), so the vector deviation
    # is checked against the deviation the operand rounding itself explains
    e_logits, e_loss, e_g, _ = _oracle(p, x, y, w, False)
    assert rel_l2(lg, e_logits) < 1e-2
    assert abs(loss.item() - e_loss) / e_loss < 1e-2
    whole_e, worst_e = _whole(got, e_g)
    whole_q, _ = _whole(q_g, e_g)
    tot = np.sqrt(sum(float((np.asarray(v, np.float64) ** 2).sum()) for k, v in got.items() if v is not None))
    tot_e = np.sqrt(sum(float((np.asarray(v, np.float64) ** 2).sum()) for v in e_g.values()))
    print(f"B={B}: bf16 HIP vs exact oracle: whole-gradient rel-L2 {whole_e:.2e} (rounded-operand oracle vs exact: "
          f"{whole_q:.2e}; worst tensor {worst_e}), total norm rel err {abs(tot - tot_e) / tot_e:.2e}")
    assert abs(tot - tot_e) / tot_e < 1e-2
    # the yardstick is the reference itself: its modules under torch.autocast("cpu", torch.bfloat16) on these very inputs
    # (tests/golden/bf16_autocast.npz): scores, loss, every >= 1000-element tensor (norm and element-wise), the whole
    # vector and the total norm within max(1e-2, 1.5 x the reference's own bf16 deviation)
    assert rel_l2(lg, e_logits) <= bf16_yardstick.bound(tag + "scores_dev")
    assert abs(loss.item() - e_loss) / e_loss <= bf16_yardstick.bound(tag + "loss_dev")
    bf16_yardstick.check_gradients(tag, {k: v for k, v in got.items() if v is not None}, {k: v for k, v in e_g.items() if got.get(k) is not None})


@pytest.mark.parametrize("B", [530, 1024])
def test_fused_trainer_bf16_steps_vs_oracle_adam(B):
    """Two FusedTrainer steps (forward + weighted CE + backward + Adam, the sequence bench.py times) against the
    oracle's Adam loop with the same operand rounding."""
    from deeptreeattention_amd.engine import FusedTrainer
    m, p = _model("bf16", seed=5)
    x = prng.uniform01(60 + B, 1, (B, BANDS, 11, 11))
    y = prng.randint(60 + B, 2, (B,), CLASSES)
    w = (0.1 + (np.arange(CLASSES) % 5)).astype(np.float32)
    tr = FusedTrainer(m, lr=1e-3, loss_weight=torch.from_numpy(w))
    xt, yt = torch.from_numpy(x).to(dev()), torch.from_numpy(y).to(dev())
    state = {}
    for step in range(2):
        loss = tr.train_step(xt, yt).item()
        q_logits, q_loss, q_g, q_upd = _oracle(p, x, y, w, True)
        assert abs(loss - q_loss) / q_loss < 2e-3, (step, loss, q_loss)
        # step 1 runs on parameters Adam has moved by +-lr per element (sign-like updates: an element whose tiny gradient
        # differs in the last bits moves the other way), so the second comparison is looser than the first
        assert rel_l2(tr.logits.cpu().numpy(), q_logits) < (1e-3 if step == 0 else 6e-3), step
        p = O.adam_step(p, q_g, state, lr=1e-3)
        p.update(q_upd)
    sd = m.state_dict()
    num = den = 0.0
    for k, v in p.items():
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == 2
            continue
        if k.endswith("conv_layer.bias"):
            continue   # zero gradient analytically: Adam's sign(noise) updates are not comparable
        a, b = sd[k].double().cpu().numpy(), np.asarray(v, np.float64)
        num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum())
    print(f"B={B}: parameters after 2 bf16 steps vs oracle Adam: rel-L2 {np.sqrt(num / den):.2e}")
    assert np.sqrt(num / den) < 3e-3
    assert tr.step_count == 2


# ---------------------------------------------------------------------------------------------------------------
# full bench shape
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full():
    from deeptreeattention_amd import Hang2020 as H
    torch.manual_seed(7)
    m = H.Hang2020(369, 200, precision="fp32")
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data.uniform_(0.5, 1.5)
            mod.bias.data.uniform_(-0.3, 0.3)
    m = m.to(dev()).train()
    g = torch.Generator(device=dev())
    g.manual_seed(11)
    x = torch.rand(1024, 369, 11, 11, device=dev(), generator=g)
    y = torch.randint(0, 200, (1024,), device=dev(), generator=g)
    return m, x, y


def _train_grads(m, x, y):
    m.zero_grad(set_to_none=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.reset_running_stats()
    out = m(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    return (out.detach().double().cpu().numpy(), loss.item(),
            {k: q.grad.detach().double().cpu().numpy() for k, q in m.named_parameters() if q.grad is not None},
            {k: v.detach().double().cpu().numpy() for k, v in m.state_dict().items() if "running_" in k})


def test_full_size_bf16_gradients_vs_fp32(full, bf16_yardstick):
    """B=1024, 369 bands, 200 classes, TRAIN mode (batch-statistics BatchNorm backward over 256 conv partials and
    1024 per-patch partials): bf16 path against the fp32 path of the same library (which the small cases pin to the
    oracle at 1e-5): logits, loss, every BatchNorm buffer and the whole gradient vector within the bf16 budget."""
    m, x, y = full
    mb = copy.deepcopy(m)
    mb.precision = mb.spectral_network.precision = mb.spatial_network.precision = "bf16"
    o32, l32, g32, b32 = _train_grads(m, x, y)
    o16, l16, g16, b16 = _train_grads(mb, x, y)
    assert rel_l2(o16, o32) < 1e-2
    assert abs(l16 - l32) / l32 < 1e-2
    whole, worst = _whole(g16, g32)
    tot16 = np.sqrt(sum(float((v ** 2).sum()) for v in g16.values()))
    tot32 = np.sqrt(sum(float((v ** 2).sum()) for v in g32.values()))
    print(f"full size: bf16 vs fp32 whole-gradient rel-L2 {whole:.2e} (worst {worst}); total norm rel "
          f"{abs(tot16 - tot32) / tot32:.2e}; logits {rel_l2(o16, o32):.2e}")
    assert abs(tot16 - tot32) / tot32 < 1e-2
    # a consistency check of the two precision modes of the SAME kernels at the bench shape (random torch-default weights,
    # not a parity case: bf16 parity at this shape is test_fused_input_bf16_step_vs_oracle).  The allowances are the
    # reference's own: its bf16-autocast run at this shape and batch (bf16_autocast.npz, hang1024/) moves each tensor's
    # norm / the whole vector by the figures below; max(1e-2, 1.5 x that) is what the bf16 mode may differ from the fp32 mode
    for k, v in g32.items():
        if v.size < 1000 or k.endswith("conv_layer.bias"):
            continue
        tol = bf16_yardstick.bound("hang1024/gnorm_dev/" + k)
        assert abs(np.linalg.norm(g16[k]) - np.linalg.norm(g32[k])) <= tol * np.linalg.norm(g32[k]), (k, tol)
    assert whole < bf16_yardstick.bound("hang1024/whole_elem_dev")
    for k in b32:
        assert rel_l2(b16[k], b32[k]) < 1e-2, k


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_full_size_train_mode_bn_statistics_split_consistency(full, precision):
    """Train-mode BatchNorm at B=1024: the batch mean / unbiased variance that end up in the running buffers (Chan
    combination of 256 per-workgroup partials) must equal the combination of the two half batches' statistics for the
    FIRST layer (whose input does not depend on other BatchNorm layers): mean = (m1 + m2)/2 and
    var_pop = (v1 + v2)/2 + ((m1 - m2)/2)^2."""
    m, x, y = full
    mm = copy.deepcopy(m)
    mm.precision = mm.spectral_network.precision = mm.spatial_network.precision = precision

    def stats(xs):
        for mod in mm.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.reset_running_stats()
                mod.momentum = 1.0          # running buffers = this batch's statistics
        with torch.no_grad():
            mm(xs)
        n = xs.shape[0] * 121
        out = {}
        for br in ("spectral_network", "spatial_network"):
            bn = getattr(mm, br).conv1.bn1
            out[br] = (bn.running_mean.double().cpu().numpy(), bn.running_var.double().cpu().numpy() * (n - 1) / n)
        return out

    from deeptreeattention_amd import Hang2020 as H
    momentum = H.BN_MOMENTUM
    try:
        H.BN_MOMENTUM = 1.0
        sf, s1, s2 = stats(x), stats(x[:512].contiguous()), stats(x[512:].contiguous())
    finally:
        H.BN_MOMENTUM = momentum
    for br in sf:
        mean = 0.5 * (s1[br][0] + s2[br][0])
        var = 0.5 * (s1[br][1] + s2[br][1]) + (0.5 * (s1[br][0] - s2[br][0])) ** 2
        assert rel_l2(sf[br][0], mean) < 1e-5, br
        assert rel_l2(sf[br][1], var) < 1e-4, br
