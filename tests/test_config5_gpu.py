"""BASELINE.json configs[4] at full size: the year ensemble of three 369-band spectral networks on 24x24 crops, bf16,
B = 64 (round-1 verdict item 7).  Anchors: (1) one year of the grouped launch against the NumPy oracle run with the
same bf16-mode roundings (the other two years zero-filled, i.e. skipped exactly as reference src/models/year.py:27
skips them), (2) the three-year ensemble against the mean of the three networks run one by one, (3) the documented size
limit of the per-patch stage kernels (the patch must fit LDS) raises instead of mis-computing."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import hang2020_np as O
from oracle import prng

pytestmark = pytest.mark.gpu
BANDS, CLASSES, HW, B, YEARS = 369, 200, 24, 64, 3


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ensemble():
    from deeptreeattention_amd.year import learned_ensemble
    p = O.init_params(O.learned_ensemble_spec(YEARS, BANDS, CLASSES), seed=17)
    m = learned_ensemble(YEARS, CLASSES, {"pretrain_state_dict": None, "bands": BANDS})
    for net in m.year_models:
        net.precision = "bf16"
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
    imgs = [prng.uniform01(18, yy, (B, BANDS, HW, HW)) for yy in range(YEARS)]
    y = prng.randint(18, 9, (B,), CLASSES)
    return m.to(dev()).train(), p, imgs, y


def test_one_year_of_the_grouped_launch_vs_oracle(ensemble, bf16_yardstick):
    m, p, imgs, y = ensemble
    w = np.ones(CLASSES, np.float32)
    xs = [torch.from_numpy(imgs[0]).to(dev())] + [torch.zeros(B, BANDS, HW, HW, device=dev()) for _ in range(YEARS - 1)]
    m.zero_grad(set_to_none=True)
    scores = m(xs)                                   # years 1, 2 all-zero: skipped (year.py:27), mean over one year
    loss = torch.nn.functional.cross_entropy(scores, torch.from_numpy(y).to(dev()))
    loss.backward()
    pre = "year_models.0."
    O.bf16_mode(True)
    try:
        heads, cache, upd = O.subnet_fwd(p, pre, "spectral", imgs[0], True, np.float64)
        rl, dl = O.weighted_cross_entropy(heads[2], y, w)
        g = O.subnet_bwd(p, pre, cache, [None, None, dl.astype(np.float64)], np.float64)
    finally:
        O.bf16_mode(False)
    assert rel_l2(scores.detach().cpu().numpy(), heads[2]) < 2e-3
    assert abs(loss.item() - rl) / rl < 2e-3
    num = den = 0.0
    worst = (0.0, None)
    for k, prm in m.named_parameters():
        if not k.startswith(pre):
            assert prm.grad is None, k                # skipped years: grad None, as in the reference
            continue
        if "classifier1" in k or "classifier2" in k:
            assert prm.grad is None, k
            continue
        if k.endswith("conv_layer.bias") or not np.any(g[k]):
            continue
        a, b = prm.grad.double().cpu().numpy(), np.asarray(g[k], np.float64)
        num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum())
        worst = max(worst, (rel_l2(a, b), k))
        if b.size >= 1000:      # north_star's bf16 budget (1e-2) is on the gradient NORMS
            assert abs(np.linalg.norm(a) - np.linalg.norm(b)) <= 1e-2 * np.linalg.norm(b), k
    whole = np.sqrt(num / den)
    print(f"config 5 (24x24, 369 bands, B={B}): year 0 whole-gradient rel-L2 vs bf16-mode oracle {whole:.2e}, worst {worst}")
    # element-wise against the oracle with the kernels' own roundings (implementation exactness; observed 1.46e-2: the
    # resolution of two float accumulations of the same rounded step, tests/test_hip_benched_path.py) -- 0.11 x what the
    # reference's own bf16-autocast run moves this vector (1.28e-1, bf16_autocast.npz spec24/)
    ref_whole = bf16_yardstick.ref("spec24/whole_elem_dev")
    print(f"  = {whole / ref_whole:.2f} x the reference's own bf16 deviation ({ref_whole:.2e})")
    assert whole < min(2e-2, 0.25 * ref_whole)
    # the parity criterion proper: against the EXACT oracle, within max(1e-2, 1.5 x the reference's own bf16 deviation)
    e_heads, e_cache, _ = O.subnet_fwd(p, pre, "spectral", imgs[0], True, np.float64)
    e_loss, e_dl = O.weighted_cross_entropy(e_heads[2], y, w)
    e_g = O.subnet_bwd(p, pre, e_cache, [None, None, e_dl.astype(np.float64)], np.float64)
    del e_cache
    assert rel_l2(scores.detach().cpu().numpy(), e_heads[2]) <= bf16_yardstick.bound("spec24/scores_dev")
    assert abs(loss.item() - e_loss) / e_loss <= bf16_yardstick.bound("spec24/loss_dev")
    got = {k[len(pre):]: q.grad.double().cpu().numpy() for k, q in m.named_parameters() if k.startswith(pre) and q.grad is not None}
    bf16_yardstick.check_gradients("spec24/", got, {k[len(pre):]: v for k, v in e_g.items() if k[len(pre):] in got})
    sd = m.state_dict()
    for k, v in upd.items():
        assert rel_l2(sd[k].cpu().numpy(), v) < 2e-3, k
    assert int(sd["year_models.1.conv1.bn1.num_batches_tracked"]) == 0     # a skipped year's BatchNorm is untouched


def test_three_year_ensemble_is_the_mean_of_its_networks(ensemble):
    m, p, imgs, y = ensemble
    xs = [torch.from_numpy(a).to(dev()) for a in imgs]
    m.eval()                                           # running statistics: calling order does not matter
    with torch.no_grad():
        s = m(xs)
        one_by_one = torch.stack([net(x)[-1] for net, x in zip(m.year_models, xs)]).mean(0)
    m.train()
    assert torch.isfinite(s).all()
    assert rel_l2(s.cpu().numpy(), one_by_one.cpu().numpy()) < 1e-5


def test_fused_ensemble_step_runs_at_full_size(ensemble):
    from deeptreeattention_amd.engine import EnsembleTrainer
    import copy
    m, p, imgs, y = ensemble
    tr = EnsembleTrainer(copy.deepcopy(m), lr=1e-3)
    xs = [torch.from_numpy(a).to(dev()) for a in imgs]
    yt = torch.from_numpy(y).to(dev())
    losses = [float(tr.train_step(xs, yt, present=[True, True, True])) for _ in range(4)]
    print("config 5 fused steps:", [round(v, 4) for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]


def test_patch_size_limit_of_the_stage_kernels_is_an_error():
    """The per-patch stage kernels keep one patch in LDS (160 KiB per workgroup): 24x24 fits, 32x32 does not -- a clear
    error, never a silent wrong answer (known limits)."""
    from deeptreeattention_amd import Hang2020 as H
    m = H.spectral_network(16, 5, precision="bf16").to(dev()).train()
    ok = m(torch.rand(2, 16, 24, 24, device=dev()))
    assert torch.isfinite(ok[-1]).all()
    with pytest.raises(RuntimeError, match="LDS|too|exceeds"):
        out = m(torch.rand(2, 16, 40, 40, device=dev()))
        torch.nn.functional.cross_entropy(out[-1], torch.zeros(2, dtype=torch.int64, device=dev())).backward()


def test_ensemble_trainer_full_size_two_steps_vs_bf16_oracle():
    """The step `bench.py --workload ensemble24` times -- EnsembleTrainer at 3 x 369 x 24x24 bf16: gated forward with the
    missing-year decision on the device, dta_weighted_ce_scaled_dev, gated per-year Adam with device step counters; at
    B = 130 the plan takes the fused fp32-input first conv (390 first-conv workgroups), the lean 24x24 stage kernels and
    the paired wide-window weight gradients -- two steps, year 1 missing in the second, against the bf16-mode oracle with
    one Adam state per year (reference year.py:24-33, multi_stage.py:277-288): loss, scores, every parameter, every
    BatchNorm buffer and the per-year step counts."""
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.year import learned_ensemble
    Bt, lr = 130, 1e-3
    p = O.init_params(O.learned_ensemble_spec(YEARS, BANDS, CLASSES), seed=23)
    m = learned_ensemble(YEARS, CLASSES, {"pretrain_state_dict": None, "bands": BANDS})
    for net in m.year_models:
        net.precision = "bf16"
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
    m = m.to(dev()).train()
    w = (0.1 + (np.arange(CLASSES) % 5)).astype(np.float32)
    tr = EnsembleTrainer(m, lr=lr, loss_weight=torch.from_numpy(w))
    states = [dict() for _ in range(YEARS)]
    y = prng.randint(77, 3, (Bt,), CLASSES)
    yt = torch.from_numpy(y).to(dev())
    for step in range(2):
        imgs = [prng.uniform01(500 + step, yy, (Bt, BANDS, HW, HW)) for yy in range(YEARS)]
        if step == 1:
            imgs[1] = np.zeros_like(imgs[1])
        loss = float(tr.train_step([torch.from_numpy(a).to(dev()) for a in imgs], yt))      # present=None: decided on the device
        scores = tr.scores.cpu().numpy()
        O.bf16_mode(True)
        try:
            q_scores, cache, upd = O.learned_ensemble_fwd(p, imgs, True, np.float64)
            q_loss, dl = O.weighted_cross_entropy(q_scores, y, w)
            g = O.learned_ensemble_bwd(p, cache, dl.astype(np.float64), np.float64)
        finally:
            O.bf16_mode(False)
        assert abs(loss - q_loss) / q_loss < 2e-3, (step, loss, q_loss)
        assert rel_l2(scores, q_scores) < (1e-3 if step == 0 else 6e-3), step
        for yy in range(YEARS):
            # (conv biases under batch-statistics BatchNorm: zero gradient analytically -- exact zero on the HIP path, ~1e-17
            #  noise in the oracle, which Adam would turn into +-lr steps of the bias and so of the running means)
            gy = {k: (np.zeros_like(v) if k.endswith("conv_layer.bias") else v) for k, v in g.items() if k.startswith(f"year_models.{yy}.")}
            if gy:
                p = O.adam_step(p, gy, states[yy], lr=lr)      # a skipped year: no gradient, no moment decay, no step
        p.update(upd)
        del cache, g
    sd = m.state_dict()
    for yy in range(YEARS):
        num = den = 0.0
        for k, v in p.items():
            if not k.startswith(f"year_models.{yy}."):
                continue
            if k.endswith("num_batches_tracked"):
                assert int(sd[k]) == (1 if yy == 1 else 2), k
                continue
            if k.endswith("conv_layer.bias"):
                continue   # zero gradient analytically: Adam's sign(noise) updates are not comparable
            a, b = sd[k].double().cpu().numpy(), np.asarray(v, np.float64)
            if O.is_buffer(k):
                # after step 1 Adam has moved every weight by ~lr with the SIGN of its gradient: elements whose tiny gradients
                # differ in the last bits move the other way, so step 2's batch statistics see slightly different weights
                # (observed 7e-3 on a running mean; the one-step buffer check at 2e-3 is the test above)
                assert rel_l2(a, b) < 1e-2, k
                continue
            num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum())
        print(f"config 5 trainer, year {yy}: parameters after 2 bf16 steps vs per-year oracle Adam: rel-L2 {np.sqrt(num / den):.2e}")
        # (3e-3 at 11x11, tests/test_hip_benched_path.py; the 576-pixel maps' gradients sit 1.5e-2 from the oracle's element-wise
        #  -- first test of this file -- so more of Adam's sign-like first steps go the other way: observed 3.8e-3)
        assert np.sqrt(num / den) < 6e-3, yy
    assert tr.step_counts() == [2, 1, 2]
