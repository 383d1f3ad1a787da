"""The specialised stage kernels against the generic ones on the same inputs.

The 11x11 networks run the "lean" stage kernels (persistent workgroups that walk patch batches with a register prefetch
of the next one, register-level column sums, four patches per workgroup in the last stage); DTA_NO_LEAN=1 routes the same
step through the generic per-patch kernels.  Both implement the same reference lines (Hang2020.py:24-31, :105-124,
:149-168), so losses and gradients must agree -- exactly up to summation order in fp32 mode, up to the storage formats of
the intermediate maps in bf16 mode -- at batch sizes that leave the persistent grid with ragged last rounds, a partly
filled four-patch workgroup, or a single patch."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_developer_switch_variants_in_the_developer_library():
    """Every test marked `devlib` (this file and tests/test_hip_parity.py) compares an alternative launch plan, selected by
    a developer switch, with the default plan.  The product library has no switches (it reads nothing from the environment),
    so those tests are skipped in this process and run HERE, in a process that loads libdta_hip_dev.so: the same sources
    compiled with -DDTA_DEV_SWITCHES (same dta_build_id)."""
    from deeptreeattention_amd import _lib
    L = _lib.lib()
    if L.dta_dev_switches_enabled():
        pytest.skip("already running in the developer library")
    env = dict(os.environ, DTA_DEV_LIB="1")
    probe = subprocess.run([sys.executable, "-c", "from deeptreeattention_amd import _lib; L = _lib.lib(); "
                            "print(L.dta_dev_switches_enabled(), L.dta_build_id().decode())"], cwd=REPO, env=env,
                           capture_output=True, text=True)
    assert probe.returncode == 0, probe.stderr[-2000:]
    enabled, build_id = probe.stdout.split()
    assert enabled == "1" and build_id == L.dta_build_id().decode(), (probe.stdout, "libdta_hip_dev.so is stale: python -m deeptreeattention_amd.build")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu and devlib", "-p", "no:cacheprovider",
                        os.path.join(REPO, "tests", "test_kernel_variants_gpu.py"), os.path.join(REPO, "tests", "test_hip_parity.py")],
                       cwd=REPO, env=env, capture_output=True, text=True)
    tail = r.stdout[-3000:] + r.stderr[-1000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], tail
    print(r.stdout.splitlines()[-1])


def _step(m, x, y):
    m.zero_grad(set_to_none=True)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.reset_running_stats()
    out = m(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    return float(loss.detach()), out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


@pytest.mark.devlib
@pytest.mark.parametrize("precision,B", [("fp32", 2), ("fp32", 5), ("fp32", 1031), ("bf16", 3), ("bf16", 257), ("bf16", 2051)])
def test_lean_stage_kernels_match_generic_kernels(precision, B, monkeypatch, devlib):
    from deeptreeattention_amd import Hang2020 as H, _lib
    torch.manual_seed(B)
    bands, classes = 24, 11
    m = H.Hang2020(bands, classes, precision=precision).cuda().train()
    x = torch.rand(B, bands, 11, 11, device="cuda")
    y = torch.randint(0, classes, (B,), device="cuda")
    L = _lib.lib()
    monkeypatch.delenv("DTA_NO_LEAN", raising=False)
    L.dta_dev_reload_switches()
    l1, o1, g1 = _step(m, x, y)
    monkeypatch.setenv("DTA_NO_LEAN", "1")
    L.dta_dev_reload_switches()
    try:
        l2, o2, g2 = _step(m, x, y)
    finally:
        monkeypatch.delenv("DTA_NO_LEAN", raising=False)
        L.dta_dev_reload_switches()
    fp32 = precision == "fp32"
    assert torch.isfinite(o1).all() and abs(l1 - l2) < (1e-5 if fp32 else 2e-3) * max(1.0, abs(l2))
    assert rel_l2(o1.cpu().numpy(), o2.cpu().numpy()) < (1e-5 if fp32 else 5e-3)
    assert g1.keys() == g2.keys()
    for k in g2:
        n2 = float(g2[k].norm())
        # conv biases (zero under batch-statistics BatchNorm) and the gates' output biases are cancelling sums: noise
        if n2 == 0 or k.endswith("conv_layer.bias") or k.endswith("attention_conv2.bias"):
            continue
        if not fp32 and g2[k].numel() == 1:     # one-element bias sums over bf16-stored maps: in the whole-vector check below
            continue
        assert rel_l2(g1[k].cpu().numpy(), g2[k].cpu().numpy()) < (2e-4 if fp32 else 3e-2), k
    # the whole gradient as one vector
    keep = [k for k in g2 if not (k.endswith("conv_layer.bias") or k.endswith("attention_conv2.bias"))]
    a = torch.cat([g1[k].reshape(-1).double() for k in keep]); b = torch.cat([g2[k].reshape(-1).double() for k in keep])
    assert float((a - b).norm() / b.norm()) < (5e-5 if fp32 else 1e-2)


@pytest.mark.devlib
@pytest.mark.parametrize("B", [1, 3, 130])
def test_lean_stage_kernels_for_24x24_crops_match_generic_kernels(B, monkeypatch, devlib):
    """BASELINE configs[4] geometry (spectral network on 24x24 crops, bf16 mode): the lean stage kernels run with several
    (pixel, octet) items per thread (5 / 3 / 2 for the three stages) and without the patch image in LDS; the generic
    kernels on the same step must give the same loss, head scores and gradients up to the bf16 storage of the
    intermediate maps.  B = 130 also takes the fused fp32-input first conv (>= 100 workgroups)."""
    from deeptreeattention_amd import Hang2020 as H, _lib
    torch.manual_seed(40 + B)
    bands, classes = 24, 9
    m = H.spectral_network(bands, classes, precision="bf16").cuda().train()
    x = torch.rand(B, bands, 24, 24, device="cuda")
    y = torch.randint(0, classes, (B,), device="cuda")

    def step():
        m.zero_grad(set_to_none=True)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.reset_running_stats()
        heads = m(x)
        loss = sum(torch.nn.functional.cross_entropy(h, y) for h in heads)      # all three heads: every stage's feature path
        loss.backward()
        return float(loss.detach()), [h.detach().clone() for h in heads], {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    L = _lib.lib()
    monkeypatch.delenv("DTA_NO_LEAN", raising=False)
    L.dta_dev_reload_switches()
    l1, o1, g1 = step()
    monkeypatch.setenv("DTA_NO_LEAN", "1")
    L.dta_dev_reload_switches()
    try:
        l2, o2, g2 = step()
    finally:
        monkeypatch.delenv("DTA_NO_LEAN", raising=False)
        L.dta_dev_reload_switches()
    assert abs(l1 - l2) < 2e-3 * max(1.0, abs(l2))
    for a, b in zip(o1, o2):
        assert torch.isfinite(a).all() and rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 5e-3
    assert g1.keys() == g2.keys()
    keep = [k for k in g2 if not (k.endswith("conv_layer.bias") or k.endswith("attention_conv2.bias"))]
    a = torch.cat([g1[k].reshape(-1).double() for k in keep]); b = torch.cat([g2[k].reshape(-1).double() for k in keep])
    assert float((a - b).norm() / b.norm()) < (1e-2 if B > 1 else 3e-2)
    for k in keep:
        if g2[k].numel() >= 1000:
            assert abs(float(g1[k].norm()) - float(g2[k].norm())) <= 1e-2 * float(g2[k].norm()), k


@pytest.mark.devlib
@pytest.mark.parametrize("precision,B", [("fp32", 3), ("fp32", 64), ("bf16", 70), ("bf16", 530)])
def test_in_launch_fanin_of_batchnorm_sums_matches_the_finalize_launches(precision, B, monkeypatch, devlib):
    """DTA_FANIN=3 (a measured experiment, off by default: profiles/README.md round 4) folds BatchNorm's batch statistics
    and backward batch sums inside the conv / stage launches (logical fan-in groups, kernels.h) instead of the finalize
    launches.  Same sums in another order: loss, scores, gradients and BatchNorm buffers must agree -- including launches
    of fewer workgroups than fan-in groups (B = 3)."""
    from deeptreeattention_amd import Hang2020 as H, _lib
    torch.manual_seed(B)
    bands, classes = 24, 11
    m = H.Hang2020(bands, classes, precision=precision).cuda().train()
    x = torch.rand(B, bands, 11, 11, device="cuda")
    y = torch.randint(0, classes, (B,), device="cuda")
    L = _lib.lib()
    monkeypatch.delenv("DTA_FANIN", raising=False)
    L.dta_dev_reload_switches()
    l1, o1, g1 = _step(m, x, y)
    b1 = {k: v.clone() for k, v in m.named_buffers()}
    monkeypatch.setenv("DTA_FANIN", "3")
    L.dta_dev_reload_switches()
    try:
        l2, o2, g2 = _step(m, x, y)
        b2 = {k: v.clone() for k, v in m.named_buffers()}
    finally:
        monkeypatch.delenv("DTA_FANIN", raising=False)
        L.dta_dev_reload_switches()
    tol = 1e-5 if precision == "fp32" else 2e-3
    assert abs(l1 - l2) < tol * max(1.0, abs(l1)) and rel_l2(o2.cpu().numpy(), o1.cpu().numpy()) < tol
    assert g1.keys() == g2.keys()
    num = den = 0.0
    for k in g1:
        if k.endswith("conv_layer.bias"):
            continue
        a, b = g2[k].double().cpu().numpy(), g1[k].double().cpu().numpy()
        num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum())
    assert (num / den) ** 0.5 < (2e-5 if precision == "fp32" else 5e-3)
    for k in b1:
        assert rel_l2(b2[k].double().cpu().numpy(), b1[k].double().cpu().numpy()) < 1e-5, k


@pytest.mark.devlib
@pytest.mark.parametrize("precision,B", [("fp32", 3), ("fp32", 130), ("bf16", 5), ("bf16", 1026)])
def test_fused_forward_tail_matches_the_separate_launches(precision, B, monkeypatch, devlib):
    """The fused forward tail (stage.hip k_tail_fwd: third stage of both branches + the two last heads + blend + weighted CE in
    one launch) against the launches it replaces (developer switch DTA_NO_TAIL: k_stage_fwd_lean<128,5,5>, k_gemm_group,
    k_blend_ce): the same reference lines (Hang2020.py:24-31, :105-124, :149-168, :55-66, :256-261; src/main.py:78), sums in
    another order -- scores, loss, every gradient and the parameters after two fused steps agree to float32 reordering noise
    (bf16 mode: the tail itself computes in float32 on both routes); ragged batches leave slots of the last workgroup empty."""
    import copy
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    torch.manual_seed(B)
    bands, classes = 24, 11
    m0 = H.Hang2020(bands, classes, precision=precision).cuda().train()
    x = torch.rand(B, bands, 11, 11, device="cuda")
    y = torch.randint(0, classes, (B,), device="cuda")
    w = (0.1 + (torch.arange(classes) % 5)).float()
    L = devlib

    def run():
        m = copy.deepcopy(m0)
        tr = FusedTrainer(m, lr=1e-3, loss_weight=w, keep_grads=True)
        l1 = float(tr.train_step(x, y))
        out = tr.logits.clone()
        g = {k: tr.grad_of(p).clone() for k, p in m.named_parameters() if p.dtype == torch.float32}
        l2 = float(tr.train_step(x, y))
        return l1, l2, out, g, {k: v.clone() for k, v in m.state_dict().items()}

    monkeypatch.delenv("DTA_NO_TAIL", raising=False)
    L.dta_dev_reload_switches()
    a = run()
    monkeypatch.setenv("DTA_NO_TAIL", "1")
    L.dta_dev_reload_switches()
    try:
        b = run()
    finally:
        monkeypatch.delenv("DTA_NO_TAIL", raising=False)
        L.dta_dev_reload_switches()
    tol = 2e-5 if precision == "fp32" else 2e-3
    assert abs(a[0] - b[0]) < tol * max(1.0, abs(b[0])) and abs(a[1] - b[1]) < 5 * tol * max(1.0, abs(b[1]))
    assert rel_l2(a[2].cpu().numpy(), b[2].cpu().numpy()) < tol
    num = den = 0.0
    for k in b[3]:
        if k.endswith("conv_layer.bias"):
            continue
        ga, gb = a[3][k].double().cpu().numpy(), b[3][k].double().cpu().numpy()
        num += float(((ga - gb) ** 2).sum()); den += float((gb ** 2).sum())
    assert (num / den) ** 0.5 < (2e-5 if precision == "fp32" else 5e-3)
    # parameters after the two steps: the large tensors (Adam's first steps move every element by ~lr against the SIGN of its
    # gradient, so the handful of elements whose gradient is reordering noise may go the other way: small tensors of such
    # elements -- zero-gradient biases -- are not comparable; the second step's loss above covers them)
    for k in b[4]:
        if b[4][k].numel() < 1000 or not b[4][k].dtype.is_floating_point:
            continue
        assert rel_l2(a[4][k].double().cpu().numpy(), b[4][k].double().cpu().numpy()) < (2e-3 if precision == "fp32" else 5e-3), k


@pytest.mark.devlib
@pytest.mark.parametrize("precision,B", [("fp32", 3), ("fp32", 130), ("bf16", 5), ("bf16", 1026)])
def test_lead_workgroups_match_the_finalize_launches(precision, B, monkeypatch, devlib):
    """BatchNorm statistics as LEAD workgroups of the consuming launch (stage.hip bn_lead_block / bn_lead_wait: developer
    switch DTA_LEAD=1, an experiment that removes two launches and measured -1 us per step) against the separate
    k_bn_finalize launches (the default): the same sums in double, other slice counts -- coefficients, running statistics
    (Hang2020.py:24-31 BatchNorm2d in train mode), loss, scores and gradients agree to rounding of the last float32 bit of
    a coefficient."""
    import copy
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    torch.manual_seed(B + 1)
    bands, classes = 24, 11
    m0 = H.Hang2020(bands, classes, precision=precision).cuda().train()
    x = torch.rand(B, bands, 11, 11, device="cuda")
    y = torch.randint(0, classes, (B,), device="cuda")
    L = devlib

    def run():
        m = copy.deepcopy(m0)
        tr = FusedTrainer(m, lr=1e-3, keep_grads=True)
        l1 = float(tr.train_step(x, y))
        out = tr.logits.clone()
        g = {k: tr.grad_of(p).clone() for k, p in m.named_parameters() if p.dtype == torch.float32}
        st = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
        return l1, out, g, st

    monkeypatch.delenv("DTA_LEAD", raising=False)
    L.dta_dev_reload_switches()
    b = run()
    monkeypatch.setenv("DTA_LEAD", "1")
    L.dta_dev_reload_switches()
    try:
        a = run()
    finally:
        monkeypatch.delenv("DTA_LEAD", raising=False)
        L.dta_dev_reload_switches()
    tol = 2e-6 if precision == "fp32" else 2e-3
    assert abs(a[0] - b[0]) < tol * max(1.0, abs(b[0]))
    assert rel_l2(a[1].cpu().numpy(), b[1].cpu().numpy()) < tol
    for k in b[3]:
        if k.endswith("num_batches_tracked"):
            assert int(a[3][k]) == int(b[3][k]) == 1, k
        else:
            assert rel_l2(a[3][k].double().cpu().numpy(), b[3][k].double().cpu().numpy()) < (1e-6 if precision == "fp32" else 2e-3), k
    num = den = 0.0
    for k in b[2]:
        if k.endswith("conv_layer.bias"):
            continue
        ga, gb = a[2][k].double().cpu().numpy(), b[2][k].double().cpu().numpy()
        num += float(((ga - gb) ** 2).sum()); den += float((gb ** 2).sum())
    assert (num / den) ** 0.5 < (2e-5 if precision == "fp32" else 5e-3)


@pytest.mark.devlib
def test_lead_workgroups_in_a_year_ensemble_keep_a_skipped_years_statistics(monkeypatch, devlib):
    """Three spectral year models in one grouped launch, the second year's input all zero (year.py:27: skipped): the lead
    workgroups honour the device-side gate exactly as k_bn_finalize does -- the skipped year's running statistics and
    num_batches_tracked stay, the others move, and both routes agree."""
    import copy
    from deeptreeattention_amd.year import learned_ensemble
    from deeptreeattention_amd.engine import EnsembleTrainer
    torch.manual_seed(4)
    years, bands, classes, B = 3, 12, 5, 10
    m0 = learned_ensemble(years, classes, {"pretrain_state_dict": None, "bands": bands}).cuda().train()
    xs = [torch.rand(B, bands, 11, 11, device="cuda") for _ in range(years)]
    xs[1].zero_()
    y = torch.randint(0, classes, (B,), device="cuda")
    L = devlib

    def run():
        m = copy.deepcopy(m0)
        tr = EnsembleTrainer(m, lr=1e-3)
        loss = float(tr.train_step(xs, y))
        return loss, {k: v.clone() for k, v in m.state_dict().items()}

    monkeypatch.delenv("DTA_LEAD", raising=False)
    L.dta_dev_reload_switches()
    b = run()
    monkeypatch.setenv("DTA_LEAD", "1")
    L.dta_dev_reload_switches()
    try:
        a = run()
    finally:
        monkeypatch.delenv("DTA_LEAD", raising=False)
        L.dta_dev_reload_switches()
    assert abs(a[0] - b[0]) < 2e-3 * max(1.0, abs(b[0]))
    before = m0.state_dict()
    for k in b[1]:
        if "running_" in k or "num_batches" in k:
            skipped = k.startswith("year_models.1.")
            assert torch.equal(a[1][k], before[k]) == skipped, k
            if k.endswith("num_batches_tracked"):
                assert int(a[1][k]) == int(b[1][k]), k
            else:
                assert rel_l2(a[1][k].double().cpu().numpy(), b[1][k].double().cpu().numpy()) < 2e-3, k
