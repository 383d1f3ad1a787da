"""The missing-year decision of the year ensemble (reference src/models/year.py:27: a year whose whole batch tensor sums
to zero is skipped) taken ON THE DEVICE: dta_year_flags + dta_ensemble_forward_gated + device-scaled loss gradient + gated
Adam with device step counters.  Reference behaviour of a skipped year: not in the mean, BatchNorm running statistics and
num_batches_tracked untouched, no gradient, no optimizer step (torch's Adam passes over grad None).  The device-decided
step must equal the step driven by host-side `present` flags, which the reference golden pins (test_hip_modules)."""
import copy
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import prng

pytestmark = pytest.mark.gpu

BANDS, CLASSES, B, YEARS = 20, 7, 6, 3


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def test_year_flags_kernel():
    from deeptreeattention_amd import _lib
    L = _lib.lib()
    n = 5 * 20 * 11 * 11 + 3          # not a multiple of 4: the scalar tail is checked too
    xs = [torch.zeros(n + 1, device=dev())[:n] for _ in range(4)]      # (views at offset 0: 16-byte aligned)
    xs[1][n - 1] = 1e-30              # one tiny element at the very end
    xs[2][17] = float("nan")          # NaN != 0: kept, as `nan == 0` is False in the reference's test
    xs[3][5] = -0.0                   # negative zero is zero
    flags = torch.full((4,), 7.0, device=dev())
    ptrs = (C.c_void_p * 4)(*[x.data_ptr() for x in xs])
    _lib.check(L.dta_year_flags(ptrs, 4, n, _lib.ptr(flags), None, _lib.current_stream_ptr()), "dta_year_flags")
    assert flags.tolist() == [0.0, 1.0, 1.0, 0.0]
    assert L.dta_year_flags(ptrs, 9, n, _lib.ptr(flags), None, None) != 0          # more years than a grouped launch takes


def _images(step, zero=()):
    imgs = [torch.from_numpy(prng.uniform01(700 + 10 * step, yy, (B, BANDS, 11, 11))).to(dev()) for yy in range(YEARS)]
    for z in zero:
        imgs[z].zero_()
    return imgs


ZEROS = [(), (1,), (0, 2), (), (2,)]      # which years are missing in each step


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_device_decided_steps_equal_host_flag_steps(precision):
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.year import learned_ensemble
    import deeptreeattention_amd
    old = deeptreeattention_amd.get_default_precision()
    deeptreeattention_amd.set_default_precision(precision)
    try:
        torch.manual_seed(31)
        a = learned_ensemble(YEARS, CLASSES, {"pretrain_state_dict": None, "bands": BANDS}).to(dev()).train()
    finally:
        deeptreeattention_amd.set_default_precision(old)
    b = copy.deepcopy(a)
    ta, tb = EnsembleTrainer(a, lr=1e-3), EnsembleTrainer(b, lr=1e-3)
    for step, zero in enumerate(ZEROS):
        imgs = _images(step, zero)
        y = torch.from_numpy(prng.randint(700 + step, 5, (B,), CLASSES)).to(dev())
        la = ta.train_step(imgs, y)                                                   # decided on the device
        lb = tb.train_step(imgs, y, present=[i not in zero for i in range(YEARS)])    # host flags: only kept years launched
        assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(lb)), step
    assert ta.step_counts() == tb.step_counts() == [4, 4, 3]
    sa, sb = a.state_dict(), b.state_dict()
    for k in sb:
        assert torch.allclose(sa[k].float(), sb[k].float(), rtol=2e-6, atol=1e-8), k
    assert int(sa["year_models.2.conv1.bn1.num_batches_tracked"]) == 3       # skipped twice out of five steps
    # validation / prediction take the same decision without a host round trip
    imgs = _images(9, (1,))
    y = torch.from_numpy(prng.randint(709, 5, (B,), CLASSES)).to(dev())
    s1, l1 = ta.forward_loss(imgs, y)
    s2, l2 = tb.forward_loss(imgs, y, present=[True, False, True])
    assert torch.allclose(s1, s2, rtol=1e-6, atol=1e-7) and abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l2))
    from deeptreeattention_amd.engine import Predictor
    p1 = Predictor(a)(imgs)[0].clone()
    p2 = Predictor(b)(imgs, True, [True, False, True])[0].clone()
    assert torch.allclose(p1, p2, rtol=1e-5, atol=1e-7)


def test_switching_between_device_and_host_decisions_keeps_the_step_counts():
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.year import learned_ensemble
    torch.manual_seed(32)
    a = learned_ensemble(YEARS, CLASSES, {"pretrain_state_dict": None, "bands": BANDS}).to(dev()).train()
    b = copy.deepcopy(a)
    ta, tb = EnsembleTrainer(a, lr=1e-3), EnsembleTrainer(b, lr=1e-3)
    for step, zero in enumerate(ZEROS):
        imgs = _images(20 + step, zero)
        y = torch.from_numpy(prng.randint(720 + step, 5, (B,), CLASSES)).to(dev())
        present = [i not in zero for i in range(YEARS)]
        ta.train_step(imgs, y, None if step % 2 == 0 else present)      # alternates: counters move device <-> host
        tb.train_step(imgs, y, present)
    assert ta.step_counts() == tb.step_counts() == [4, 4, 3]
    sa, sb = a.state_dict(), b.state_dict()
    for k in sb:
        assert torch.allclose(sa[k].float(), sb[k].float(), rtol=2e-6, atol=1e-8), k


def test_no_year_present_gives_nan_loss_not_an_update():
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.year import learned_ensemble
    torch.manual_seed(33)
    a = learned_ensemble(YEARS, CLASSES, {"pretrain_state_dict": None, "bands": BANDS}).to(dev()).train()
    before = {k: v.clone() for k, v in a.state_dict().items()}
    tr = EnsembleTrainer(a, lr=1e-3)
    y = torch.from_numpy(prng.randint(740, 5, (B,), CLASSES)).to(dev())
    loss = tr.train_step(_images(40, (0, 1, 2)), y)
    assert torch.isnan(loss)                                   # (the reference raises: nothing to average, year.py:33)
    assert tr.step_counts() == [0, 0, 0]
    for k, v in a.state_dict().items():
        assert torch.equal(v, before[k]), k
