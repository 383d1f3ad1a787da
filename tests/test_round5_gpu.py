"""Round 5: regressions for the round-4 advisor findings on the drop-in module path (optim.DtaAdam around the UNCHANGED
reference step, src/main.py:71-80,135-149; year ensembles src/models/year.py:24-33, multi_stage.py:258-288):

* a learned_ensemble on its host-decided path (more than DTA_MAX_YEARS years) is stepped by DtaAdam, year by year, exactly
  as torch.optim.Adam steps it;
* a forward that builds no graph (validation metrics) between a training forward and its backward does not disturb the
  year flags the backward and the optimizer read; gradient accumulation steps a year that ANY micro-batch kept;
* zero_grad() after step() + backward() really clears (discarding a batch does not leak its gradients into the next);
* optim.cross_entropy's backward can run twice (retain_graph)."""
import copy

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _ensemble(years, bands=12, classes=5, seed=3):
    from deeptreeattention_amd.year import learned_ensemble
    torch.manual_seed(seed)
    return learned_ensemble(years, classes, {"pretrain_state_dict": None, "bands": bands}).to(dev()).train()


def _batch(years, B=4, bands=12, classes=5, seed=0, zero=()):
    g = torch.Generator(device=dev())
    g.manual_seed(100 + seed)
    xs = [torch.rand(B, bands, 11, 11, device=dev(), generator=g) for _ in range(years)]
    for i in zero:
        xs[i].zero_()
    return xs, torch.randint(0, classes, (B,), device=dev(), generator=g)


@pytest.mark.parametrize("beyond", [False, True])
def test_dta_adam_steps_a_many_year_ensemble_like_torch_adam(beyond):
    """More years than one grouped launch takes (DTA_MAX_YEARS = 16 since round 6; 4 before): learned_ensemble.forward takes
    its host-decided path (year.py:27 on the host, the kept years in chunks of DTA_MAX_YEARS).  DtaAdam must step the kept
    years and pass over the skipped ones -- torch.optim.Adam on a copy of the model is the yardstick (a skipped year has grad
    None there).  beyond=False: five years, which one device-gated launch chain now takes."""
    from deeptreeattention_amd.optim import DtaAdam
    from deeptreeattention_amd import _lib
    years = _lib.MAX_YEARS + 1 if beyond else 5
    a = _ensemble(years)
    b = copy.deepcopy(a)
    opt_a = DtaAdam(a.parameters(), lr=1e-3)
    opt_b = torch.optim.Adam(b.parameters(), lr=1e-3)
    before = {k: v.detach().clone() for k, v in a.named_parameters()}
    for step, zero in enumerate([(), (1,), (0, 4)]):
        xs, y = _batch(years, seed=step, zero=zero)
        for m, opt in ((a, opt_a), (b, opt_b)):
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(m(xs), y).backward()
            opt.step()
    sa, sb = a.state_dict(), b.state_dict()
    moved = 0
    for k in sb:
        if k.endswith("num_batches_tracked"):
            assert int(sa[k]) == int(sb[k]), k
            continue
        if k.endswith("conv_layer.bias"):
            continue        # zero gradient analytically: Adam's sign(noise) steps are not comparable
        assert rel_l2(sa[k].float().cpu().numpy(), sb[k].float().cpu().numpy()) < 2e-4, k
        if k in before and "classifier1" not in k and "classifier2" not in k:
            moved += int(not torch.equal(sa[k], before[k]))
    assert moved > 100                                     # the ensemble WAS stepped (the bug left every parameter where it was)
    assert opt_a.step_counts() == [2, 2, 3, 3, 2] + [3] * (years - 5)


def test_no_grad_forward_between_training_forward_and_backward_keeps_the_flags():
    """train forward -> validation-style forward under no_grad (other inputs, another year missing) -> backward -> step:
    the backward and the optimizer must see the TRAINING forward's flags (the two-bank scheme handed them the cleared
    bank: exact-zero gradients for every year and a silently skipped step)."""
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    years = 3
    a = _ensemble(years)
    b = copy.deepcopy(a)
    opt_a, opt_b = DtaAdam(a.parameters(), lr=1e-3), DtaAdam(b.parameters(), lr=1e-3)
    xs, y = _batch(years, seed=1, zero=(2,))
    xv, _ = _batch(years, seed=2, zero=(0,))
    # a: with the interleaved no_grad forward; b: without
    opt_a.zero_grad()
    loss = cross_entropy(a(xs), y)
    with torch.no_grad():
        a(xv)
        a(xv)
    loss.backward()
    assert float(a.year_models[0].conv1.conv_layer.weight.grad.abs().sum()) > 0
    assert float(a.year_models[2].conv1.conv_layer.weight.grad.abs().sum()) == 0.0      # the year the TRAINING batch lacks
    opt_a.step()
    opt_b.zero_grad()
    cross_entropy(b(xs), y).backward()
    opt_b.step()
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.equal(p, q), k
    assert opt_a.step_counts() == opt_b.step_counts() == [1, 1, 0]


def test_gradient_accumulation_steps_a_year_any_micro_batch_kept():
    """Two micro-batches before one step: year 1 present only in the first, year 2 only in the second.  Both are stepped
    (torch: both have a gradient), year 0 -- present in both -- too; a year missing from both is passed over."""
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    years = 4
    a = _ensemble(years)
    opt = DtaAdam(a.parameters(), lr=1e-3)
    before = {k: v.detach().clone() for k, v in a.named_parameters()}
    opt.zero_grad()
    xs, y = _batch(years, seed=3, zero=(2, 3))
    cross_entropy(a(xs), y).backward()
    xs, y = _batch(years, seed=4, zero=(1, 3))
    cross_entropy(a(xs), y).backward()
    opt.step()
    assert opt.step_counts() == [1, 1, 1, 0]
    for k, p in a.named_parameters():
        if k.endswith("conv1.conv_layer.weight"):
            yy = int(k.split(".")[1])
            assert torch.equal(p, before[k]) == (yy == 3), k


@pytest.mark.parametrize("fuse", [True, False])
def test_zero_grad_after_step_and_backward_really_clears(fuse):
    """step(); backward() [a batch that is then discarded]; zero_grad(); backward(); the gradient must be the LAST batch's
    alone (a stale "cleared by step" flag skipped the clear and the in-place write summed both batches)."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    torch.manual_seed(5)
    m = H.spectral_network(12, 5).to(dev()).train()
    opt = DtaAdam(m.parameters(), lr=1e-3, fuse_zero_grad=fuse)
    g = torch.Generator(device=dev()); g.manual_seed(9)
    xa, xb = (torch.rand(4, 12, 11, 11, device=dev(), generator=g) for _ in range(2))
    y = torch.randint(0, 5, (4,), device=dev(), generator=g)
    opt.zero_grad()
    cross_entropy(m(xa)[-1], y).backward()
    opt.step()
    cross_entropy(m(xa)[-1], y).backward()         # the batch that gets discarded
    opt.zero_grad()
    cross_entropy(m(xb)[-1], y).backward()
    got = m.conv2.conv_layer.weight.grad.clone()
    ref = copy.deepcopy(m)
    for p in ref.parameters():
        p.grad = None
    for mod in (m, ref):
        for bn in (mod.conv1.bn1, mod.conv2.bn1, mod.conv3.bn1):
            bn.reset_running_stats()
    torch.nn.functional.cross_entropy(ref(xb)[-1], y).backward()
    assert rel_l2(got.cpu().numpy(), ref.conv2.conv_layer.weight.grad.cpu().numpy()) < 1e-5


def test_zero_grad_clears_gradients_autograd_accumulated_into_the_views():
    """Parameters whose gradients arrive through autograd's own accumulation (a plain torch module sharing the optimizer)
    leave no trace in take_inplace; with fuse_zero_grad=False the buffer stays dirty after step() and zero_grad() must
    clear it."""
    from deeptreeattention_amd.optim import DtaAdam
    torch.manual_seed(6)
    lin = torch.nn.Linear(8, 3).to(dev())
    opt = DtaAdam(lin.parameters(), lr=1e-2, fuse_zero_grad=False)
    x = torch.rand(5, 8, device=dev())
    opt.zero_grad()
    lin(x).sum().backward()
    g1 = lin.weight.grad.clone()
    opt.step()
    opt.zero_grad()
    assert float(lin.weight.grad.abs().sum()) == 0.0
    lin(x).sum().backward()
    assert torch.allclose(lin.weight.grad, g1)


def test_cross_entropy_backward_twice_with_retain_graph():
    from deeptreeattention_amd.optim import cross_entropy
    torch.manual_seed(7)
    z = torch.randn(6, 9, device=dev(), requires_grad=True)
    y = torch.randint(0, 9, (6,), device=dev())
    loss = cross_entropy(z, y)
    g1, = torch.autograd.grad(loss, z, retain_graph=True)
    g2, = torch.autograd.grad(loss, z)
    zr = z.detach().clone().requires_grad_(True)
    gr, = torch.autograd.grad(torch.nn.functional.cross_entropy(zr, y), zr)
    assert torch.equal(g1, g2)
    assert rel_l2(g1.cpu().numpy(), gr.cpu().numpy()) < 1e-5


def test_ensemble_forward_loss_in_one_call_equals_forward_then_loss():
    """dta_ensemble_forward_loss (EnsembleTrainer.forward_loss / train_step: the mean over the kept years formed inside the
    loss launch) against the module path's two calls (learned_ensemble.forward -> k_mean_scores, then optim.cross_entropy ->
    k_blend_ce): the same float operations in the same order, so the scores and the loss have the same bits -- all years
    present, one year missing (device-decided), and with `present=` flags."""
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.optim import cross_entropy
    years = 3
    a = _ensemble(years)
    w = (0.1 + (torch.arange(5) % 3)).float().to(dev())
    tr = EnsembleTrainer(a, lr=1e-3, loss_weight=w)
    for zero, present in (((), None), ((1,), None), ((2,), [True, True, False])):
        xs, y = _batch(years, seed=11 + len(zero), zero=zero)
        scores, loss = tr.forward_loss(xs, y, present)
        with torch.no_grad():
            ref_scores = a(xs)
        ref_loss = cross_entropy(ref_scores, y, weight=w)
        assert torch.equal(scores, ref_scores), zero
        assert torch.equal(loss, ref_loss), (zero, float(loss), float(ref_loss))


def test_eval_mode_forward_with_grad_does_not_mark_years_for_the_next_step():
    """model.eval() forward with grad enabled (a saliency map; a validation loop without no_grad) that keeps a
    year must NOT make DtaAdam step that year when the TRAINING batch lacks it: the reference passes over a skipped year
    (grad None), a step on zero gradients would still decay its moments and move it by its momentum."""
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    years = 3
    a = _ensemble(years)
    opt = DtaAdam(a.parameters(), lr=1e-3)
    xs, y = _batch(years, seed=1)
    opt.zero_grad(); cross_entropy(a(xs), y).backward(); opt.step()          # every year has momentum now
    before = {k: v.detach().clone() for k, v in a.named_parameters()}
    xv, _ = _batch(years, seed=2)                     # validation batch: all years present
    a.eval()
    s = a(xv)                                         # grad enabled, eval mode
    s.sum().backward()                                # (its own backward still sees its own flags)
    a.train()
    xs, y = _batch(years, seed=3, zero=(2,))          # the training batch lacks year 2
    opt.zero_grad(); cross_entropy(a(xs), y).backward(); opt.step()
    assert opt.step_counts() == [2, 2, 1]
    for k, p in a.named_parameters():
        if k.startswith("year_models.2."):
            assert torch.equal(p, before[k]), k


# ---- DtaAdam under data parallelism passes over an 'other' parameter NO rank produced a gradient for --------
def _others_worker(rank, world, port, out):
    import datetime
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    d = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(5)
    m = H.spectral_network(12, 5).to(d).train()
    # a parameter DtaAdam cannot keep in its float32 flat buffers (float64, more than one element): stepped by `_step_others`
    extra = torch.nn.Parameter(torch.full((7,), 0.5, device=d, dtype=torch.float64))
    opt = DtaAdam(list(m.parameters()) + [extra], lr=1e-2, exchange="torch")
    g = torch.Generator(device=d); g.manual_seed(40 + rank)
    hist = []
    # step 0: every rank uses `extra`; step 1: nobody does (grad None everywhere -> passed over, as torch.optim.Adam / DDP do);
    # step 2: only rank 1 does (rank 0 sends zeros: the mean gradient is half of rank 1's)
    for step, use in enumerate([True, False, rank == 1]):
        x = torch.rand(4, 12, 11, 11, device=d, generator=g)
        y = torch.randint(0, 5, (4,), device=d, generator=g)
        opt.zero_grad(set_to_none=True)
        loss = cross_entropy(m(x)[-1], y)
        if use:
            loss = loss + (extra * extra).sum().float()
        loss.backward()
        opt.step()
        hist.append(extra.detach().cpu().numpy().copy())
    out[rank] = (hist, int(opt.state[extra]["step"]) if extra in opt.state else 0)
    opt.close()
    dist.destroy_process_group()


def test_dta_adam_dp_passes_over_an_other_parameter_without_any_gradient():
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_others_worker, args=(2, port, out), nprocs=2, join=True)
    (h0, n0), (h1, n1) = out[0], out[1]
    for a, b in zip(h0, h1):
        assert np.array_equal(a, b)                 # replicas stay identical
    assert np.all(h0[0] < 0.5)                      # step 0: stepped
    assert np.array_equal(h0[1], h0[0])             # step 1: NO rank had a gradient -> untouched (no momentum move, no decay)
    assert np.all(h0[2] < h0[1])                    # step 2: one rank's gradient is enough
    assert n0 == n1 == 2                            # two steps counted, not three
