"""Round 4: the drop-in module path (optim.DtaAdam + optim.cross_entropy around the UNCHANGED reference step,
src/main.py:71-80,135-149; multi_stage.py:258-288), the data-parallel year ensemble with a year missing on ONE rank only,
and the peer-exchange probe."""
import copy
import datetime
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import rel_l2
from oracle import hang2020_np as O
from oracle import prng

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIGHT = 2e-4


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def load(model, p):
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
    return model.to(dev())


def test_cross_entropy_matches_torch():
    from deeptreeattention_amd.optim import cross_entropy
    torch.manual_seed(0)
    for B, classes in ((7, 5), (130, 200)):
        z = torch.randn(B, classes, device=dev(), requires_grad=True)
        z2 = z.detach().clone().requires_grad_(True)
        y = torch.randint(0, classes, (B,), device=dev())
        y[0] = -100                                       # torch's ignore_index
        w = torch.rand(classes, device=dev()) + 0.1
        for weight in (None, w):
            z.grad = z2.grad = None
            a = cross_entropy(z, y, weight=weight)
            b = torch.nn.functional.cross_entropy(z2, y, weight=weight)
            (a * 3.0).backward()
            (b * 3.0).backward()
            assert abs(float(a) - float(b)) < 1e-5 * abs(float(b))
            assert rel_l2(z.grad.cpu().numpy(), z2.grad.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_module_path_with_dta_adam_equals_the_fused_trainer(precision):
    """TreeModel.training_step unchanged (forward, cross-entropy, loss.backward(), optimizer.step()) with DtaAdam and
    optim.cross_entropy: gradients land in the optimizer's flat buffer in place, three steps give the FusedTrainer's
    parameters BIT FOR BIT (same kernels, same order; the sigmoid(alpha) blend is one unfused definition shared by the
    stand-alone blend kernel and the loss kernel) and the oracle's Adam loop."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    bands, classes, B, lr = 20, 7, 6, 1e-3
    p = O.init_params(O.hang2020_spec(bands, classes), seed=41)
    w = torch.from_numpy((0.1 + (np.arange(classes) % 7)).astype(np.float32)).to(dev())
    m1 = load(H.Hang2020(bands, classes, precision=precision), p).train()
    m2 = load(H.Hang2020(bands, classes, precision=precision), p).train()
    tr = FusedTrainer(m1, lr=lr, loss_weight=w)
    opt = DtaAdam(m2.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.75, patience=8)      # attaches like to torch's Adam
    state, pp = {}, {k: np.array(v) for k, v in p.items()}
    for step in range(3):
        x = prng.uniform01(300 + step, 1, (B, bands, 11, 11))
        y = prng.randint(300 + step, 2, (B,), classes)
        xt, yt = torch.from_numpy(x).to(dev()), torch.from_numpy(y).to(dev())
        l1 = tr.train_step(xt, yt)
        opt.zero_grad()
        l2 = cross_entropy(m2(xt), yt, weight=w)
        l2.backward()
        if step == 0:
            g = m2.spectral_network.conv1.conv_layer.weight.grad
            assert g is not None and g.data_ptr() == opt._gview[id(m2.spectral_network.conv1.conv_layer.weight)].data_ptr()
            assert float(g.abs().sum()) > 0 and float(m2.alpha.grad.abs()) > 0
            assert float(m2.spectral_network.classifier1.fc1.weight.grad.abs().sum()) == 0.0      # heads 1-2: no gradient
        opt.step()
        assert float(l1.detach()) == float(l2.detach()), step      # same kernels, same bits (the blend is ONE unfused definition)
        if precision == "fp32":
            logits, cache, upd = O.hang2020_fwd(pp, x, True, np.float64)
            _, dl = O.weighted_cross_entropy(logits, y, w.cpu().numpy())
            pp.update(upd)
            pp = O.adam_step(pp, O.hang2020_bwd(pp, cache, dl, np.float64), state, lr=lr)
    sched.step(1.0)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k             # module path == fused path, bit for bit
        a, b = sd1[k].double().cpu().numpy(), sd2[k].double().cpu().numpy()
        if precision == "fp32" and not (O.is_buffer(k) or k.endswith("conv_layer.bias")):
            assert rel_l2(b, pp[k]) < 2e-3, k             # == the oracle's Adam loop
    assert opt.step_counts() == [3]
    osd = opt.state_dict()
    assert len(osd["state"]) == len(list(m2.parameters())) and float(osd["state"][0]["step"]) == 3.0


def test_second_backward_before_step_accumulates_like_torch():
    """Gradient accumulation: the second backward before step() does not overwrite the in-place gradients -- it goes
    through autograd's accumulation into the same views."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    torch.manual_seed(2)
    m = H.spectral_network(12, 5).to(dev()).train()
    opt = DtaAdam(m.parameters(), lr=1e-3, fuse_zero_grad=False)
    x = torch.rand(4, 12, 11, 11, device=dev())
    y = torch.randint(0, 5, (4,), device=dev())
    opt.zero_grad()
    cross_entropy(m(x)[-1], y).backward()
    g1 = m.conv2.conv_layer.weight.grad.clone()
    m.conv2.bn1.reset_running_stats()
    cross_entropy(m(x)[-1], y).backward()
    assert rel_l2(m.conv2.conv_layer.weight.grad.cpu().numpy(), (2 * g1).cpu().numpy()) < 1e-5
    opt.step()
    assert float(m.conv2.conv_layer.weight.grad.abs().sum()) > 0        # fuse_zero_grad=False: still readable
    opt.zero_grad()
    assert float(m.conv2.conv_layer.weight.grad.abs().sum()) == 0.0


def test_dta_adam_resumes_from_its_own_and_from_torch_adam_state_dict():
    """Optimizer resume (Lightning restores optimizer_states from its checkpoints): three uninterrupted steps == two steps,
    state_dict() -> a NEW DtaAdam on a reloaded model -> load_state_dict() -> one step; and the state_dict of a
    torch.optim.Adam that took the first two steps loads the same way."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    bands, classes, B, lr = 14, 6, 5, 1e-3
    p = O.init_params(O.hang2020_spec(bands, classes), seed=43)
    batches = [(torch.from_numpy(prng.uniform01(500 + s, 1, (B, bands, 11, 11))).to(dev()),
                torch.from_numpy(prng.randint(500 + s, 2, (B,), classes)).to(dev())) for s in range(3)]

    def steps(m, opt, which, ce):
        for s in which:
            x, y = batches[s]
            opt.zero_grad()
            ce(m(x), y).backward()
            opt.step()

    full = load(H.Hang2020(bands, classes), p).train()
    ofull = DtaAdam(full.parameters(), lr=lr)
    steps(full, ofull, range(3), cross_entropy)
    want = {k: v.detach().double().cpu().numpy() for k, v in full.state_dict().items()}

    # (a) DtaAdam -> DtaAdam
    a = load(H.Hang2020(bands, classes), p).train()
    oa = DtaAdam(a.parameters(), lr=lr)
    steps(a, oa, range(2), cross_entropy)
    msd, osd = copy.deepcopy(a.state_dict()), copy.deepcopy(oa.state_dict())
    oa.close()
    b = H.Hang2020(bands, classes).to(dev()).train()
    b.load_state_dict(msd)
    ob = DtaAdam(b.parameters(), lr=lr * 7)                # (the learning rate comes back from the state dict as well)
    ob.load_state_dict(osd)
    assert ob.param_groups[0]["lr"] == lr and ob.step_counts() == [2]
    w = b.spectral_network.conv1.conv_layer.weight
    assert ob.state[w]["exp_avg"].data_ptr() == ob.flat_m.data_ptr() + 4 * ob._offs[id(w)]      # still views of the flat moments
    steps(b, ob, [2], cross_entropy)
    for k, v in b.state_dict().items():
        assert rel_l2(v.double().cpu().numpy(), want[k]) < 1e-6, k
    assert ob.step_counts() == [3]

    # (b) torch.optim.Adam (first two steps, torch's cross-entropy) -> DtaAdam
    c = load(H.Hang2020(bands, classes), p).train()
    oc = torch.optim.Adam(c.parameters(), lr=lr)
    steps(c, oc, range(2), torch.nn.functional.cross_entropy)
    d = H.Hang2020(bands, classes).to(dev()).train()
    d.load_state_dict(copy.deepcopy(c.state_dict()))
    od = DtaAdam(d.parameters(), lr=lr)
    od.load_state_dict(copy.deepcopy(oc.state_dict()))
    assert od.step_counts() == [2]
    steps(d, od, [2], cross_entropy)
    for k, v in d.state_dict().items():
        if k.endswith("conv_layer.bias"):
            continue                                       # (no gradient signal in front of a BatchNorm: Adam noise only)
        assert rel_l2(v.double().cpu().numpy(), want[k]) < 2e-3, k


def test_fused_trainer_optimizer_state_interoperates_with_torch_adam():
    """FusedTrainer.optimizer_state_dict() / load_optimizer_state_dict() in torch.optim.Adam's layout (what a Lightning
    checkpoint's optimizer_states holds): two fused steps -> torch Adam takes the third, and two torch steps -> the fused
    trainer takes the third; both land on three uninterrupted fused steps."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    bands, classes, B, lr = 14, 8, 5, 1e-3
    p = O.init_params(O.hang2020_spec(bands, classes), seed=45)
    batches = [(torch.from_numpy(prng.uniform01(700 + s, 1, (B, bands, 11, 11))).to(dev()),
                torch.from_numpy(prng.randint(700 + s, 2, (B,), classes)).to(dev())) for s in range(3)]
    full = load(H.Hang2020(bands, classes), p).train()
    tf = FusedTrainer(full, lr=lr)
    for x, y in batches:
        tf.train_step(x, y)
    want = {k: v.detach().double().cpu().numpy() for k, v in full.state_dict().items()}

    def close(m, tol):
        for k, v in m.state_dict().items():
            if k.endswith("conv_layer.bias") or "classifier1" in k or "classifier2" in k:
                continue
            assert rel_l2(v.double().cpu().numpy(), want[k]) < tol, k

    # fused -> torch
    a = load(H.Hang2020(bands, classes), p).train()
    ta = FusedTrainer(a, lr=lr)
    for x, y in batches[:2]:
        ta.train_step(x, y)
    osd, msd = copy.deepcopy(ta.optimizer_state_dict()), copy.deepcopy(a.state_dict())
    b = H.Hang2020(bands, classes).to(dev()).train()
    b.load_state_dict(msd)
    ob = torch.optim.Adam(b.parameters(), lr=lr * 5)
    ob.load_state_dict(copy.deepcopy(osd))                # (torch keeps the 'step' tensors it is handed and steps them in place)
    assert ob.param_groups[0]["lr"] == lr
    x, y = batches[2]
    ob.zero_grad()
    torch.nn.functional.cross_entropy(b(x), y).backward()
    ob.step()
    close(b, 2e-3)
    # torch -> fused
    c = load(H.Hang2020(bands, classes), p).train()
    oc = torch.optim.Adam(c.parameters(), lr=lr)
    for x, y in batches[:2]:
        oc.zero_grad()
        torch.nn.functional.cross_entropy(c(x), y).backward()
        oc.step()
    d = H.Hang2020(bands, classes).to(dev()).train()
    d.load_state_dict(copy.deepcopy(c.state_dict()))
    td = FusedTrainer(d, lr=lr * 3)
    td.load_optimizer_state_dict(copy.deepcopy(oc.state_dict()))
    assert td.lr == lr and td.step_count == 2
    td.train_step(*batches[2])
    close(d, 2e-3)
    # fused -> fused is exact
    e = H.Hang2020(bands, classes).to(dev()).train()
    e.load_state_dict(msd)
    te = FusedTrainer(e, lr=lr)
    te.load_optimizer_state_dict(osd)
    te.train_step(*batches[2])
    for k, v in e.state_dict().items():
        assert torch.equal(v, full.state_dict()[k]), k


def test_ensemble_trainer_optimizer_state_round_trip():
    """EnsembleTrainer's Adam state in torch.optim.Adam's layout: per-year step counts survive (a year skipped once is one
    step behind), a fresh trainer resumes bit for bit, and torch.optim.Adam / optim.DtaAdam accept the same dict."""
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.optim import DtaAdam
    from deeptreeattention_amd.year import learned_ensemble
    years, bands, classes, B, lr = 3, 16, 8, 6, 1e-3
    p = O.init_params(O.learned_ensemble_spec(years, bands, classes), seed=83)
    cfg = {"pretrain_state_dict": None, "bands": bands}
    w = torch.ones(classes)

    def batch(step):
        imgs, y = _ensemble_step_inputs(step, years, B, bands, classes)
        return [torch.from_numpy(a).to(dev()) for a in imgs], torch.from_numpy(y).to(dev())

    full = load(learned_ensemble(years=years, classes=classes, config=cfg), p).train()
    tf = EnsembleTrainer(full, lr=lr, loss_weight=w)
    for step in range(4):
        xs, y = batch(step)
        tf.train_step(xs, y, [bool(float(x.abs().sum()) > 0) for x in xs])
    a = load(learned_ensemble(years=years, classes=classes, config=cfg), p).train()
    ta = EnsembleTrainer(a, lr=lr, loss_weight=w)
    for step in range(2):
        xs, y = batch(step)
        ta.train_step(xs, y, [bool(float(x.abs().sum()) > 0) for x in xs])
    assert ta.step_counts() == [2, 2, 1]
    osd, msd = copy.deepcopy(ta.optimizer_state_dict()), copy.deepcopy(a.state_dict())
    b = learned_ensemble(years=years, classes=classes, config=cfg).to(dev()).train()
    b.load_state_dict(msd)
    tb = EnsembleTrainer(b, lr=lr * 2, loss_weight=w)
    tb.load_optimizer_state_dict(copy.deepcopy(osd))
    assert tb.lr == lr and tb.step_counts() == [2, 2, 1]
    for step in (2, 3):
        xs, y = batch(step)
        tb.train_step(xs, y, [bool(float(x.abs().sum()) > 0) for x in xs])
    assert tb.step_counts() == tf.step_counts() == [3, 4, 3]
    for (k, v), (_, u) in zip(b.state_dict().items(), full.state_dict().items()):
        assert torch.equal(v, u), k
    # the same dict in the reference's optimizer and in the drop-in one
    c = learned_ensemble(years=years, classes=classes, config=cfg).to(dev()).train()
    c.load_state_dict(msd)
    torch.optim.Adam(c.parameters(), lr=lr).load_state_dict(copy.deepcopy(osd))
    oc = DtaAdam(c.parameters(), lr=lr)
    oc.load_state_dict(copy.deepcopy(osd))
    assert oc.step_counts() == [2, 2, 1]
    oc.close()


def test_metadata_trainer_optimizer_state_round_trip():
    """MetadataTrainer's Adam state over the whole fusion model (site MLP, HSI branch, fusion layer) in torch's layout:
    a fresh trainer resumes bit for bit (dropout off: no random op in the step), and torch.optim.Adam accepts the dict."""
    from deeptreeattention_amd.engine import MetadataTrainer
    from deeptreeattention_amd.metadata import metadata_sensor_fusion
    bands, classes, sites, B, lr = 12, 8, 5, 16, 1e-3
    torch.manual_seed(9)
    base = metadata_sensor_fusion(bands=bands, sites=sites, classes=classes).to(dev()).train()
    base.metadata_model.dropout.p = 0.0
    g = torch.Generator(device=dev())
    g.manual_seed(2)
    batches = [(torch.rand(B, bands, 11, 11, device=dev(), generator=g), torch.randint(0, sites, (B,), device=dev(), generator=g),
                torch.randint(0, classes, (B,), device=dev(), generator=g)) for _ in range(3)]
    full = copy.deepcopy(base)
    tf = MetadataTrainer(full, lr=lr)
    for x, s_, y in batches:
        tf.train_step(x, s_, y)
    a = copy.deepcopy(base)
    ta = MetadataTrainer(a, lr=lr)
    for x, s_, y in batches[:2]:
        ta.train_step(x, s_, y)
    osd, msd = copy.deepcopy(ta.optimizer_state_dict()), copy.deepcopy(a.state_dict())
    b = copy.deepcopy(base)
    b.load_state_dict(msd)
    tb = MetadataTrainer(b, lr=lr * 4)
    tb.load_optimizer_state_dict(copy.deepcopy(osd))
    assert tb.lr == lr and tb.sensor.step_count == 2
    tb.train_step(*batches[2])
    for (k, v), (_, u) in zip(b.state_dict().items(), full.state_dict().items()):
        assert torch.equal(v, u), k
    c = copy.deepcopy(base)
    torch.optim.Adam(c.parameters(), lr=lr).load_state_dict(copy.deepcopy(osd))


def test_torch_adam_resumes_from_a_dta_adam_state_dict():
    """The other direction: a DtaAdam state dict loads into torch.optim.Adam (same group keys, same state layout) and
    the third step there lands on the uninterrupted DtaAdam run."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    bands, classes, B, lr = 14, 8, 5, 1e-3
    p = O.init_params(O.hang2020_spec(bands, classes), seed=44)
    batches = [(torch.from_numpy(prng.uniform01(600 + s, 1, (B, bands, 11, 11))).to(dev()),
                torch.from_numpy(prng.randint(600 + s, 2, (B,), classes)).to(dev())) for s in range(3)]

    def steps(m, opt, which, ce):
        for s in which:
            x, y = batches[s]
            opt.zero_grad()
            ce(m(x), y).backward()
            opt.step()

    full = load(H.Hang2020(bands, classes), p).train()
    ofull = DtaAdam(full.parameters(), lr=lr)
    steps(full, ofull, range(3), cross_entropy)
    a = load(H.Hang2020(bands, classes), p).train()
    oa = DtaAdam(a.parameters(), lr=lr)
    steps(a, oa, range(2), cross_entropy)
    msd, osd = copy.deepcopy(a.state_dict()), copy.deepcopy(oa.state_dict())
    oa.close()
    b = H.Hang2020(bands, classes).to(dev()).train()
    b.load_state_dict(msd)
    ob = torch.optim.Adam(b.parameters(), lr=lr * 3)
    ob.load_state_dict(osd)
    assert ob.param_groups[0]["lr"] == lr
    steps(b, ob, [2], torch.nn.functional.cross_entropy)
    for (k, v), (_, u) in zip(b.state_dict().items(), full.state_dict().items()):
        if k.endswith("conv_layer.bias") or "classifier1" in k or "classifier2" in k:
            continue      # no gradient signal (BatchNorm behind the bias) / torch leaves the unused heads' grad None
        assert rel_l2(v.double().cpu().numpy(), u.double().cpu().numpy()) < 2e-3, k


def test_dta_adam_resume_keeps_per_year_step_counts():
    from deeptreeattention_amd.year import learned_ensemble
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    years, bands, classes, B, lr = 3, 16, 7, 6, 1e-3
    p = O.init_params(O.learned_ensemble_spec(years, bands, classes), seed=81)
    cfg = {"pretrain_state_dict": None, "bands": bands}

    def run(m, opt, which):
        for step in which:
            imgs, y = _ensemble_step_inputs(step, years, B, bands, classes)
            opt.zero_grad()
            cross_entropy(m([torch.from_numpy(a).to(dev()) for a in imgs]), torch.from_numpy(y).to(dev())).backward()
            opt.step()

    full = load(learned_ensemble(years=years, classes=classes, config=cfg), p).train()
    ofull = DtaAdam(full.parameters(), lr=lr)
    run(full, ofull, range(4))
    a = load(learned_ensemble(years=years, classes=classes, config=cfg), p).train()
    oa = DtaAdam(a.parameters(), lr=lr)
    run(a, oa, range(2))                                   # step 1 skips year 2
    assert oa.step_counts() == [2, 2, 1]
    msd, osd = copy.deepcopy(a.state_dict()), copy.deepcopy(oa.state_dict())
    oa.close()
    b = learned_ensemble(years=years, classes=classes, config=cfg).to(dev()).train()
    b.load_state_dict(msd)
    ob = DtaAdam(b.parameters(), lr=lr)
    ob.load_state_dict(osd)
    assert ob.step_counts() == [2, 2, 1]
    run(b, ob, [2, 3])
    assert ob.step_counts() == ofull.step_counts() == [3, 4, 3]
    for (k, v), (_, u) in zip(b.state_dict().items(), full.state_dict().items()):
        assert rel_l2(v.double().cpu().numpy(), u.double().cpu().numpy()) < 1e-6, k


@pytest.mark.parametrize("classes", [7, 199])
def test_class_counts_that_are_not_a_multiple_of_four_are_reproducible(classes):
    """Score rows of 7 or 199 floats are not 16-byte loadable: the two head GEMMs that read the score gradient then take
    the element-wise wave-tile form (no split-K atomics), so reruns give the same bits, as for aligned class counts."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer

    def run():
        torch.manual_seed(11)
        m = H.Hang2020(24, classes, precision="bf16").to(dev()).train()
        tr = FusedTrainer(m, lr=1e-3, loss_weight=torch.linspace(0.2, 1.0, classes).to(dev()))
        g = torch.Generator(device=dev())
        g.manual_seed(3)
        x = torch.rand(530, 24, 11, 11, device=dev(), generator=g)
        y = torch.randint(0, classes, (530,), device=dev(), generator=g)
        for _ in range(3):
            tr.train_step(x, y)
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    a, b, c = run(), run(), run()
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k]), k


@pytest.mark.parametrize("classes,sites,B,p_drop", [(200, 23, 64, 0.0), (200, 23, 64, 0.7), (5, 4, 9, 0.0), (7, 300, 33, 0.5)])
def test_native_metadata_head_vs_torch_autograd(classes, sites, B, p_drop):
    """csrc/meta.hip (dta_meta_head_forward / backward + optim's cross-entropy launch) against the reference's modules run
    by torch autograd (metadata.py:9-44: Embedding -> BatchNorm1d -> Dropout -> Linear -> ReLU, cat, Linear, ReLU, unweighted
    CE): fused scores, loss, d(loss)/d(hsi scores), every parameter gradient and the running statistics -- same Philox
    draw for the dropout mask (torch's generator, same shape, same position in the stream)."""
    from deeptreeattention_amd.engine import MetadataTrainer
    from deeptreeattention_amd.metadata import metadata_sensor_fusion
    torch.manual_seed(21)
    bands = 12
    a = metadata_sensor_fusion(bands=bands, sites=sites, classes=classes).to(dev()).train()
    a.metadata_model.dropout.p = p_drop
    with torch.no_grad():
        a.metadata_model.batch_norm.weight.uniform_(0.5, 1.5)
        a.metadata_model.batch_norm.bias.uniform_(-0.3, 0.3)
    b = copy.deepcopy(a)
    tr = MetadataTrainer(a, lr=1e-3)
    assert tr.native_head
    g = torch.Generator(device=dev())
    g.manual_seed(5)
    scores = torch.randn(B, classes, device=dev(), generator=g)
    site = torch.randint(0, sites, (B,), device=dev(), generator=g)
    y = torch.randint(0, classes, (B,), device=dev(), generator=g)
    # torch reference (module b), same RNG position for the dropout draw
    torch.manual_seed(77)
    leaf = scores.clone().requires_grad_(True)
    out_ref = torch.relu(b.fc1(torch.cat([b.metadata_model(site), leaf], dim=1)))
    loss_ref = torch.nn.functional.cross_entropy(out_ref, y)
    loss_ref.backward()
    torch.manual_seed(77)
    tr.sensor._zero_grads()
    tr._attach_grads()
    dscores, loss = tr._native_step(scores, site, y)
    out = tr._mh[2]
    tol = 2e-5
    assert rel_l2(out.cpu().numpy(), out_ref.detach().cpu().numpy()) < tol
    assert abs(float(loss) - float(loss_ref)) < tol * abs(float(loss_ref))
    assert rel_l2(dscores.cpu().numpy(), leaf.grad.cpu().numpy()) < 1e-4
    for (k, pa), (_, pb) in zip(list(a.metadata_model.named_parameters()) + list(a.fc1.named_parameters()),
                                list(b.metadata_model.named_parameters()) + list(b.fc1.named_parameters())):
        assert rel_l2(pa.grad.cpu().numpy(), pb.grad.cpu().numpy()) < 2e-4, k
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        va, vb = getattr(a.metadata_model.batch_norm, k), getattr(b.metadata_model.batch_norm, k)
        assert rel_l2(va.double().cpu().numpy(), vb.double().cpu().numpy()) < 1e-5, k
    # eval mode: running statistics, no dropout
    a.eval(); b.eval()
    with torch.no_grad():
        o2, _ = tr._native_forward(scores, site, False)
        r2 = torch.relu(b.fc1(torch.cat([b.metadata_model(site), scores], dim=1)))
    assert rel_l2(o2.cpu().numpy(), r2.cpu().numpy()) < tol
    tr.close()


def _ensemble_step_inputs(step, years, B, bands, classes):
    imgs = [prng.uniform01(82 + step, yy, (B, bands, 11, 11)) for yy in range(years)]
    if step == 1:
        imgs[2] = np.zeros_like(imgs[2])
    if step == 2:
        imgs[0] = np.zeros_like(imgs[0])
    return imgs, prng.randint(82 + step, 7, (B,), classes)


def test_module_path_ensemble_steps_vs_reference_golden(golden):
    """The reference's MultiStage step on one level (multi_stage.py:277-288: forward, F.cross_entropy, backward, Adam)
    through the plugin modules with DtaAdam: four steps, two with an all-zero year, decided on the device (no host
    round trip in learned_ensemble.forward), against the reference's own learned_ensemble + torch Adam golden."""
    from deeptreeattention_amd.year import learned_ensemble
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    g = golden("ensemble_steps.npz")
    years, bands, classes, B, lr = 3, 16, 7, 6, 1e-3
    p = O.init_params(O.learned_ensemble_spec(years, bands, classes), seed=81)
    m = load(learned_ensemble(years=years, classes=classes, config={"pretrain_state_dict": None, "bands": bands}), p)
    m.train()
    w = torch.from_numpy((0.1 + (np.arange(classes) % 7)).astype(np.float32)).to(dev())
    opt = DtaAdam(m.parameters(), lr=lr)
    for step in range(4):
        imgs, y = _ensemble_step_inputs(step, years, B, bands, classes)
        xs = [torch.from_numpy(a).to(dev()) for a in imgs]
        opt.zero_grad()
        scores = m(xs)
        loss = cross_entropy(scores, torch.from_numpy(y).to(dev()), weight=w)
        loss.backward()
        opt.step()
        assert rel_l2(scores.detach().cpu().numpy(), g[f"step{step}/score"]) < TIGHT, step
        ref = float(g[f"step{step}/loss"])
        assert abs(float(loss) - ref) < TIGHT * abs(ref), step
        for k, prm in m.named_parameters():
            if k.endswith("conv_layer.bias"):
                continue
            a = prm.detach().cpu().numpy()
            ref = float(g[f"step{step}/pnorm/{k}"])
            assert abs(np.sqrt((a.astype(np.float64) ** 2).sum()) - ref) <= 1e-3 * ref, (step, k)
            if f"step{step}/pfull/{k}" in g:
                assert rel_l2(a, g[f"step{step}/pfull/{k}"]) < 2e-3, (step, k)
        for k, b in m.named_buffers():
            tol = 2e-3 if k.endswith("running_mean") else TIGHT
            assert rel_l2(b.cpu().numpy(), g[f"step{step}/buf/{k}"]) < tol, (step, k)
    assert opt.step_counts() == [3, 4, 3]                 # skipped years: passed over, as torch's Adam passes over grad None
    m.eval()
    with torch.no_grad():
        s = m(xs)                                         # validation forward: gated on the device too, flags not republished
    assert torch.isfinite(s).all()


def test_ensemble_without_dta_adam_keeps_grad_none_for_skipped_years():
    from deeptreeattention_amd.year import learned_ensemble
    torch.manual_seed(1)
    m = learned_ensemble(3, 5, {"pretrain_state_dict": None, "bands": 12}).to(dev()).train()
    xs = [torch.rand(4, 12, 11, 11, device=dev()) for _ in range(3)]
    xs[1].zero_()
    torch.nn.functional.cross_entropy(m(xs), torch.randint(0, 5, (4,), device=dev())).backward()
    assert m.year_models[1].conv1.conv_layer.weight.grad is None
    assert m.year_models[0].conv1.conv_layer.weight.grad is not None


# ---- data-parallel year ensemble: a year missing on ONE rank only -------------------------------------
BANDS, CLASSES, B = 20, 7, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _ens_worker(rank, world, port, mode, exchange, out, BANDS=BANDS, B=B, prec=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.year import learned_ensemble
    d = torch.device("cuda:0")
    torch.cuda.set_device(0)
    torch.manual_seed(5)
    m = learned_ensemble(3, CLASSES, {"pretrain_state_dict": None, "bands": BANDS}).to(d).train()
    if prec:
        for net in m.year_models:
            net.precision = prec
    tr = EnsembleTrainer(m, lr=1e-3, exchange=exchange, keep_grads=True,
                         exchange_opts={"max_workgroups": 32, "timeout_s": 20.0} if exchange == "peer" else None)
    losses = []
    for step in range(2):
        imgs = [torch.from_numpy(prng.uniform01(200 + 10 * step + rank, yy, (B, BANDS, 11, 11))).to(d) for yy in range(3)]
        if rank == 0:
            imgs[1].zero_()                           # year 1 is missing on rank 0 ONLY
        if step == 1 and rank == 1:
            imgs[0].zero_(); imgs[1].zero_(); imgs[2].zero_()          # rank 1 has NO year at all in step 2
        y = torch.from_numpy(prng.randint(200 + rank, 2, (B,), CLASSES)).to(d)
        present = None if mode == "device" else [bool(x.any()) for x in imgs]
        if mode == "present" and not any(present):
            present = None                            # (a rank without any year cannot use the host-flag form)
        losses.append(float(tr.train_step(imgs, y, present)))
    torch.cuda.synchronize()
    tr.check_exchange()
    g1 = tr.grad_of(m.year_models[1].classifier3.fc1.bias).detach().cpu().numpy().copy()
    out[(mode, rank)] = ({k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}, losses, tr.step_counts(), g1)
    tr.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,geom", [("torch", "small"), ("peer", "small"), ("peer", "bench")])
def test_dp_ensemble_year_missing_on_one_rank_contributes_zeros(exchange, geom):
    """A rank whose batch lacks a year that the other rank kept must send ZEROS for that year (the reference's skipped year
    has grad None): the device-decided step (all years launched, gradients gated by the rank's own flags) must equal the
    `present=[...]` step (only the kept years launched), which the reference's golden pins.  Second step: rank 1 has no
    year at all -- NaN loss there, nothing contributed, rank 0's years still stepped on both ranks.
    geom = "bench" (ADVICE r5): 369 bands, batch 512, bf16 -- the geometry at which the overlapped peer exchange has a
    combined weight-gradient + head-reduce kernel for SOME year counts: with present= flags rank 0 launches two years and
    rank 1 three, so a per-rank choice of that kernel would leave one rank waiting for head flags the other never posts."""
    world = 2
    extra = (369, 512, "bf16") if geom == "bench" else (BANDS, B, None)
    tol = 1e-4 if geom == "bench" else 2e-5
    mgr = mp.Manager()
    out = mgr.dict()
    for mode in ("device", "present"):
        for attempt in range(2):
            try:
                mp.spawn(_ens_worker, args=(world, _free_port(), mode, exchange, out) + extra, nprocs=world, join=True)
                break
            except Exception:
                if attempt == 1:
                    raise
    for rank in range(world):
        sd_d, l_d, steps_d, g_d = out[("device", rank)]
        sd_p, l_p, steps_p, g_p = out[("present", rank)]
        assert steps_d == steps_p == [2, 1, 2], (steps_d, steps_p)
        assert np.isfinite(l_d[0]) and abs(l_d[0] - l_p[0]) < 1e-5 * abs(l_p[0])
        if rank == 1:
            assert np.isnan(l_d[1])                   # no year on this rank: an empty mean (the reference raises)
        for k in sd_d:
            if "num_batches_tracked" in k:
                assert int(sd_d[k]) == int(sd_p[k]), (rank, k)
            elif k.endswith("conv_layer.bias"):
                continue
            else:
                assert rel_l2(sd_d[k], sd_p[k]) < tol, (rank, k)
    # replicas identical; year 1's summed gradient after step 2 is zero (nobody kept it)
    for k in out[("device", 0)][0]:
        if "running_" in k or "num_batches_tracked" in k:
            continue
        assert rel_l2(out[("device", 0)][0][k], out[("device", 1)][0][k]) < 1e-6, k
    assert np.all(out[("device", 0)][3] == 0.0)


# ---- peer-exchange probe (the probe always failed on a ctypes argument-count error) -----------------
def test_peer_probe_runs_single_rank():
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([sys.executable, os.path.join(REPO, "deeptreeattention_amd", "peer_probe.py"), d, "0", "1", "0"],
                           capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]


def test_peer_probe_two_ranks_sharing_the_gpu():
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "deeptreeattention_amd", "peer_probe.py"), d, str(r), "2", "0"],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
        outs = [p.communicate(timeout=180) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]


def _probe_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    torch.cuda.set_device(0)
    from deeptreeattention_amd.dist import choose_exchange, probe_peer_exchange
    ok, why = probe_peer_exchange(None)
    out[rank] = (ok, why, choose_exchange(None))
    dist.destroy_process_group()


def test_choose_exchange_selects_peer_when_the_probe_passes():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_probe_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in range(2):
        ok, why, choice = out[rank]
        assert ok, why
        assert choice == "peer"


# ---- DtaAdam data-parallel: the module-level step under DDP semantics, without a DDP wrapper ----------------------------
def _dta_adam_dp_worker(rank, world, port, kind, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    from deeptreeattention_amd.optim import DtaAdam, cross_entropy
    d = torch.device("cuda:0")
    torch.cuda.set_device(0)
    p = O.init_params(O.hang2020_spec(BANDS, CLASSES), seed=31)
    m = H.Hang2020(BANDS, CLASSES)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
    m = m.to(d).train()
    w = torch.from_numpy((0.1 + (np.arange(CLASSES) % 7)).astype(np.float32)).to(d)
    opts = {"max_workgroups": 32, "timeout_s": 20.0}
    if kind == "fused":
        tr = FusedTrainer(m, lr=1e-3, loss_weight=w, exchange="torch")
    else:
        opt = DtaAdam(m.parameters(), lr=1e-3, exchange=kind, exchange_opts=opts if kind == "peer" else None)
    for step in range(2):
        x = torch.from_numpy(prng.uniform01(700 + 10 * step + rank, 1, (B, BANDS, 11, 11))).to(d)
        y = torch.from_numpy(prng.randint(700 + rank, 2, (B,), CLASSES)).to(d)
        if kind == "fused":
            tr.train_step(x, y)
        else:
            opt.zero_grad()
            cross_entropy(m(x), y, weight=w).backward()
            opt.step()
    torch.cuda.synchronize()
    out[(kind, rank)] = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    if kind == "fused":
        tr.close()
    else:
        opt.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["torch", "peer"])
def test_dta_adam_data_parallel_equals_the_fused_trainer(kind):
    """DtaAdam(process_group) sums the flat gradient over the ranks inside step() (DDP's mean, per-rank BatchNorm): two
    ranks, two steps, against the FusedTrainer's data-parallel step (pinned to the oracle in tests/test_ddp_gpu.py)."""
    mgr = mp.Manager()
    out = mgr.dict()
    for k in ("fused", kind):
        mp.spawn(_dta_adam_dp_worker, args=(2, _free_port(), k, out), nprocs=2, join=True)
    for rank in range(2):
        a, b = out[("fused", rank)], out[(kind, rank)]
        for k in a:
            if k.endswith("conv_layer.bias"):
                continue
            assert rel_l2(b[k], a[k]) < 2e-5, (rank, k)
    for k in out[(kind, 0)]:
        if "running_" in k or "num_batches_tracked" in k:
            continue
        assert rel_l2(out[(kind, 0)][k], out[(kind, 1)][k]) < 1e-6, k


def test_native_metadata_head_poisons_an_out_of_range_site():
    """A site index outside [0, sites) (torch's Embedding device-asserts there): the loss becomes NaN instead of training on
    garbage, and nothing is read out of bounds."""
    from deeptreeattention_amd.engine import MetadataTrainer
    from deeptreeattention_amd.metadata import metadata_sensor_fusion
    torch.manual_seed(3)
    m = metadata_sensor_fusion(bands=12, sites=4, classes=8).to(dev()).train()
    tr = MetadataTrainer(m, lr=1e-3)
    x = torch.rand(6, 12, 11, 11, device=dev())
    y = torch.randint(0, 8, (6,), device=dev())
    site = torch.tensor([0, 1, 2, 3, 9, -1], device=dev())
    assert not np.isfinite(float(tr.train_step(x, site, y)))
    tr.close()
