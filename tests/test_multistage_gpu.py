"""The reference's multi-stage step -- every level of the hierarchy on every batch (train.py:75-100, multi_stage.py:41-66,
:258-288) -- as ONE launch chain over levels x years (dta_multistage_*), against the reference's own golden steps and against
the level-by-level trainers; year ensembles of more than four years through the fused trainer."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import hang2020_np as O
from oracle import prng
from oracle.recipes import MULTISTAGE, multistage_inputs, multistage_weight

pytestmark = pytest.mark.gpu
TIGHT = 2e-4


def dev():
    return torch.device("cuda:0")


def _load(mod, params):
    mod.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in params.items()})
    return mod.to(dev())


def _levels(prec="fp32"):
    from deeptreeattention_amd.year import learned_ensemble
    c = MULTISTAGE
    models, ws = [], []
    for l, classes in enumerate(c["classes"]):
        m = _load(learned_ensemble(years=c["years"], classes=classes, config={"pretrain_state_dict": None, "bands": c["bands"]}),
                  O.init_params(O.learned_ensemble_spec(c["years"], c["bands"], classes), seed=301 + l))
        for net in m.year_models:
            net.precision = prec
        m.train()
        models.append(m)
        ws.append(torch.from_numpy(multistage_weight(classes)))
    return models, ws


def _batch(step):
    c = MULTISTAGE
    batch, present = [], []
    for l, classes in enumerate(c["classes"]):
        imgs, y = multistage_inputs(step, l, c["years"], c["B"], c["bands"], classes)
        present.append([bool(a.any()) for a in imgs])
        batch.append((["id"] * c["B"], {"HSI": [torch.from_numpy(a).to(dev()) for a in imgs]}, torch.from_numpy(y).to(dev())))
    return batch, present


def _check_level(g, tag, tr, m, loss, lr=1e-3):
    assert rel_l2(tr.scores.cpu().numpy(), g[f"{tag}/score"]) < TIGHT, tag
    ref = float(g[f"{tag}/loss"])
    assert abs(float(loss) - ref) < TIGHT * abs(ref), tag
    for k, prm in m.named_parameters():
        if k.endswith("conv_layer.bias"):
            continue   # conv biases under BN: gradient is rounding noise, Adam turns its sign into +-lr steps
        a = prm.detach().cpu().numpy()
        ref = float(g[f"{tag}/pnorm/{k}"])
        assert abs(np.sqrt((a.astype(np.float64) ** 2).sum()) - ref) <= 1e-3 * ref, (tag, k)
        if f"{tag}/pfull/{k}" in g:
            assert rel_l2(a, g[f"{tag}/pfull/{k}"]) < 2e-3, (tag, k)
        else:
            idx = (prng.hash_u64(7, 99, 256) % np.uint64(a.size)).astype(np.int64)
            assert rel_l2(a.reshape(-1)[idx], g[f"{tag}/psamp/{k}"]) < 2e-3, (tag, k)
    for k, b in m.named_buffers():
        # running means carry the conv bias, whose gradient under BatchNorm is rounding noise that Adam turns into +-lr steps
        # (tests/test_hip_modules.py holds 2e-3 at lr 1e-3): the bound scales with the level's learning rate
        tol = 2e-3 * (lr / 1e-3) if k.endswith("running_mean") else TIGHT
        assert rel_l2(b.cpu().numpy(), g[f"{tag}/buf/{k}"]) < tol, (tag, k)


@pytest.mark.parametrize("use_present", [False, True])
def test_multistage_batched_steps_vs_reference_golden(golden, use_present):
    """Two levels (3 and 5 classes, learning rates 1e-3 and 2e-3) x three years, three steps, two of them with an all-zero
    year in one level: scores, loss, every parameter and every BatchNorm buffer of BOTH levels after each step against the
    reference's per-level learned_ensemble + F.cross_entropy + torch Adam -- all networks of a step in one launch chain."""
    from deeptreeattention_amd.engine import MultiStageTrainer
    g = golden("multistage_steps.npz")
    models, ws = _levels()
    driver = MultiStageTrainer(models, list(MULTISTAGE["lrs"]), ws)
    for step in range(MULTISTAGE["steps"]):
        batch, present = _batch(step)
        losses = driver.training_step_all(batch, step, present if use_present else None)
        assert driver.batched_last
        for l, (tr, m) in enumerate(zip(driver.levels, models)):
            _check_level(g, f"step{step}/level{l}", tr, m, losses[l], MULTISTAGE["lrs"][l])
    # the skipped years' optimizer step counts did not advance (torch's Adam passes over grad-None parameters)
    assert driver.levels[0].step_counts() == [2, 3, 3]
    assert driver.levels[1].step_counts() == [3, 3, 2]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_multistage_batched_equals_level_by_level(prec):
    """The one-chain step against the same levels stepped one after the other (training_step per optimizer_idx, as
    Lightning does): the same kernels on the same data in another grouping -- parameters after three steps agree to float32
    rounding of the reductions whose split depends on the group count."""
    from deeptreeattention_amd.engine import MultiStageTrainer
    ma, wa = _levels(prec)
    mb, wb = _levels(prec)
    a = MultiStageTrainer(ma, list(MULTISTAGE["lrs"]), wa)
    b = MultiStageTrainer(mb, list(MULTISTAGE["lrs"]), wb)
    for step in range(MULTISTAGE["steps"]):
        batch, present = _batch(step)
        la = a.training_step_all(batch, step, present)
        lb = [b.training_step(batch, step, l, present[l]) for l in range(len(mb))]
        assert a.batched_last
        for x, y in zip(la, lb):
            assert abs(float(x) - float(y)) <= 1e-5 * abs(float(y))
    tol = 1e-5 if prec == "fp32" else 2e-3
    for m1, m2 in zip(ma, mb):
        for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            if k.endswith("conv_layer.bias"):
                continue
            assert rel_l2(p1.detach().cpu().numpy(), p2.detach().cpu().numpy()) < tol, k
        for (k, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
            assert rel_l2(b1.cpu().numpy(), b2.cpu().numpy()) < max(tol, 1e-5), k


def test_multistage_five_levels_full_width():
    """The reference's shape: 5 levels x 3 years of spectral_network(369, classes_l) on 11x11 crops, batch 128
    (config.yml:58), bf16: one chain of 15 networks; the losses equal the level-by-level step's."""
    from deeptreeattention_amd.engine import MultiStageTrainer
    from deeptreeattention_amd.year import learned_ensemble
    classes = [2, 2, 12, 7, 5]
    cfg = {"pretrain_state_dict": None, "bands": 369}

    def build():
        torch.manual_seed(5)
        ms = []
        for c in classes:
            m = learned_ensemble(3, c, cfg).to(dev()).train()
            for net in m.year_models:
                net.precision = "bf16"
            ms.append(m)
        return MultiStageTrainer(ms, [1e-4] * 5)
    a, b = build(), build()
    torch.manual_seed(6)
    batch = [(None, {"HSI": [torch.rand(128, 369, 11, 11, device=dev()) for _ in range(3)]}, torch.randint(0, c, (128,), device=dev()))
             for c in classes]
    for step in range(2):
        la = a.training_step_all(batch, step)
        assert a.batched_last
        lb = [b.training_step(batch, step, l) for l in range(5)]
        for x, y in zip(la, lb):
            assert torch.isfinite(x) and abs(float(x) - float(y)) <= 2e-3 * abs(float(y)), (step, float(x), float(y))


def test_multistage_falls_back_when_levels_differ():
    """Levels whose batches differ in size cannot share a chain: stepped one after the other, same results as ever."""
    from deeptreeattention_amd.engine import MultiStageTrainer
    models, ws = _levels()
    driver = MultiStageTrainer(models, list(MULTISTAGE["lrs"]), ws)
    batch, present = _batch(0)
    ind, inp, y = batch[1]
    batch[1] = (ind[:4], {"HSI": [x[:4].contiguous() for x in inp["HSI"]]}, y[:4])
    losses = driver.training_step_all(batch, 0, present)
    assert not driver.batched_last and len(losses) == 2 and all(torch.isfinite(v) for v in losses)


@pytest.mark.parametrize("years", [5, 7, 18])
def test_fused_ensemble_trainer_more_than_four_years(years):
    """A year ensemble of more than four years through the fused trainer (the reference takes the year count from the data,
    multi_stage.py:39, :61-66): against the module path (autograd.Function per network + torch Adam), two steps, one year
    all-zero.  18 years: more than one grouped launch takes (DTA_MAX_YEARS = 16) -- the fused trainer chunks."""
    from deeptreeattention_amd.engine import EnsembleTrainer
    from deeptreeattention_amd.year import learned_ensemble
    bands, classes, B = 16, 6, 10
    cfg = {"pretrain_state_dict": None, "bands": bands}
    torch.manual_seed(11)
    m1 = learned_ensemble(years, classes, cfg).to(dev()).train()
    m2 = learned_ensemble(years, classes, cfg).to(dev()).train()
    m2.load_state_dict(m1.state_dict())
    for m in (m1, m2):
        for net in m.year_models:
            net.precision = "fp32"
    tr = EnsembleTrainer(m1, lr=1e-3)
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3)
    for step in range(2):
        imgs = [torch.rand(B, bands, 11, 11, device=dev()) for _ in range(years)]
        imgs[step + 1].zero_()
        y = torch.randint(0, classes, (B,), device=dev())
        loss = tr.train_step(imgs, y, present=None if step == 0 else [bool(x.any()) for x in imgs])
        opt.zero_grad(set_to_none=True)
        ref = torch.nn.functional.cross_entropy(m2(imgs), y)
        ref.backward()
        opt.step()
        assert abs(float(loss) - float(ref.detach())) <= 1e-4 * abs(float(ref.detach())), step
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        if k.endswith("conv_layer.bias"):
            continue
        assert rel_l2(p1.detach().cpu().numpy(), p2.detach().cpu().numpy()) < 2e-3, k


def test_fit_multistage_epochs_reduce_every_levels_loss():
    """loop.fit_multistage: the reference's train.py loop over a MultiStage module (train.py:75-100) -- per-level loaders of
    different lengths zipped per batch (the shorter one cycles), every level of a batch in one launch chain, per-level
    validation and plateau schedulers.  On a small separable problem every level's training loss falls."""
    from deeptreeattention_amd.engine import MultiStageTrainer
    from deeptreeattention_amd.loop import PlateauScheduler, SyntheticTreeDataset, fit_multistage
    from deeptreeattention_amd.year import learned_ensemble
    classes, bands, years = [2, 3], 12, 3
    torch.manual_seed(7)
    models = [learned_ensemble(years, c, {"pretrain_state_dict": None, "bands": bands}).to(dev()).train() for c in classes]
    for m in models:
        for net in m.year_models:
            net.precision = "fp32"
    tr = MultiStageTrainer(models, [2e-3, 2e-3])
    train = [SyntheticTreeDataset(64, bands, classes[0], years=years, missing=0.2, seed=1, device=dev()),
             SyntheticTreeDataset(48, bands, classes[1], years=years, seed=2, device=dev())]      # 4 and 3 batches of 16
    val = [SyntheticTreeDataset(16, bands, c, years=years, seed=3 + i, device=dev()) for i, c in enumerate(classes)]
    sch = [PlateauScheduler(t, patience=0) for t in tr.levels]
    hist = fit_multistage(tr, train, val, epochs=6, batch_size=16, schedulers=sch, shuffle=False)
    assert tr.batched_last and len(hist) == 6
    for l in range(2):
        assert hist[-1]["train_loss"][l] < hist[0]["train_loss"][l], (l, hist[0], hist[-1])
        assert np.isfinite(hist[-1]["val_loss"][l])
    assert all(h["lr"][0] <= 2e-3 for h in hist)
    for m in models:
        assert m.training


@pytest.mark.parametrize("use_present", [False, True])
def test_multistage_predict_step_one_chain_equals_per_level_predictors(use_present):
    """MultiStage.predict_step (multi_stage.py:306-318: every level's model on the SAME crops): the one-chain eval forward
    over levels x years (MultiStagePredictor) against a Predictor per level and against the modules' own eval forward; one
    year of the batch is all-zero."""
    from deeptreeattention_amd.engine import MultiStagePredictor, MultiStageTrainer, Predictor
    models, ws = _levels()
    driver = MultiStageTrainer(models, list(MULTISTAGE["lrs"]), ws)
    batch, present = _batch(1)                      # step 1: level 1's inputs have year 2 zeroed
    for step in range(2):                           # move the BatchNorm running statistics off their initial values
        b, pr = _batch(step)
        driver.training_step_all(b, step, pr)
    individual, inputs, _ = batch[1]
    pres = present[1] if use_present else None
    ids, yhats = driver.predict_step((individual, inputs), 0, pres)
    assert ids == individual and len(yhats) == 2 and driver._ms_predictor._key is not None
    for l, m in enumerate(models):
        want = Predictor(m)(inputs["HSI"], True, pres)[0]
        assert torch.equal(yhats[l], want), l       # same kernels, same order of the mean: identical bits
        m.eval()
        with torch.no_grad():
            ref = torch.softmax(m(inputs["HSI"]), dim=1)
        m.train()
        assert rel_l2(yhats[l].cpu().numpy(), ref.cpu().numpy()) < 1e-5, l
        assert abs(float(yhats[l].sum()) - MULTISTAGE["B"]) < 1e-4
    top = MultiStagePredictor(models)(inputs["HSI"], True, pres)
    for l in range(2):
        tv, ti = torch.topk(yhats[l], 2, dim=1)
        assert torch.equal(top[l][1].cpu(), ti.cpu())


def test_multistage_predict_wide_levels_take_the_re_formed_tail():
    """dta_multistage_predict's one-launch epilogue keeps 256 classes of a row in registers and re-forms the rest: a level
    wider than that (300 classes) beside a 2-class level, scores/top-2 identical to the per-level Predictor."""
    from deeptreeattention_amd.engine import MultiStagePredictor, Predictor
    from deeptreeattention_amd.year import learned_ensemble
    torch.manual_seed(5)
    dev = torch.device("cuda:0")
    cfg = {"pretrain_state_dict": None, "bands": 12}
    models = [learned_ensemble(2, c, cfg).to(dev).eval() for c in (300, 2)]
    xs = [torch.rand(9, 12, 11, 11, device=dev) for _ in range(2)]
    xs[1][:] = 0                                     # one empty year: the gate drops it from every level's mean
    got = MultiStagePredictor(models)(xs, True, None)
    for l, m in enumerate(models):
        probs, ti, ts = Predictor(m)(xs, True, None)
        assert torch.equal(got[l][0], probs), l
        assert torch.equal(got[l][1], ti) and torch.equal(got[l][2], ts), l
        assert int(got[l][1].max()) < (300, 2)[l]


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_frozen_predictors_reuse_the_weight_re_layouts(prec):
    """DTA_REUSE_PACKED (tile prediction with a trained model, reference predict.py:140-151): a frozen predictor packs the
    conv / attention weights once per workspace; every later batch must give the bits of a predictor that re-packs each call.
    BatchNorm statistics stay live; an in-place WEIGHT update needs refresh()."""
    import deeptreeattention_amd
    from deeptreeattention_amd.engine import MultiStagePredictor, Predictor
    from deeptreeattention_amd.Hang2020 import Hang2020, get_default_precision
    from deeptreeattention_amd.year import learned_ensemble
    default = get_default_precision()
    deeptreeattention_amd.set_default_precision(prec)
    try:
        torch.manual_seed(11)
        dev = torch.device("cuda:0")
        cfg = {"pretrain_state_dict": None, "bands": 20}
        levels = [learned_ensemble(2, c, cfg).to(dev).eval() for c in (3, 6)]
        hang = Hang2020(20, 7).to(dev).eval()
        live, frozen = MultiStagePredictor(levels), MultiStagePredictor(levels, frozen=True)
        live1, frozen1 = Predictor(hang), Predictor(hang, frozen=True)
        for step in range(3):
            xs = [torch.rand(10, 20, 11, 11, device=dev) for _ in range(2)]
            if step == 2:
                xs[0][:] = 0
            a, b = live(xs), frozen(xs)
            for l in range(2):
                for u, v in zip(a[l], b[l]):
                    assert torch.equal(u, v), (step, l)
            for u, v in zip(live1(xs[1]), frozen1(xs[1])):
                assert torch.equal(u, v), step
        assert frozen._packed and frozen1._packed
        xs = [torch.rand(10, 20, 11, 11, device=dev) for _ in range(2)]
        # running statistics are read in every call: an in-place change is seen by the frozen predictors too
        with torch.no_grad():
            levels[0].year_models[0].conv1.bn1.running_mean.add_(0.05)
            hang.spectral_network.conv2.bn1.running_var.mul_(1.5)
        for u, v in zip(live(xs)[0], frozen(xs)[0]):
            assert torch.equal(u, v)
        for u, v in zip(live1(xs[0]), frozen1(xs[0])):
            assert torch.equal(u, v)
        # a whole-model update made through torch (here: load_state_dict) moves the sentinels' version counters: re-packed
        # without being asked
        for mod in (levels[1], hang):
            sd = {k: (v * 1.25 if v.dim() > 1 else v.clone()) for k, v in mod.state_dict().items()}
            mod.load_state_dict(sd)
        for u, v in zip(live(xs)[1], frozen(xs)[1]):
            assert torch.equal(u, v)
        for u, v in zip(live1(xs[0]), frozen1(xs[0])):
            assert torch.equal(u, v)
        # one written behind torch's back (as the fused trainers write: raw pointers) is NOT seen until refresh()
        before = [t.clone() for t in frozen(xs)[1]]
        before1 = frozen1(xs[0])[0].clone()
        w, w1 = levels[1].year_models[1].conv2.conv_layer.weight, hang.spatial_network.conv2.conv_layer.weight
        for t, f in ((w, 1.5), (w1, 0.5)):
            alias = torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage(), t.storage_offset(), t.shape, t.stride())
            v0 = t._version
            alias.mul_(f)
            assert t._version == v0
        assert torch.equal(frozen(xs)[1][0], before[0]) and torch.equal(frozen1(xs[0])[0], before1)
        assert not torch.equal(live(xs)[1][0], before[0])
        frozen.refresh(); frozen1.refresh()
        for u, v in zip(live(xs)[1], frozen(xs)[1]):
            assert torch.equal(u, v)
        for u, v in zip(live1(xs[0]), frozen1(xs[0])):
            assert torch.equal(u, v)
    finally:
        deeptreeattention_amd.set_default_precision(default)
