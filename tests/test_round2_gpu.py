"""GPU tests of the round-2 additions around the hot path: the three-head loss option against the reference's own golden
(tests/golden/three_head.npz), the Trainer-shaped loop (BASELINE configs[0]: vanilla_CNN(5, 3) through
training_step / validation_step), and the robustness fixes (stale Predictor tables, aliased loss tensors, out-of-range
labels, vanilla_CNN patch-size guard)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import hang2020_np as O
from oracle import prng

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _load(m, p):
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
    return m.to(dev()).train()


@pytest.mark.parametrize("tag", ["hang/", "spectral/"])
def test_three_head_loss_vs_reference_golden(golden, tag):
    """FusedTrainer(three_head_loss=True): loss = sum over every classifier head of the class-weighted CE (six heads for
    Hang2020, three for a lone sub-network), as the reference's modules produce it when their three heads are all used
    (Hang2020.py:204, :240): heads, loss, every gradient, BatchNorm buffers and two Adam steps vs the reference."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    g = golden("three_head.npz")
    bands, classes, B = 20, 7, 6
    x = torch.from_numpy(prng.uniform01(92, 1, (B, bands, 11, 11))).to(dev())
    y = torch.from_numpy(prng.randint(92, 2, (B,), classes)).to(dev())
    w = torch.from_numpy((0.1 + (np.arange(classes) % 7)).astype(np.float32))
    if tag == "hang/":
        m = _load(H.Hang2020(bands, classes), O.init_params(O.hang2020_spec(bands, classes), seed=91))
    else:
        m = _load(H.spectral_network(bands, classes), O.init_params(O.subnet_spec("spectral", bands, classes), seed=91))
    alpha0 = float(m.alpha) if tag == "hang/" else None
    tr = FusedTrainer(m, lr=1e-3, loss_weight=w, three_head_loss=True, keep_grads=True)
    for step in range(2):
        loss = tr.train_step(x, y)
        ref = float(g[f"{tag}loss_step{step}"])
        assert abs(float(loss) - ref) / ref < 2e-4, (step, float(loss), ref)
        if step == 0:
            hs = tr.head_scores
            k = 0
            for net in range(hs.shape[0]):
                for hd in range(3):
                    assert rel_l2(hs[net, hd].cpu().numpy(), g[f"{tag}head{k}"]) < 2e-4, k
                    k += 1
            none = set(g[f"{tag}grad_none"].tolist())
            for name, prm in m.named_parameters():
                if name in none or name.endswith("conv_layer.bias") or prm.dtype != torch.float32:
                    continue
                got = tr.grad_of(prm).cpu().numpy()
                ref_n = float(g[f"{tag}grad_norm/{name}"])
                assert abs(np.linalg.norm(got.astype(np.float64)) - ref_n) <= 1e-3 * ref_n, name
                if f"{tag}grad_full/{name}" in g and np.any(g[f"{tag}grad_full/{name}"]):
                    assert rel_l2(got, g[f"{tag}grad_full/{name}"]) < 1e-3, name
            sd = m.state_dict()
            for name in sd:
                if O.is_buffer(name):
                    assert rel_l2(sd[name].cpu().numpy(), g[f"{tag}buf1/{name}"]) < 2e-4, name
    sd = m.state_dict()
    for name, prm in m.named_parameters():
        if name.endswith("conv_layer.bias"):
            continue
        ref_n = float(g[f"{tag}p2_norm/{name}"])
        assert abs(float(prm.double().norm()) - ref_n) <= 1e-3 * max(ref_n, 1e-12), name
    if alpha0 is not None:
        assert float(m.alpha) == alpha0          # the blend is not on this loss' graph: alpha untouched, as in torch


def test_fit_loop_vanilla_cnn_config1():
    """BASELINE configs[0]: vanilla_CNN(bands=5, classes=3) driven through training_step / validation_step by the
    Trainer-shaped loop with the reference's plateau scheduler; the reference's own living test of this plumbing is
    tests/test_multi_stage.py:12-19 (fit one epoch, losses finite)."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    from deeptreeattention_amd.loop import PlateauScheduler, SyntheticTreeDataset, fit
    torch.manual_seed(0)
    m = H.vanilla_CNN(bands=5, classes=3).to(dev()).train()
    tr = FusedTrainer(m, lr=5e-3)
    train = SyntheticTreeDataset(96, bands=5, classes=3, seed=1)
    val = SyntheticTreeDataset(32, bands=5, classes=3, seed=2)
    # learnable synthetic labels: the class is the brightest of the first three bands' means
    for ds in (train, val):
        ds.labels = ds.hsi[:, :3].mean(dim=(2, 3)).argmax(1)
    sched = PlateauScheduler(tr, patience=1)
    hist = fit(tr, train, val, epochs=6, batch_size=16, scheduler=sched)
    assert len(hist) == 6 and all(np.isfinite(h["train_loss"]) and np.isfinite(h["val_loss"]) for h in hist)
    assert hist[-1]["train_loss"] < hist[0]["train_loss"] - 0.05
    assert tr.step_count == 6 * 6
    assert int(m.conv1.bn1.num_batches_tracked) == 6 * 6          # validation ran in eval mode, as under Lightning
    assert m.training


def test_fit_loop_hang2020_and_losses_are_fresh_tensors():
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    from deeptreeattention_amd.loop import SyntheticTreeDataset
    torch.manual_seed(1)
    m = H.Hang2020(bands=12, classes=4).to(dev()).train()
    tr = FusedTrainer(m, lr=1e-3)
    ds = SyntheticTreeDataset(48, bands=12, classes=4, seed=3)
    losses = [tr.training_step(b, i) for i, b in enumerate(ds.loader(16))]
    vals = [float(v) for v in losses]
    assert len({v.data_ptr() for v in losses}) == 3 and len(set(vals)) == 3     # collected losses are not aliases
    m.eval()
    v = tr.validation_step(next(ds.loader(16)))
    assert np.isfinite(float(v))
    logits, loss = tr.forward_loss(ds.hsi[:16], ds.labels[:16])
    l2, _ = tr.forward_loss(ds.hsi[16:32], ds.labels[16:32])
    assert logits.data_ptr() != l2.data_ptr()


def test_predictor_follows_rehomed_parameters():
    """predict() before and after a FusedTrainer moves the parameters into its flat buffer (and trains): the cached
    pointer tables must follow (the old storages are freed)."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd import engine
    torch.manual_seed(2)
    m = H.Hang2020(bands=9, classes=5).to(dev()).eval()
    x = torch.rand(8, 9, 11, 11, device=dev())
    y = torch.randint(0, 5, (8,), device=dev())
    p0, idx0, _ = engine.predict(m, x)
    p0 = p0.clone()
    with torch.no_grad():
        assert rel_l2(p0.cpu().numpy(), torch.softmax(m(x), 1).cpu().numpy()) < 1e-5
    m.train()
    tr = engine.FusedTrainer(m, lr=1e-2)
    junk = [torch.full((1 << 20,), float("nan"), device=dev()) for _ in range(4)]   # recycle the freed parameter blocks
    for _ in range(3):
        tr.train_step(x, y)
    m.eval()
    p1, _, _ = engine.predict(m, x)
    with torch.no_grad():
        want = torch.softmax(m(x), 1)
    assert torch.isfinite(p1).all()
    assert rel_l2(p1.cpu().numpy(), want.cpu().numpy()) < 1e-5
    assert rel_l2(p1.cpu().numpy(), p0.cpu().numpy()) > 1e-3       # the three steps did change the model
    del junk
    import copy
    copy.deepcopy(m)                                               # nothing un-copyable was hung on the module


def test_cross_entropy_flags_out_of_range_labels():
    from deeptreeattention_amd import _lib
    L = _lib.lib()
    B, classes = 6, 5
    z = torch.randn(B, classes, device=dev())
    loss = torch.zeros((), device=dev())
    dl = torch.empty_like(z)
    scratch = torch.empty(B + 1, device=dev())

    def run(y):
        _lib.check(L.dta_weighted_ce(_lib.ptr(z), _lib.ptr(y), None, B, classes, _lib.ptr(loss), _lib.ptr(dl),
                                     _lib.ptr(scratch), _lib.current_stream_ptr()), "ce")
        return float(loss), dl.clone()
    y = torch.tensor([0, 1, 2, 3, 4, 1], device=dev())
    ref = torch.nn.functional.cross_entropy(z, y)
    l, d = run(y)
    assert abs(l - float(ref)) < 1e-5
    y_ign = y.clone(); y_ign[2] = -100                           # torch's ignore_index: dropped from the mean
    l, d = run(y_ign)
    assert abs(l - float(torch.nn.functional.cross_entropy(z, y_ign))) < 1e-5
    assert float(d[2].abs().max()) == 0.0
    y_bad = y.clone(); y_bad[4] = classes                        # a label-mapping bug must not train silently
    l, d = run(y_bad)
    assert np.isnan(l) and torch.isnan(d[4]).all() and torch.isfinite(d[0]).all()


def test_vanilla_cnn_rejects_patches_its_head_cannot_take():
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd import _lib
    m = H.vanilla_CNN(bands=5, classes=3).to(dev())
    with pytest.raises(RuntimeError, match="512"):
        m(torch.rand(2, 5, 24, 24, device=dev()))
    L = _lib.lib()
    desc = _lib.NetDesc(2, 5, 24, 24, 3, _lib.NET_VANILLA, _lib.DTA_F32, 1, 4, 0.1, 1e-5)
    assert L.dta_net_workspace_bytes(C.byref(desc)) == 0
    assert b"512" in L.dta_last_error()


@pytest.mark.parametrize("precision,classes", [("fp32", 13), ("bf16", 13), ("fp32", 300)])
def test_fused_loss_launch_matches_blend_plus_cross_entropy(precision, classes):
    """dta_net_loss (blend + weighted CE + loss in one launch, last-block finalisation) against the three-launch route
    (k_blend inside the forward, dta_weighted_ce) on the same network and batch: same scores, loss and score gradient
    (300 classes: rows wider than the 256 scores a wave keeps in registers);
    repeated calls leave the block counter clean; ignored (-100) labels drop out of the mean."""
    import copy
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    torch.manual_seed(3)
    m1 = H.Hang2020(24, classes, precision=precision).cuda().train()
    m2 = copy.deepcopy(m1)
    w = torch.rand(classes) + 0.5
    t1 = FusedTrainer(m1, lr=1e-3, loss_weight=w)
    t2 = FusedTrainer(m2, lr=1e-3, loss_weight=w)
    t2.external_loss = True                      # forward blends, the loss is the stand-alone dta_weighted_ce
    x = torch.rand(37, 24, 11, 11, device="cuda")
    y = torch.randint(0, classes, (37,), device="cuda")
    y[5] = -100
    for rep in range(3):                         # the counter must come back to zero every time
        y1 = t1._labels(y)
        lg1 = t1._forward_scores(x); l1 = t1._loss(lg1, y1, True)
        lg2 = t2._forward_scores(x); l2 = t2._loss(lg2, y1, True)
        assert t1.fused_loss and not t2.fused_loss
        torch.cuda.synchronize()
        assert torch.allclose(t1.logits, t2.logits, rtol=1e-6, atol=1e-6)
        assert abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l2))
        assert torch.allclose(t1.dlogits, t2.dlogits, rtol=1e-5, atol=1e-8)
        assert float(t1.dlogits[5].abs().max()) == 0.0
        assert int(t1.ce_scratch.view(torch.int32)[-1]) == 0
    # an out-of-range label poisons the loss (same contract as dta_weighted_ce)
    y[7] = classes + 86
    assert torch.isnan(t1._loss(t1._forward_scores(x), t1._labels(y), True))


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_train_steps_are_bit_reproducible(precision):
    """No kernel of the step sums in a run-dependent order (split-K slabs and the small GEMMs are reduced in fixed order,
    the loss by the last block over a fixed sequence, d(alpha)'s atomics add grid-aligned numbers exactly): the same
    steps from the same state give the same bits -- weights, BatchNorm statistics, alpha, losses."""
    import copy
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd.engine import FusedTrainer
    torch.manual_seed(11)
    m0 = H.Hang2020(40, 9, precision=precision).cuda().train()
    x = [torch.rand(96, 40, 11, 11, device="cuda") for _ in range(2)]
    y = [torch.randint(0, 9, (96,), device="cuda") for _ in range(2)]
    runs = []
    for rep in range(3):
        m = copy.deepcopy(m0)
        tr = FusedTrainer(m, lr=1e-3)
        losses = [tr.train_step(x[i % 2], y[i % 2]) for i in range(6)]
        torch.cuda.synchronize()
        runs.append(({k: v.clone() for k, v in m.state_dict().items()}, [float(l) for l in losses]))
    for sd, ls in runs[1:]:
        assert ls == runs[0][1]
        for k in sd:
            assert torch.equal(sd[k], runs[0][0][k]), k


def test_fused_loss_over_many_launches_on_a_multi_xcd_grid():
    """The fused loss launch hands its row terms to the last block through device-scope atomics without a fence
    (csrc/heads.hip k_blend_ce).  300 back-to-back launches at B = 1024 (256 workgroups, spread over all eight XCDs) with
    fresh scores every time: each loss must equal the stand-alone dta_weighted_ce of the same blended scores -- a stale
    row term picked up by the finalising block would show as a mismatch of the order 1/B."""
    import ctypes as C
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd import _lib
    from deeptreeattention_amd.engine import FusedTrainer
    torch.manual_seed(21)
    B, classes = 1024, 200
    m = H.Hang2020(16, classes, precision="bf16").cuda().train()
    tr = FusedTrainer(m, lr=1e-3, loss_weight=torch.rand(classes) + 0.5)
    L = _lib.lib()
    losses, refs = [], []
    ref_loss = torch.zeros(300, device="cuda")
    scratch = torch.zeros(B + 1, device="cuda")
    for i in range(300):
        x = torch.rand(B, 16, 11, 11, device="cuda")
        y = torch.randint(0, classes, (B,), device="cuda")
        lg = tr._forward_scores(x)
        losses.append(tr._loss(lg, y, True))
        _lib.check(L.dta_weighted_ce(_lib.ptr(tr.logits), _lib.ptr(y), _lib.ptr(tr.loss_weight), B, classes,
                                     C.c_void_p(ref_loss.data_ptr() + 4 * i), None, _lib.ptr(scratch),
                                     _lib.current_stream_ptr()), "dta_weighted_ce")
    torch.cuda.synchronize()
    got = torch.stack(losses)
    assert torch.allclose(got, ref_loss, rtol=2e-6, atol=0), float((got - ref_loss).abs().max())


def test_nan_input_propagates_through_bf16_inference():
    """A NaN pixel (nodata) must reach the scores in eval mode as it does in the reference: the saturating half store of
    the conv outputs clamps only ordered values (csrc/common.h pack2_fmt / st_fmt)."""
    from deeptreeattention_amd import Hang2020 as H
    torch.manual_seed(2)
    m = H.Hang2020(24, 7, precision="bf16").cuda().eval()
    x = torch.rand(5, 24, 11, 11, device="cuda")
    x[3, 7, 4, 4] = float("nan")
    with torch.no_grad():
        out = m(x)
    assert torch.isnan(out[3]).all()
    assert torch.isfinite(out[[0, 1, 2, 4]]).all()
