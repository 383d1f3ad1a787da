"""Pin the NumPy oracle against outputs of the reference itself (tests/golden/*.npz, produced by
tests/golden/make_golden.py from /root/reference/src/models/Hang2020.py and year.py).  CPU only."""
import numpy as np
import pytest

from oracle import hang2020_np as O
from oracle import prng
from conftest import rel_l2

TOL = 2e-5  # fp64 oracle vs the reference's fp32 torch run


def sample_idx(n, count=256):
    return (prng.hash_u64(7, 99, count) % np.uint64(n)).astype(np.int64)


def test_prng_is_stable():
    # known-answer: guards the fixtures' input recipe against accidental edits of oracle/prng.py
    u = prng.uniform01(3, 5, (4,))
    assert u.dtype == np.float32 and np.all((u >= 0) & (u < 1))
    assert prng.hash_u64(1, 2, 3).tolist() == prng.hash_u64(1, 2, 3).tolist()
    assert prng.randint(9, 2, (100,), 7).max() < 7


@pytest.mark.parametrize("name,cin,cout,pool", [("cm_nopool", 5, 32, False), ("cm_pool", 32, 64, True)])
def test_conv_module(golden, name, cin, cout, pool):
    g = golden("modules.npz")
    p = O.init_params(O.conv_module_spec("", cin, cout), seed=11)
    x = prng.uniform(12, 1, (3, cin, 11, 11), -1, 1)
    z, cache, upd = O.conv_module_fwd(p, "", x, pool, True, np.float64)
    assert rel_l2(z, g[f"{name}/z"]) < TOL
    dz = prng.uniform(12, 3, z.shape, -1, 1).astype(np.float64)
    dx, grads = O.conv_module_bwd(cache, "", dz)
    assert rel_l2(dx, g[f"{name}/dx"]) < TOL
    for k, v in grads.items():
        if k.endswith("conv_layer.bias"):   # analytically zero under train-mode BN
            assert np.abs(v).max() < 1e-4
            continue
        assert rel_l2(v, g[f"{name}/g/{k}"]) < TOL, k
    for k, v in upd.items():
        assert rel_l2(v, g[f"{name}/buf/{k}"]) < TOL, k
    p2 = dict(p)
    z_eval, _, _ = O.conv_module_fwd(p2, "", x, pool, False, np.float64)
    # eval forward in the golden run happened after one training forward: use its updated buffers
    p2.update({k: g[f"{name}/buf/{k}"] for k in upd})
    z_eval, _, _ = O.conv_module_fwd(p2, "", x, pool, False, np.float64)
    assert rel_l2(z_eval, g[f"{name}/z_eval"]) < TOL


@pytest.mark.parametrize("C,hw", [(32, 11), (64, 5), (128, 2)])
@pytest.mark.parametrize("kind", ["spectral", "spatial"])
def test_attention(golden, kind, C, hw):
    g = golden("modules.npz")
    name = f"{kind}_att{C}"
    specf = O.spectral_attention_spec if kind == "spectral" else O.spatial_attention_spec
    fwd = O.spectral_attention_fwd if kind == "spectral" else O.spatial_attention_fwd
    bwd = O.spectral_attention_bwd if kind == "spectral" else O.spatial_attention_bwd
    p = O.init_params(specf("", C), seed=21)
    x = prng.uniform01(22, C, (3, C, hw, hw))
    a, f, cache = fwd(p, "", x, np.float64)
    assert rel_l2(a, g[f"{name}/a"]) < TOL
    assert rel_l2(f, g[f"{name}/f"]) < TOL
    da = prng.uniform(22, 3, a.shape, -1, 1).astype(np.float64)
    df = prng.uniform(22, 4, f.shape, -1, 1).astype(np.float64)
    dz, grads = bwd(cache, "", da, df)
    assert rel_l2(dz, g[f"{name}/dx"]) < TOL
    for k, v in grads.items():
        ref = g[f"{name}/g/{k}"]
        assert v.shape == ref.shape
        assert rel_l2(v, ref) < TOL, k
    if kind == "spectral":
        k = O.SPECTRAL_K[C]
        w = g[f"{name}/g/attention_conv1.weight"]
        off = np.delete(w, k // 2, axis=2)
        assert np.all(off == 0)  # length-1 sequence: only the centre tap is live


def check_hang(golden, fname, bands, classes, B, seed, dt, tol):
    g = golden(fname)
    spec = O.hang2020_spec(bands, classes)
    p = O.init_params(spec, seed=seed)
    keys = [k for k, *_ in spec]
    assert list(g["keys"]) == keys, "state_dict key order differs from the reference"
    x = prng.uniform01(seed + 1, 1, (B, bands, 11, 11))
    y = prng.randint(seed + 1, 2, (B,), classes)
    w_non = (0.1 + (np.arange(classes) % 7)).astype(np.float32)
    ev, _, _ = O.hang2020_fwd(p, x, False, dt)
    assert rel_l2(ev, g["eval_logits"]) < tol
    logits, cache, upd = O.hang2020_fwd(p, x, True, dt)
    assert rel_l2(logits, g["logits"]) < tol
    for i in range(3):
        assert rel_l2(cache["s_spec"][i], g[f"spec_head{i + 1}"]) < tol
        assert rel_l2(cache["s_spat"][i], g[f"spat_head{i + 1}"]) < tol
    assert abs(cache["w"] - g["sigmoid_alpha"]) < 1e-12
    loss, dl = O.weighted_cross_entropy(logits, y, w_non)
    assert abs(loss - g["loss_non"]) / abs(g["loss_non"]) < tol
    loss_u, _ = O.weighted_cross_entropy(logits, y, np.ones(classes, np.float32))
    assert abs(loss_u - g["loss_uni"]) / abs(g["loss_uni"]) < tol
    grads = O.hang2020_bwd(p, cache, dl.astype(dt), dt)
    none = set(g["grad_none"].tolist())
    assert none == {k for k, *_ in spec if not O.is_buffer(k) and k not in grads}
    assert len(none) == 8
    tot = 0.0
    for k, v in grads.items():
        tot += float((np.asarray(v, np.float64) ** 2).sum())
        if k.endswith("conv_layer.bias"):
            assert np.abs(v).max() < 1e-5
            continue
        nref = float(g[f"grad_norm/{k}"])
        assert abs(np.sqrt((np.asarray(v, np.float64) ** 2).sum()) - nref) <= tol * max(nref, 1e-12) * 4, k
        if f"grad_full/{k}" in g:
            assert rel_l2(v, g[f"grad_full/{k}"]) < tol * 4, k
        else:
            s = np.asarray(v).reshape(-1)[sample_idx(np.asarray(v).size)]
            assert rel_l2(s, g[f"grad_samp/{k}"]) < tol * 4, k
    assert abs(np.sqrt(tot) - g["grad_total_norm"]) / g["grad_total_norm"] < tol * 4
    for k, v in upd.items():
        assert rel_l2(v, g[f"buf1/{k}"]) < tol, k
    return p, grads, upd, g


def test_hang2020_small_fp64_oracle(golden):
    check_hang(golden, "hang2020_3_10.npz", 3, 10, 4, 41, np.float64, TOL)


def test_hang2020_small_fp32_oracle(golden):
    check_hang(golden, "hang2020_3_10.npz", 3, 10, 4, 41, np.float32, 2e-4)


def test_hang2020_full_and_adam(golden):
    p, grads, upd, g = check_hang(golden, "hang2020_369_200.npz", 369, 200, 8, 31, np.float64, TOL)
    # three Adam steps (lr 1e-3), compared after step 1 and step 3
    x = prng.uniform01(32, 1, (8, 369, 11, 11))
    y = prng.randint(32, 2, (8,), 200)
    w_non = (0.1 + (np.arange(200) % 7)).astype(np.float32)
    state = {}
    for step in range(3):
        logits, cache, upd = O.hang2020_fwd(p, x, True, np.float64)
        loss, dl = O.weighted_cross_entropy(logits, y, w_non)
        assert abs(loss - g[f"loss_step{step}"]) / g[f"loss_step{step}"] < 1e-4
        grads = O.hang2020_bwd(p, cache, dl, np.float64)
        p = O.adam_step(p, grads, state, lr=1e-3)
        p.update(upd)
        if step in (0, 2):
            for k, *_ in O.hang2020_spec(369, 200):
                if O.is_buffer(k):
                    continue
                a = np.asarray(p[k]).reshape(-1)
                s = a[(prng.hash_u64(7, 99, 64) % np.uint64(a.size)).astype(np.int64)] if a.size > 64 else a
                # Adam's first steps move every weight by ~lr regardless of |g|: conv biases under BN
                # have pure-noise gradients, so their updates are noise-signed; skip those.
                if k.endswith("conv_layer.bias"):
                    continue
                assert rel_l2(s, g[f"p{step + 1}_samp/{k}"]) < 2e-3, (step, k)


def test_fp64_truth(golden):
    g32 = golden("hang2020_369_200.npz")
    g64 = golden("hang2020_369_200_fp64.npz")
    assert rel_l2(g32["logits"], g64["logits"]) < 1e-5   # fp32 vs fp64 of the reference


def test_subnets_and_callers(golden):
    g = golden("subnets.npz")
    bands, classes, B = 16, 7, 2
    for kind, hw in (("spectral", 24), ("spectral", 11), ("spatial", 11)):
        tag = f"{kind}{hw}/"
        p = O.init_params(O.subnet_spec(kind, bands, classes), seed=51)
        x = prng.uniform01(52, hw, (B, bands, hw, hw))
        s, cache, upd = O.subnet_fwd(p, "", kind, x, True, np.float64)
        ds = [prng.uniform(52, 10 + i, (B, classes), -1, 1).astype(np.float64) for i in range(3)]
        for i in range(3):
            assert rel_l2(s[i], g[f"{tag}head{i + 1}"]) < TOL
        grads = O.subnet_bwd(p, "", cache, ds, np.float64)
        for k, v in grads.items():
            if k.endswith("conv_layer.bias"):
                continue
            assert rel_l2(v, g[f"{tag}g/{k}"]) < 1e-4, (tag, k)
        for k, v in upd.items():
            assert rel_l2(v, g[f"{tag}buf/{k}"]) < TOL
    # learned_ensemble
    p = O.init_params(O.learned_ensemble_spec(3, bands, classes), seed=61)
    imgs = [prng.uniform01(62, yy, (B, bands, 11, 11)) for yy in range(3)]
    imgs[1] = np.zeros_like(imgs[1])
    s, cache, _ = O.learned_ensemble_fwd(p, imgs, True, np.float64)
    assert rel_l2(s, g["ens/score"]) < TOL
    grads = O.learned_ensemble_bwd(p, cache, prng.uniform(62, 9, (B, classes), -1, 1).astype(np.float64), np.float64)
    none = set(g["ens/none"].tolist())
    assert all(k.startswith("year_models.1.") or "classifier1" in k or "classifier2" in k for k in none)
    for k, v in grads.items():
        if k.endswith("conv_layer.bias"):
            continue
        ref = float(g[f"ens/gnorm/{k}"])
        assert abs(np.sqrt((v ** 2).sum()) - ref) <= 1e-4 * max(ref, 1e-9), k
    # vanilla_CNN
    p = O.init_params(O.vanilla_spec(5, 3), seed=71)
    x = prng.uniform01(72, 1, (2, 5, 11, 11))
    y = prng.randint(72, 2, (2,), 3)
    lg, cache, _ = O.vanilla_fwd(p, x, True, np.float64)
    assert rel_l2(lg, g["vanilla/logits"]) < TOL
    loss, dl = O.weighted_cross_entropy(lg, y, np.ones(3, np.float32))
    assert abs(loss - g["vanilla/loss"]) / g["vanilla/loss"] < TOL
    grads = O.vanilla_bwd(p, cache, dl, np.float64)
    for k, v in grads.items():
        if k.endswith("conv_layer.bias"):
            continue
        assert rel_l2(v, g[f"vanilla/g/{k}"]) < 1e-4, k


def test_torch_port_matches_reference_golden(golden):
    """The torch-eager port (bench.py's cpu_baseline) against the reference's own outputs."""
    import torch
    from oracle import hang2020_torch as TP
    g = golden("hang2020_3_10.npz")
    p = TP.to_tensors(O.init_params(O.hang2020_spec(3, 10), seed=41))
    x = torch.from_numpy(prng.uniform01(42, 1, (4, 3, 11, 11)))
    y = torch.from_numpy(prng.randint(42, 2, (4,), 10))
    w = torch.from_numpy((0.1 + (np.arange(10) % 7)).astype(np.float32))
    step = TP.TrainStep(p, lr=1e-3, loss_weight=w)
    for i in range(3):
        logits, loss = step(x, y)
        if i == 0:
            assert rel_l2(logits.numpy(), g["logits"]) < 1e-5
        assert abs(loss.item() - g[f"loss_step{i}"]) / g[f"loss_step{i}"] < 1e-4
    for k, t in p.items():
        if not t.requires_grad or k.endswith("conv_layer.bias"):
            continue
        a = t.detach().numpy().reshape(-1)
        s = a[(prng.hash_u64(7, 99, 64) % np.uint64(a.size)).astype(np.int64)] if a.size > 64 else a
        assert rel_l2(s, g[f"p3_samp/{k}"]) < 2e-3, k


def _ensemble_step_inputs(step, years, B, bands, classes):
    imgs = [prng.uniform01(82 + step, yy, (B, bands, 11, 11)) for yy in range(years)]
    if step == 1:
        imgs[2] = np.zeros_like(imgs[2])
    if step == 2:
        imgs[0] = np.zeros_like(imgs[0])
    return imgs, prng.randint(82 + step, 7, (B,), classes)


def test_torch_oracle_ensemble_steps_match_reference_golden(golden):
    """Year-ensemble train steps (MultiStage recipe) incl. two steps with an all-zero year: the torch restatement
    against the reference's learned_ensemble + F.cross_entropy + Adam."""
    import torch
    from oracle import hang2020_torch as OT
    g = golden("ensemble_steps.npz")
    years, bands, classes, B, lr = 3, 16, 7, 6, 1e-3
    p = OT.to_tensors(O.init_params(O.learned_ensemble_spec(years, bands, classes), seed=81))
    w = torch.from_numpy((0.1 + (np.arange(classes) % 7)).astype(np.float32))
    step_fn = OT.EnsembleTrainStep(p, lr, w)
    for step in range(4):
        imgs, y = _ensemble_step_inputs(step, years, B, bands, classes)
        s, loss = step_fn([torch.from_numpy(a) for a in imgs], torch.from_numpy(y))
        assert rel_l2(s.numpy(), g[f"step{step}/score"]) < 1e-4, step
        assert abs(float(loss) - float(g[f"step{step}/loss"])) < 1e-4 * abs(float(g[f"step{step}/loss"]))
        for k, t in p.items():
            if not t.requires_grad:
                assert rel_l2(t.numpy(), g[f"step{step}/buf/{k}"]) < 1e-4, (step, k)
                continue
            a = t.detach().numpy()
            assert abs(np.sqrt((a.astype(np.float64) ** 2).sum()) - float(g[f"step{step}/pnorm/{k}"])) \
                <= 1e-5 * float(g[f"step{step}/pnorm/{k}"]), (step, k)


def test_torch_oracle_multistage_steps_match_reference_golden(golden):
    """The reference's MultiStage loop over two levels with different class counts and learning rates (multi_stage.py:41-66,
    :258-288; three steps, two of them with an all-zero year in one level): the torch restatement, one EnsembleTrainStep
    per level, against the reference's own learned_ensembles + per-level F.cross_entropy + per-level Adam."""
    import torch
    from oracle import hang2020_torch as OT
    from oracle.recipes import MULTISTAGE as c, multistage_inputs, multistage_weight
    g = golden("multistage_steps.npz")
    steps = []
    for l, classes in enumerate(c["classes"]):
        p = OT.to_tensors(O.init_params(O.learned_ensemble_spec(c["years"], c["bands"], classes), seed=301 + l))
        steps.append((p, OT.EnsembleTrainStep(p, c["lrs"][l], torch.from_numpy(multistage_weight(classes)))))
    for step in range(c["steps"]):
        for l, (p, fn) in enumerate(steps):
            imgs, y = multistage_inputs(step, l, c["years"], c["B"], c["bands"], c["classes"][l])
            s, loss = fn([torch.from_numpy(a) for a in imgs], torch.from_numpy(y))
            tag = f"step{step}/level{l}"
            assert rel_l2(s.numpy(), g[f"{tag}/score"]) < 1e-4, tag
            assert abs(float(loss) - float(g[f"{tag}/loss"])) < 1e-4 * abs(float(g[f"{tag}/loss"])), tag
            for k, t in p.items():
                if not t.requires_grad:
                    assert rel_l2(t.numpy(), g[f"{tag}/buf/{k}"]) < 1e-4, (tag, k)
                    continue
                a = t.detach().numpy()
                ref = float(g[f"{tag}/pnorm/{k}"])
                assert abs(np.sqrt((a.astype(np.float64) ** 2).sum()) - ref) <= 1e-5 * ref, (tag, k)


def _metadata_params(g, bands, classes):
    """Sensor weights from the portable PRNG, the small site branch / fusion layer from the stored torch init."""
    import torch
    from oracle import hang2020_torch as OT
    p = OT.to_tensors({"sensor_model." + k: v for k, v in O.init_params(O.hang2020_spec(bands, classes), seed=9).items()})
    for k in g.files:
        if k.startswith("init/"):
            t = torch.tensor(g[k])
            if t.is_floating_point() and "running_" not in k:
                t.requires_grad_(True)
            p[k[len("init/"):]] = t
    return p


def test_torch_oracle_metadata_fusion_matches_reference_golden(golden):
    """metadata_sensor_fusion of the reference (imported with the GIS / Lightning stack stubbed): eval output, and a
    train-mode forward/backward of the unweighted-CE step with the site branch's dropout disabled."""
    import torch
    from oracle import hang2020_torch as OT
    g = golden("metadata.npz")
    bands, classes, sites, B = 12, 5, 4, 6
    x = torch.from_numpy(prng.uniform01(10, 1, (B, bands, 11, 11)))
    site = torch.from_numpy(prng.randint(10, 2, (B,), sites))
    y = torch.from_numpy(prng.randint(10, 3, (B,), classes))
    p = _metadata_params(g, bands, classes)
    with torch.no_grad():
        assert rel_l2(OT.metadata_sensor_fusion(p, x, site, False).numpy(), g["eval/out"]) < 1e-5
    out = OT.metadata_sensor_fusion(p, x, site, True, dropout_p=0.0)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    assert rel_l2(out.detach().numpy(), g["train/out"]) < 1e-5
    assert abs(float(loss.detach()) - float(g["train/loss"])) < 1e-5 * abs(float(g["train/loss"]))
    none = set(g["train/none"].tolist())
    for k, t in p.items():
        if not t.requires_grad:
            continue
        if k in none:
            assert t.grad is None or not t.grad.any(), k
            continue
        ref = float(g[f"train/gnorm/{k}"])
        if k.endswith("conv_layer.bias"):
            continue
        assert abs(float(t.grad.double().norm()) - ref) <= 1e-4 * max(ref, 1e-9), k


def _three_head_oracle(p, kind, x, y, w, dt=np.float64):
    """Sum over all classifier heads of the class-weighted cross-entropy (the Hang et al. multi-head loss) through the
    oracle's sub-network forward/backward: loss, per-head scores, gradients, BatchNorm updates."""
    nets = [("spectral_network.", "spectral"), ("spatial_network.", "spatial")] if kind == "hang" else [("", kind)]
    heads, grads, upd, loss = [], {}, {}, 0.0
    for pre, k in nets:
        s, cache, u = O.subnet_fwd(p, pre, k, x, True, dt)
        ds = []
        for h in s:
            l, d = O.weighted_cross_entropy(h, y, w)
            loss += l
            ds.append(d.astype(dt))
        grads.update(O.subnet_bwd(p, pre, cache, ds, dt))
        upd.update(u)
        heads += list(s)
    return loss, heads, grads, upd


def test_three_head_loss_oracle_vs_reference_golden(golden):
    """tests/golden/three_head.npz: the reference's sub-networks' three heads each (Hang2020.py:204, :240) summed into
    one class-weighted loss, gradients and two torch-Adam steps (make_golden.case_three_head)."""
    g = golden("three_head.npz")
    bands, classes, B = 20, 7, 6
    x = prng.uniform01(92, 1, (B, bands, 11, 11))
    y = prng.randint(92, 2, (B,), classes)
    w = (0.1 + (np.arange(classes) % 7)).astype(np.float32)
    for tag, kind, spec in (("hang/", "hang", O.hang2020_spec(bands, classes)),
                            ("spectral/", "spectral", O.subnet_spec("spectral", bands, classes))):
        p = O.init_params(spec, seed=91)
        state = {}
        for step in range(2):
            loss, heads, grads, upd = _three_head_oracle(p, kind, x, y, w)
            assert abs(loss - float(g[f"{tag}loss_step{step}"])) / float(g[f"{tag}loss_step{step}"]) < 2e-5, (tag, step)
            if step == 0:
                for i, h in enumerate(heads):
                    assert rel_l2(h, g[f"{tag}head{i}"]) < 2e-5, (tag, i)
                none = set(g[f"{tag}grad_none"].tolist())
                assert none == ({"alpha"} if kind == "hang" else set())      # the blend is not on this loss' graph
                for k, v in grads.items():
                    if k.endswith("conv_layer.bias") or k in none:
                        continue
                    assert abs(np.sqrt((np.asarray(v, np.float64) ** 2).sum()) - float(g[f"{tag}grad_norm/{k}"])) \
                        <= 2e-4 * float(g[f"{tag}grad_norm/{k}"]), (tag, k)
                    if f"{tag}grad_full/{k}" in g and np.any(g[f"{tag}grad_full/{k}"]):
                        assert rel_l2(v, g[f"{tag}grad_full/{k}"]) < 2e-4, (tag, k)
                for k, v in upd.items():
                    assert rel_l2(v, g[f"{tag}buf1/{k}"]) < 2e-5, (tag, k)
            grads.pop("alpha", None)
            gg = {k: v for k, v in grads.items()}
            if kind == "hang":
                gg["alpha"] = None               # torch's Adam passes over a parameter without gradient
            p = O.adam_step(p, {k: v for k, v in gg.items() if v is not None}, state, lr=1e-3)
            p.update(upd)
        for k in spec:
            name = k[0]
            if O.is_buffer(name) or name.endswith("conv_layer.bias"):
                continue
            ref = float(g[f"{tag}p2_norm/{name}"])
            assert abs(np.sqrt((np.asarray(p[name], np.float64) ** 2).sum()) - ref) <= 2e-4 * max(ref, 1e-12), (tag, name)


def test_oracle_metadata_sensor_branch_at_full_size_vs_reference_golden(golden):
    """BASELINE configs[3] at its real size (369 bands, 200 classes, 23 sites, B=64): the NumPy oracle's Hang2020 branch
    against what the reference's metadata_sensor_fusion produced -- the HSI scores in eval and train mode, and, from the
    reference's own d(loss)/d(HSI scores), every sensor gradient norm."""
    g = golden("metadata_full.npz")
    bands, classes, B = 369, 200, 64
    p = O.init_params(O.hang2020_spec(bands, classes), seed=21)
    x = prng.uniform01(30, 1, (B, bands, 11, 11))
    logits, _, _ = O.hang2020_fwd(p, x, False, np.float64)
    assert rel_l2(logits, g["eval/hsi"]) < TOL
    logits, cache, _ = O.hang2020_fwd(p, x, True, np.float64)
    assert rel_l2(logits, g["train/hsi"]) < TOL
    grads = O.hang2020_bwd(p, cache, g["train/dhsi"].astype(np.float64), np.float64)
    checked = 0
    for k, v in grads.items():
        key = f"train/gnorm/sensor_model.{k}"
        if key not in g.files or k.endswith("conv_layer.bias"):
            continue
        ref = float(g[key])
        # (+2e-9: the reference's float32 run leaves that much noise on single-scalar gradients of size 1e-6)
        assert abs(np.linalg.norm(np.asarray(v, np.float64)) - ref) <= 1e-4 * ref + 2e-9, k
        checked += 1
    assert checked >= 50


@pytest.mark.parametrize("tag,bands,classes,B,seed_p,seed_x,weighted", [("hang9/", 20, 7, 9, 5, 6, False), ("hang48_421/", 48, 11, 421, 3, 461, True)])
def test_oracle_bf16_mode_against_the_references_own_bf16_run(bf16_yardstick, tag, bands, classes, B, seed_p, seed_x, weighted):
    """The oracle's bf16 mode (the model of the kernels' roundings the GPU tests use for implementation exactness) against
    the yardstick taken from the reference itself (tests/golden/bf16_autocast.npz: the reference under
    torch.autocast("cpu", torch.bfloat16) next to itself in fp32): the rounding model is no further from the exact step
    than 1.5 x what the reference's own bf16 run is -- scores, loss, every large tensor, the whole gradient vector."""
    p = O.init_params(O.hang2020_spec(bands, classes), seed=seed_p)
    x = prng.uniform01(seed_x, 1, (B, bands, 11, 11))
    y = prng.randint(seed_x, 2, (B,), classes)
    w = (0.1 + (np.arange(classes) % 7)).astype(np.float32) if weighted else np.ones(classes, np.float32)

    def run(q):
        O.bf16_mode(q)
        try:
            logits, cache, _ = O.hang2020_fwd(p, x, True, np.float64)
            loss, dl = O.weighted_cross_entropy(logits, y, w)
            return logits, loss, O.hang2020_bwd(p, cache, dl, np.float64)
        finally:
            O.bf16_mode(False)
    e_logits, e_loss, e_g = run(False)
    assert abs(e_loss - bf16_yardstick.ref(tag + "loss_fp32")) < 1e-4 * e_loss       # same step as the fixture's fp32 leg
    q_logits, q_loss, q_g = run(True)
    assert rel_l2(q_logits, e_logits) <= bf16_yardstick.bound(tag + "scores_dev")
    assert abs(q_loss - e_loss) / e_loss <= bf16_yardstick.bound(tag + "loss_dev")
    bf16_yardstick.check_gradients(tag, q_g, e_g, verbose=False, norms=B >= 64)


def test_golden_fixtures_are_the_committed_ones():
    """tests/golden/SHA256SUMS pins every fixture file (written when make_golden.py produced them from /root/reference in the
    build container): a fixture edited by hand, or regenerated by a script other than the committed one, fails here."""
    import hashlib
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    want = dict(line.split()[::-1] for line in open(os.path.join(here, "SHA256SUMS")).read().splitlines() if line.strip())
    have = sorted(f for f in os.listdir(here) if f.endswith(".npz"))
    assert have == sorted(want), (have, sorted(want))
    for f in have:
        assert hashlib.sha256(open(os.path.join(here, f), "rb").read()).hexdigest() == want[f], f
