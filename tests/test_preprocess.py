"""Crop preprocessing (reference src/utils.py:36-79, src/augmentation.py:13-14).
CPU: the NumPy oracle against outputs of the reference's own preprocess_image on the reference's own test crops
(bit-exact) and against torch's nearest interpolation.  GPU: the HIP kernel against the oracle, bit-exact."""
import numpy as np
import pytest
import torch

from oracle import preprocess_np as P


def test_oracle_matches_reference_preprocess_bit_exact(golden):
    g = golden("preprocess.npz")
    names = list(g["names"]) + ["syn", "rgb"]
    for name in names:
        raw = g[f"{name}/raw"]
        want = g[f"{name}/pre"]
        got = P.preprocess_image(raw, True)
        assert got.dtype == np.float32 and got.shape == want.shape
        assert np.array_equal(got, want), name
    assert g["rgb/pre"].shape[0] == 3 and g["syn/pre"].shape[0] == 20      # 3 bands: no clipping; 40 -> 20
    name = names[0]
    for size in (11, 24):
        assert np.array_equal(P.resize_nearest(g[f"{name}/pre"], size), g[f"{name}/resized{size}"])
    for name in names[1:-2]:
        assert np.array_equal(P.load_crop(g[f"{name}/raw"], 11), g[f"{name}/resized11"])
    # constant pixel -> zeros (zero range counts as scale 1); near-constant pixel (range < 10 eps) likewise
    assert not g["syn/pre"][:, 2, 3].any()
    assert g["syn/pre"][:, 4, 4].max() < 1e-6


def test_nearest_index_matches_torch_interpolate():
    for out_size in (11, 24, 7, 32):
        for in_size in list(range(1, 70)) + [100, 127, 255]:
            src = torch.arange(in_size, dtype=torch.float32).reshape(1, 1, in_size, 1)
            want = torch.nn.functional.interpolate(src, size=(out_size, 1), mode="nearest").reshape(-1).numpy().astype(np.int64)
            assert np.array_equal(P.nearest_index(out_size, in_size), want), (out_size, in_size)


def test_training_flip_is_both_flips():
    x = np.arange(2 * 5 * 7, dtype=np.int16).reshape(2, 5, 7)
    a = P.load_crop(x, 4, train=False)
    b = P.load_crop(x, 4, train=True)
    assert np.array_equal(b, a[:, ::-1, ::-1])


# ---------------------------------------------------------------------------------------------- GPU
def _dev():
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_hip_preprocess_reference_crops_bit_exact(golden):
    """The reference's own int16 test crops, band-first (rasterio order) and pixel-interleaved (on-disk order),
    against the outputs of the reference's preprocess_image + NEAREST resize: bit-identical."""
    from deeptreeattention_amd import preprocess as PP
    g = golden("preprocess.npz")
    names = list(g["names"])
    raws = [g[f"{n}/raw"] for n in names]
    want = np.stack([g[f"{n}/resized11"] for n in names])
    got = PP.preprocess_batch(raws, 11, train=False).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    hwc = [np.ascontiguousarray(np.moveaxis(r, 0, 2)) for r in raws]
    got = PP.preprocess_batch(hwc, 11, train=False, pixel_interleaved=True).cpu().numpy()
    assert np.array_equal(got, want)
    got = PP.preprocess_batch(hwc, 11, train=True, pixel_interleaved=True).cpu().numpy()
    assert np.array_equal(got, want[:, :, ::-1, ::-1])
    assert np.array_equal(PP.load_image(raws[0], 24).cpu().numpy(), g[f"{names[0]}/resized24"])
    for n in ("syn", "rgb"):      # float32 with constant / near-constant pixels; 3 uint8 bands (no band clipping)
        assert np.array_equal(PP.preprocess_image(g[f"{n}/raw"], channel_is_first=True).cpu().numpy(), g[f"{n}/pre"])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.uint8])
@pytest.mark.parametrize("size", [11, 24])
def test_hip_preprocess_ragged_batch_vs_oracle(dtype, size):
    """Ragged crops (1x1 up to 40x33), a missing year, both raw layouts, both flip settings: bit-exact vs the oracle."""
    from deeptreeattention_amd import preprocess as PP
    rng = np.random.RandomState(9)
    bands = 369 if dtype != np.uint8 else 30
    shapes = [(1, 1), (5, 7), (11, 11), (12, 9), (23, 40), (40, 33), (2, 30), (17, 3)]
    crops = []
    for h, w in shapes:
        a = rng.randint(-300, 12000, size=(bands, h, w)) if dtype != np.uint8 else rng.randint(0, 256, size=(bands, h, w))
        a = a.astype(dtype)
        if dtype == np.float32:
            a = (a * np.float32(0.37)).astype(np.float32)
        crops.append(a)
    crops[3][:, 0, 0] = 5            # a constant pixel
    crops.insert(2, None)            # a missing year
    for train in (False, True):
        want = np.stack([np.zeros((bands - 20, size, size), np.float32) if c is None else P.load_crop(c, size, train)
                         for c in crops])
        got = PP.preprocess_batch(crops, size, train=train).cpu().numpy()
        assert np.array_equal(got, want), ("chw", train)
        hwc = [None if c is None else np.ascontiguousarray(np.moveaxis(c, 0, 2)) for c in crops]
        got = PP.preprocess_batch(hwc, size, train=train, pixel_interleaved=True).cpu().numpy()
        assert np.array_equal(got, want), ("hwc", train)


@pytest.mark.gpu
def test_hip_preprocess_other_raw_dtypes_go_through_float32():
    """uint16 / float64 crops: the reference converts every dtype with np.asarray(image, dtype='float32')."""
    from deeptreeattention_amd import preprocess as PP
    rng = np.random.RandomState(4)
    for dt in (np.uint16, np.float64, np.int32):
        crops = [(rng.rand(40, 9, 7) * 60000).astype(dt), (rng.rand(40, 5, 12) * 60000).astype(dt)]
        want = np.stack([P.load_crop(c.astype(np.float32), 11, True) for c in crops])
        assert np.array_equal(PP.preprocess_batch(crops, 11, train=True).cpu().numpy(), want), dt


@pytest.mark.gpu
def test_hip_preprocess_feeds_the_network():
    """End to end: raw int16 crops -> device preprocessing -> Hang2020 forward, equal to feeding the oracle's batch."""
    from deeptreeattention_amd import preprocess as PP, Hang2020 as H
    rng = np.random.RandomState(2)
    crops = [rng.randint(0, 9000, size=(389, rng.randint(6, 30), rng.randint(6, 30))).astype(np.int16) for _ in range(16)]
    x = PP.preprocess_batch(crops, 11, train=True)
    assert x.shape == (16, 369, 11, 11) and float(x.min()) == 0.0 and abs(float(x.max()) - 1.0) < 3e-7
    want = torch.from_numpy(np.stack([P.load_crop(c, 11, True) for c in crops])).to(_dev())
    torch.manual_seed(0)
    m = H.Hang2020(369, 12).to(_dev()).eval()
    assert torch.equal(x, want)
    with torch.no_grad():
        assert torch.isfinite(m(x)).all()


def test_oracle_minmax_matches_sklearn_bit_exact_on_random_crops():
    """The restated float32 MinMaxScaler arithmetic against scikit-learn itself (the routine the reference calls,
    src/utils.py:47-49) on random crops, incl. constant and near-constant pixels and negative values."""
    sk = pytest.importorskip("sklearn.preprocessing")
    rng = np.random.RandomState(31)
    for trial in range(20):
        c, h, w = rng.randint(4, 60), rng.randint(1, 9), rng.randint(1, 9)
        img = (rng.rand(c, h, w).astype(np.float32) - np.float32(0.3)) * np.float32(10 ** rng.randint(-3, 5))
        if trial % 3 == 0:
            img[:, 0, 0] = np.float32(2.5)
        if trial % 4 == 0:
            img[:, -1, -1] = np.float32(1.0) + np.arange(c, dtype=np.float32) * np.float32(1e-8)
        data = img.reshape(c, h * w).T
        want = sk.minmax_scale(data.copy(), axis=1).T.reshape(img.shape)
        got = P.minmax_over_bands(img)
        assert got.dtype == want.dtype == np.float32
        assert np.array_equal(got, want), trial


def _bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16).float().numpy()


@pytest.mark.gpu
def test_hip_preprocess_writes_conv_tiles_directly(golden):
    """tiles=True: the same launch writes the first conv's bf16 tiles instead of the float32 batch: bit-identical to
    the reference's outputs rounded to bf16, for both raw layouts, 11x11 and 24x24, with a missing year."""
    from deeptreeattention_amd import preprocess as PP
    g = golden("preprocess.npz")
    names = list(g["names"])
    raws = [g[f"{n}/raw"] for n in names]
    want = _bf16(np.stack([g[f"{n}/resized11"] for n in names]))
    t = PP.preprocess_batch(raws, 11, train=False, tiles=True)
    assert t.shape == (len(names), want.shape[1], 11, 11) and t.tiles.dtype == torch.int16
    assert np.array_equal(t.float().cpu().numpy(), want)
    hwc = [np.ascontiguousarray(np.moveaxis(r, 0, 2)) for r in raws]
    t2 = PP.preprocess_batch(hwc, 11, train=False, pixel_interleaved=True, tiles=True)
    assert torch.equal(t2.tiles, t.tiles)
    # padded channels of the last chunk are zero; a missing crop is all zeros; flips act on the tiles too
    C = want.shape[1]
    full = t.tiles.view(len(names), -1, 121, 16).view(torch.bfloat16).float()
    assert full.shape[1] * 16 >= C and float(full.permute(0, 1, 3, 2).reshape(len(names), -1, 121)[:, C:].abs().max()) == 0.0
    t3 = PP.preprocess_batch([raws[0], None], 24, train=True, tiles=True)
    want24 = _bf16(g[f"{names[0]}/resized24"])[:, ::-1, ::-1]
    got24 = t3.float().cpu().numpy()
    assert np.array_equal(got24[0], want24) and not got24[1].any()


@pytest.mark.gpu
@pytest.mark.parametrize("B", [24, 440])      # 440: the batch size class where the float32 path fuses the conversion
def test_train_step_from_tiles_equals_train_step_from_the_float_batch(B):
    """bf16 mode rounds the network input to bf16 anyway, so a step fed with the preprocessed TILES must equal the step
    fed with the float32 batch of the same crops (reference flow: load_image -> model), without ever building it."""
    from deeptreeattention_amd import Hang2020 as H
    from deeptreeattention_amd import preprocess as PP
    from deeptreeattention_amd.engine import FusedTrainer
    import copy
    rng = np.random.RandomState(3)
    crops = [rng.randint(0, 9000, size=(43, rng.randint(8, 20), rng.randint(8, 20))).astype(np.int16) for _ in range(B)]
    # (both paths run the same kernels in the same order: the same bits)
    y = torch.from_numpy(rng.randint(0, 8, size=B)).to(_dev())
    x32 = PP.preprocess_batch(crops, 11, train=True)
    xt = PP.preprocess_batch(crops, 11, train=True, tiles=True)
    assert np.array_equal(xt.float().cpu().numpy(), _bf16(x32.cpu().numpy()))
    torch.manual_seed(5)
    m1 = H.Hang2020(23, 8, precision="bf16").to(_dev()).train()
    m2 = copy.deepcopy(m1)
    t1, t2 = FusedTrainer(m1, lr=1e-3), FusedTrainer(m2, lr=1e-3)
    for _ in range(2):
        l1, l2 = t1.train_step(x32, y), t2.train_step(xt, y)
    assert abs(float(l1) - float(l2)) <= 1e-5 * abs(float(l1))
    assert float((t1.logits - t2.logits).abs().max()) <= 1e-4 * float(t1.logits.abs().max())
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        if a.dtype.is_floating_point and not k.endswith("conv_layer.bias"):
            assert float((a.double() - b.double()).norm()) <= 1e-4 * max(float(a.double().norm()), 1e-12), k
    lg, lv = t2.forward_loss(xt, y)
    assert torch.isfinite(lg).all() and np.isfinite(float(lv))
