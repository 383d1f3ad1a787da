"""ORACLE (test infrastructure, never shipped as the product path): NumPy restatement of the reference's crop
preprocessing, the step that feeds the Hang2020 hot path.

Reference lines restated:
  src/utils.py:36-58   preprocess_image: drop the first and last 10 bands when there are more than 3, float32,
                       per-PIXEL min-max scaling over the bands (sklearn.preprocessing.minmax_scale(data, axis=1))
  src/utils.py:60-79   load_image: preprocess, then resize to (image_size, image_size) with NEAREST interpolation
  src/augmentation.py:13-14 + src/data.py:297-298   training: RandomHorizontalFlip(p=1) then RandomVerticalFlip(p=1)
  src/data.py:295-296  a missing year is an all-zero (bands, image_size, image_size) tensor

Third-party arithmetic restated here (both absent from /root/reference, installed in the build container only in part):
  * scikit-learn MinMaxScaler (sklearn/preprocessing/_data.py, `partial_fit` + `transform`, feature_range (0, 1)):
    data_min/max = nanmin/nanmax, range = max - min, ranges below 10 * eps(float32) count as constant (scale 1),
    scale = 1 / range, min_ = 0 - data_min * scale, X = X * scale + min_ as two separately rounded float32 operations.
    Pinned: tests/golden/preprocess.npz holds outputs of the reference's own preprocess_image (which calls sklearn).
  * torchvision.transforms.functional.resize(tensor, NEAREST) -> torch.nn.functional.interpolate(mode="nearest"):
    source index = min(floor(dst * float32(in / out)), in - 1).  torchvision is NOT installed here; the restatement is
    pinned against torch's own interpolate (tests/test_preprocess.py), which is the routine torchvision dispatches to.
"""
import numpy as np

F32_TINY_RANGE = np.float32(10.0) * np.finfo(np.float32).eps


def clip_bands(image, clip=10):
    """src/utils.py:40-42 (channel-first array)."""
    if image.shape[0] > 3:
        image = image[clip:, :, :]
        image = image[:-clip, :, :]
    return image


def minmax_over_bands(img):
    """Per-pixel min-max scaling over axis 0 of a float32 (C, H, W) array, rounding exactly as sklearn's float32 path."""
    img = np.asarray(img, dtype=np.float32)
    with np.errstate(all="ignore"):
        dmin = np.nanmin(img, axis=0)
        dmax = np.nanmax(img, axis=0)
    rng = (dmax - dmin).astype(np.float32)
    rng = np.where(rng < F32_TINY_RANGE, np.float32(1.0), rng).astype(np.float32)
    scale = (np.float32(1.0) / rng).astype(np.float32)
    min_ = (np.float32(0.0) - (dmin * scale).astype(np.float32)).astype(np.float32)
    out = (img * scale[None]).astype(np.float32)
    out = (out + min_[None]).astype(np.float32)
    return out


def preprocess_image(image, channel_is_first=True, clip=10):
    """src/utils.py:36-58."""
    if not channel_is_first:
        raise NotImplementedError("the hot path loads channel-first crops (load_image passes channel_is_first=True)")
    return minmax_over_bands(clip_bands(np.asarray(image), clip))


def nearest_index(out_size, in_size):
    """ATen nearest: src = min(floor(dst * float32(in / out)), in - 1)."""
    scale = np.float32(in_size) / np.float32(out_size)
    idx = np.floor(np.arange(out_size, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(idx, in_size - 1)


def resize_nearest(img, size):
    """src/utils.py:77 on a (C, H, W) array."""
    hi = nearest_index(size, img.shape[1])
    wi = nearest_index(size, img.shape[2])
    return img[:, hi][:, :, wi]


def load_crop(raw_chw, image_size, train=False, clip=10):
    """load_image (+ the training flips) for one raw channel-first crop; None / empty = missing year -> zeros."""
    if raw_chw is None or raw_chw.size == 0:
        raise ValueError("missing crops are zero-filled by the caller, which knows the band count")
    img = resize_nearest(preprocess_image(raw_chw, True, clip), image_size)
    if train:
        img = img[:, ::-1, ::-1]
    return np.ascontiguousarray(img, dtype=np.float32)
