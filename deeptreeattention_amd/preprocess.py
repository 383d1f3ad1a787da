"""Device-side crop preprocessing: the host mirror of the reference's `src/utils.py` loader functions for the step
that feeds the Hang2020 hot path.

The reference preprocesses one crop at a time in DataLoader workers (`load_image` -> `preprocess_image` -> torchvision
resize -> flips, src/utils.py:36-79, src/data.py:284-310).  Here a whole batch of ragged raw crops goes to the GPU as
ONE buffer and ONE launch (`dta_preprocess_crops`) writes the (B, bands, size, size) float32 batch the networks take;
the arithmetic is bit-identical to the reference's (tests/test_preprocess.py).  File decoding (rasterio / np.load) stays
with the caller: these functions start from the arrays those readers return, or from the raw strip bytes of the
pixel-interleaved TIFF crops (`pixel_interleaved=True`)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

_DTYPES = {np.dtype(np.float32): _lib.CROP_F32, np.dtype(np.int16): _lib.CROP_I16, np.dtype(np.uint8): _lib.CROP_U8}


def out_bands(bands_raw, clip=10):
    return bands_raw - 2 * clip if bands_raw > 3 else bands_raw


class PatchTiles:
    """A preprocessed batch stored as the first conv's bf16 tiles ([B][ceil(bands / 16)][H * W][16], halo-free) instead
    of the float32 (B, bands, H, W) tensor: what `preprocess_batch(..., tiles=True)` returns and what
    `engine.FusedTrainer.train_step / forward_loss` accept in bf16 mode.  `.float()` expands it back (bf16-rounded
    values) for anything that wants the NCHW tensor."""

    def __init__(self, tiles, batch, bands, height, width):
        self.tiles, self.shape = tiles, (int(batch), int(bands), int(height), int(width))

    def float(self):
        B, C, Hh, Ww = self.shape
        t = self.tiles.view(B, -1, Hh * Ww, 16).view(torch.bfloat16).float()       # [B][chunk][pixel][16]
        return t.permute(0, 1, 3, 2).reshape(B, -1, Hh, Ww)[:, :C].contiguous()


def preprocess_batch(crops, image_size, train=False, pixel_interleaved=False, clip=10, device="cuda", out=None,
                     tiles=False):
    """crops: list of raw arrays (numpy or torch; all one dtype among float32 / int16 / uint8), each band-first
    (bands, h, w) as rasterio's read() / np.load return it, or (h, w, bands) when pixel_interleaved (the on-disk order); `None` marks a missing year (all-zero output, reference data.py:295-296).
    Returns a (len(crops), bands_out, image_size, image_size) float32 device tensor: the reference's
    load_image(path, image_size) for every crop, followed by the training flips when train=True.
    tiles=True: returns a PatchTiles instead -- the same values rounded to bf16 and laid out as the first conv's tiles,
    written by the same single launch (the float32 batch is never materialised; bf16-mode networks only)."""
    L = _lib.lib()
    channel_is_first = not pixel_interleaved
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("deeptreeattention_amd.preprocess runs on a ROCm device only (no CPU fallback)")
    arrs, hs, ws, offs, pos = [], [], [], [], 0
    bands, dt = None, None
    for c in crops:
        if c is None:
            hs.append(0); ws.append(0); offs.append(pos)
            continue
        a = c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else np.asarray(c)
        if a.ndim != 3:
            raise ValueError("each crop must be a 3-d array")
        if a.dtype not in _DTYPES:
            a = a.astype(np.float32)          # what the reference does with any dtype: np.asarray(image, dtype='float32')
        b, h, w = (a.shape if channel_is_first else (a.shape[2], a.shape[0], a.shape[1]))
        if bands is None:
            bands, dt = b, a.dtype
        if b != bands or a.dtype != dt:
            raise ValueError("all crops of a batch must share the band count and the dtype")
        hs.append(h); ws.append(w); offs.append(pos)
        arrs.append(np.ascontiguousarray(a).reshape(-1))
        pos += a.size
    if bands is None:
        raise ValueError("at least one crop of the batch must be present")
    host = torch.from_numpy(np.concatenate(arrs))
    raw = host.to(dev, non_blocking=True)
    meta = torch.tensor([offs], dtype=torch.int64).reshape(-1).to(dev)
    hw = torch.tensor([hs, ws], dtype=torch.int32).to(dev)
    B, Cout = len(crops), out_bands(bands, clip)
    if tiles:
        desc = _lib.CropDesc(B, bands, clip, image_size, 1 if train else 0,
                             _lib.CROP_CHW if channel_is_first else _lib.CROP_HWC, _DTYPES[np.dtype(dt)])
        t = torch.empty(B * ((Cout + 15) // 16) * image_size * image_size * 16, dtype=torch.int16, device=dev)
        _lib.check(L.dta_preprocess_crops_tiles(C.byref(desc), _lib.ptr(raw), _lib.ptr(meta), _lib.ptr(hw[0]),
                                                _lib.ptr(hw[1]), _lib.ptr(t), _lib.current_stream_ptr()),
                   "dta_preprocess_crops_tiles")
        return PatchTiles(t, B, Cout, image_size, image_size)
    if out is None:
        out = torch.empty(B, Cout, image_size, image_size, dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (B, Cout, image_size, image_size) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 tensor of shape {}".format((B, Cout, image_size, image_size)))
    desc = _lib.CropDesc(B, bands, clip, image_size, 1 if train else 0,
                         _lib.CROP_CHW if channel_is_first else _lib.CROP_HWC, _DTYPES[np.dtype(dt)])
    _lib.check(L.dta_preprocess_crops(C.byref(desc), _lib.ptr(raw), _lib.ptr(meta), _lib.ptr(hw[0]), _lib.ptr(hw[1]),
                                      _lib.ptr(out), _lib.current_stream_ptr()), "dta_preprocess_crops")
    return out


def preprocess_image(image, channel_is_first=False, device="cuda"):
    """Reference src/utils.py:36-58 for one array at its own resolution: bands are axis 0 (dropped 10 + 10 when there
    are more than 3), float32, min-max over axis 0 per position; with channel_is_first=False the reference then rolls
    the LAST axis to the front (`np.rollaxis(img, 2, 0)`, :53-54) and so does this.  Returns a float32 device tensor."""
    a = image.detach().cpu().numpy() if isinstance(image, torch.Tensor) else np.asarray(image)
    if a.dtype not in _DTYPES:
        a = a.astype(np.float32)                      # np.asarray(image, dtype='float32') of the reference
    h, w = a.shape[1], a.shape[2]
    side = max(h, w)
    if h != w:   # the batch kernel produces squares: pad (min-max is per pixel, NEAREST with in == out is the identity)
        pad = np.zeros((a.shape[0], side, side), dtype=a.dtype)
        pad[:, :h, :w] = a
        a = pad
    out = preprocess_batch([a], side, False, False, device=device)[0][:, :h, :w]
    return out if channel_is_first else out.permute(2, 0, 1)


def load_image(image, image_size, device="cuda"):
    """Reference src/utils.py:60-79 from the array the file reader returned (band-first, as rasterio's read() and the
    .npy crops are): preprocess + NEAREST resize to (image_size, image_size)."""
    return preprocess_batch([image], image_size, False, False, device=device)[0]
