"""Developer tool: throughput of the device-side crop preprocessing (dta_preprocess_crops) on a batch already in HBM."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeptreeattention_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 11
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
hs = rng.randint(8, 33, size=B).astype(np.int32)
ws = rng.randint(8, 33, size=B).astype(np.int32)
bands = 389
sizes = hs.astype(np.int64) * ws * bands
offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
L = _lib.lib()
for name, tdt, code in (("int16", torch.int16, _lib.CROP_I16), ("float32", torch.float32, _lib.CROP_F32)):
    raw = (torch.rand(int(sizes.sum()), device=dev) * 9000).to(tdt)
    d_off, d_h, d_w = (torch.from_numpy(a).to(dev) for a in (offs, hs, ws))
    out = torch.empty(B, bands - 20, S, S, device=dev)
    for lname, layout in (("band-first", _lib.CROP_CHW), ("pixel-interleaved", _lib.CROP_HWC)):
        desc = _lib.CropDesc(B, bands, 10, S, 1, layout, code)
        def run():
            _lib.check(L.dta_preprocess_crops(C.byref(desc), _lib.ptr(raw), _lib.ptr(d_off), _lib.ptr(d_h), _lib.ptr(d_w),
                                              _lib.ptr(out), _lib.current_stream_ptr()), "dta_preprocess_crops")
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        samp = B * S * S * (bands - 20) * raw.element_size()      # sampled input elements (read twice, second from L2)
        outb = out.numel() * 4
        print(f"B={B} {S}x{S} {name} {lname}: {ms * 1e3:.1f} us  {B / ms * 1e3:,.0f} crops/s  "
              f"algorithmic {(samp + outb) / ms / 1e6:.0f} GB/s (raw batch {raw.numel() * raw.element_size() / 1e6:.0f} MB, out {outb / 1e6:.0f} MB)")
