"""The bf16 spectral network at crop sizes other than the benchmark's 11x11 / 24x24, against the NumPy oracle.

The reference crops whatever size the config asks for (src/data.py image_size; the spectral branch, Hang2020.py:136-168,
pools each stage to a vector and is size-agnostic -- the year models of BASELINE configs[4] run it at 24x24), and the conv
kernels' geometry changes with it: how many patches share a workgroup's tile rows (8x8: eight, 13x13: three, 20x20: one
and a split remainder), how the pixels are dealt to the MFMA tile rows (conv_row_tables: by haloed-row residue, or pixel
order when a residue class overflows), ragged last workgroups, maps split over several workgroups, odd sizes whose 2x2
pools drop a row and a column.  All three heads are driven (every stage's feature path has a gradient); comparison with
the oracle run with the same bf16 operand / storage rounding: head scores to 1e-3, the whole gradient vector and every
large tensor's norm to 1e-2 (north_star's bf16 budget)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import hang2020_np as O
from oracle import prng

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,bands,B", [(8, 20, 37), (9, 33, 29), (13, 24, 22), (16, 17, 13), (20, 24, 9), (11, 40, 3), (24, 18, 5)])
def test_bf16_spectral_network_at_other_crop_sizes_vs_oracle(size, bands, B):
    from deeptreeattention_amd import Hang2020 as H
    classes = 7
    p = O.init_params(O.subnet_spec("spectral", bands, classes), seed=size)
    m = H.spectral_network(bands, classes, precision="bf16")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in p.items()})
    m = m.cuda().train()
    x = prng.uniform01(100 + size, 1, (B, bands, size, size))
    ds = [prng.uniform(100 + size, 10 + i, (B, classes), -1, 1) for i in range(3)]
    s = m(torch.from_numpy(x).cuda())
    sum((a * torch.from_numpy(b).cuda()).sum() for a, b in zip(s, ds)).backward()
    got = {k: q.grad.detach().cpu().numpy() for k, q in m.named_parameters() if q.grad is not None}
    O.bf16_mode(True)
    try:
        rs, cache, upd = O.subnet_fwd(p, "", "spectral", x, True, np.float64)
        q_g = O.subnet_bwd(p, "", cache, [d.astype(np.float64) for d in ds], np.float64)
    finally:
        O.bf16_mode(False)
    for i in range(3):
        assert rel_l2(s[i].detach().cpu().numpy(), rs[i]) < 1e-3, i
    num = den = 0.0
    for k, v in q_g.items():
        if k.endswith("conv_layer.bias") or not np.any(v):
            continue
        g, v = np.asarray(got[k], np.float64), np.asarray(v, np.float64)
        num += float(((g - v) ** 2).sum()); den += float((v ** 2).sum())
        if v.size >= 1000:
            assert abs(np.linalg.norm(g) - np.linalg.norm(v)) <= 1e-2 * np.linalg.norm(v), k
    assert np.sqrt(num / den) < 1e-2, np.sqrt(num / den)
    sd = m.state_dict()
    for k, v in upd.items():
        assert rel_l2(sd[k].cpu().numpy(), v) < 1e-3, k
