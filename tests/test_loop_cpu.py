"""Host logic of the Trainer-shaped loop (deeptreeattention_amd/loop.py): the plateau rule against torch's own
ReduceLROnPlateau with the reference's settings (src/main.py:137-147), and the synthetic dataset's batch structure
against what the reference's TreeDataset + default collate hand to training_step (src/data.py:284-310,
src/main.py:71-76)."""
import numpy as np
import torch

from deeptreeattention_amd.loop import PlateauScheduler, SyntheticTreeDataset


class _Lr:
    def __init__(self, lr):
        self.lr = lr


def _trace(seed, n=120):
    rng = np.random.RandomState(seed)
    v, out = 2.0, []
    for i in range(n):
        # descending phases, flat noisy plateaus and the odd spike
        drift = -0.03 if (i // 15) % 2 == 0 else 0.0
        v = max(0.05, v * (1 + drift) + rng.normal(0, 2e-4 if drift == 0 else 5e-3))
        out.append(v + (0.5 if rng.rand() < 0.03 else 0.0))
    return out


def test_plateau_scheduler_matches_torch():
    for seed in range(6):
        w = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([w], lr=1e-3)
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.75, patience=8, threshold=1e-4,
                                                         threshold_mode="rel", cooldown=0, min_lr=1e-7, eps=1e-8)
        tr = _Lr(1e-3)
        mine = PlateauScheduler(tr)
        for v in _trace(seed):
            ref.step(v)
            mine.step(v)
            assert tr.lr == opt.param_groups[0]["lr"], seed
        assert tr.lr < 1e-3          # the traces do contain plateaus


def test_plateau_scheduler_min_lr_and_eps():
    tr = _Lr(2e-7)
    s = PlateauScheduler(tr, patience=0)
    s.step(1.0)
    s.step(1.0)                      # bad epoch -> 1.5e-7
    assert abs(tr.lr - 1.5e-7) < 1e-20
    s.step(1.0)                      # -> 1.125e-7
    s.step(1.0)                      # would be 8.4e-8 < min_lr -> clamps to 1e-7 (difference 1.25e-8 > eps)
    assert tr.lr == 1e-7
    s.step(1.0)
    assert tr.lr == 1e-7             # old - new = 0 <= eps: unchanged


def test_synthetic_tree_dataset_batch_structure():
    ds = SyntheticTreeDataset(10, bands=5, classes=3, size=11, device="cpu", seed=1)
    ind, inputs, y = ds[3]
    assert isinstance(ind, str) and inputs["HSI"].shape == (5, 11, 11) and y.dtype == torch.int64
    batches = list(ds.loader(4))
    assert [b[2].shape[0] for b in batches] == [4, 4, 2]
    ind, inputs, y = batches[0]
    assert len(ind) == 4 and inputs["HSI"].shape == (4, 5, 11, 11) and inputs["HSI"].dtype == torch.float32
    assert float(inputs["HSI"].min()) >= 0 and float(inputs["HSI"].max()) < 1
    ens = SyntheticTreeDataset(8, bands=4, classes=3, years=3, sites=5, missing=0.5, device="cpu", seed=2)
    ind, inputs, y = next(ens.loader(8))
    assert isinstance(inputs["HSI"], list) and len(inputs["HSI"]) == 3 and inputs["HSI"][1].shape == (8, 4, 11, 11)
    assert inputs["site"].dtype == torch.int64 and int(inputs["site"].max()) < 5
    zeroed = sum(int((t.flatten(1).abs().sum(1) == 0).sum()) for t in inputs["HSI"][1:])
    assert zeroed > 0                # missing years are zero-filled patches, as TreeDataset produces them
