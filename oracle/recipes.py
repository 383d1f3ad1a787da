"""Input recipes shared by tests/golden/make_golden.py (which runs the reference on them) and the tests (which run the
oracle and the HIP path on them).  TEST INFRASTRUCTURE like the rest of oracle/: never imported by the product path."""
import numpy as np

from . import prng

# tests/golden/multistage_steps.npz: the reference's MultiStage loop over two levels (multi_stage.py:41-66, :258-288)
MULTISTAGE = dict(years=3, bands=16, classes=(3, 5), B=6, lrs=(1e-3, 2e-3), steps=3)


def multistage_inputs(step, level, years, B, bands, classes):
    """Level 1 lacks year 2 on step 1, level 0 lacks year 0 on step 2 (all-zero tensors, reference year.py:27)."""
    imgs = [prng.uniform01(300 + 10 * level + step, yy, (B, bands, 11, 11)) for yy in range(years)]
    if step == 1 and level == 1:
        imgs[2] = np.zeros_like(imgs[2])
    if step == 2 and level == 0:
        imgs[0] = np.zeros_like(imgs[0])
    return imgs, prng.randint(300 + 10 * level + step, 7, (B,), classes)


def multistage_weight(classes):
    return (0.1 + (np.arange(classes) % 7)).astype(np.float32)
