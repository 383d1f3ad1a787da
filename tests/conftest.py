import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "devlib: compares an alternative launch plan (a developer switch) with the default one: "
                            "runs only in the developer library, libdta_hip_dev.so (DTA_DEV_LIB=1)")


def rel_l2(a, b):
    """norm-wise relative error |a-b|_2 / |b|_2 (compare norm-wise, not element-wise)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt((b ** 2).sum())
    num = np.sqrt(((a - b) ** 2).sum())
    return num / den if den > 0 else num


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return _load


class Bf16Yardstick:
    """The reference's OWN bf16 deviations (tests/golden/bf16_autocast.npz: the reference's modules under
    torch.autocast("cpu", torch.bfloat16) next to themselves in fp32, made by make_golden.case_bf16_autocast on the parity
    case's own batch -- `<quantity>` -- and on six more batches of the same shape -- `<quantity>_max` is the largest of
    the seven).  A bf16 implementation is held, per quantity, to max(1e-2, 1.5 x what the reference itself moves under
    bf16): north_star's 1e-2 budget wherever the reference's bf16 run keeps it, the reference's own figure (x 1.5) where
    it does not."""

    def __init__(self, g):
        self.table = dict(zip(g["names"].tolist(), g["values"].tolist()))
        self.files = self.table        # (membership tests read like an npz's)

    def ref(self, key):
        """The reference's bf16 deviation on the parity case's own batch."""
        return float(self.table[key])

    def bound(self, key, factor=1.5):
        return max(1e-2, factor * float(self.table[key + "_max"]))

    def check_gradients(self, tag, got, exact, prefix="", min_size=1000, verbose=True, norms=True):
        """got / exact: {name: ndarray} (exact = the fp64 oracle, pinned to the reference's fp32 golden).  Asserts, against
        the reference-autocast yardstick of case `tag`: every tensor of >= min_size elements keeps its norm and its
        element-wise distance within bound(); every SMALLER tensor its norm within bound() or an absolute deviation of
        at most 1e-3 of the total gradient norm; so do the whole gradient vector and the total norm (all tensors).
        norms=False (batches of <= 16 patches): a tensor's norm deviation is the projection of its rounding noise on the
        gradient itself -- one draw of a random sign and size, anywhere between 0 and the element-wise distance -- and with a
        handful of patches the reference's single draw is no yardstick for another implementation's; there only the
        element-wise distance is held per tensor (the norms still through the total)."""
        num = den = t_got = t_ex = 0.0
        fails, small = [], []
        for k, v in exact.items():
            key = prefix + k
            if f"{tag}gnorm_dev/{key}" not in self.table:
                continue                      # conv biases under batch-statistics BatchNorm / all-zero tensors
            a = np.asarray(got[k], np.float64); b = np.asarray(v, np.float64)
            na, nb = np.linalg.norm(a), np.linalg.norm(b)
            num += float(((a - b) ** 2).sum()); den += float((b ** 2).sum()); t_got += na ** 2; t_ex += nb ** 2
            if b.size < min_size:
                small.append((key, a, b, na, nb))      # judged below, once the total gradient norm is known
                continue
            dn, de = abs(na - nb) / nb, np.linalg.norm(a - b) / nb
            bn, be = self.bound(f"{tag}gnorm_dev/{key}"), self.bound(f"{tag}gelem_dev/{key}")
            if verbose:
                print(f"  {tag}{key:62s} {b.size:7d} norm dev {dn:.2e} (reference bf16 {self.ref(tag + 'gnorm_dev/' + key):.2e}) "
                      f"elem {de:.2e} (reference bf16 {self.ref(tag + 'gelem_dev/' + key):.2e})")
            if (norms and dn > bn) or de > be:
                fails.append((key, dn, bn, de, be))
        # Tensors under min_size elements (BatchNorm gamma / beta, biases, the spatial-attention stencils and channel pools:
        # two thirds of the gradient tensors): each keeps its norm within bound() -- the reference's own bf16 figure x 1.5 --
        # OR moves by no more than 1e-3 of the TOTAL gradient norm in absolute terms (cancellation-dominated scalars whose own
        # norm is rounding-sized: a relative figure on them measures noise against noise).  Which branch held is printed.
        gtot = np.sqrt(t_ex)
        held = {"norm": 0, "absolute": 0}
        for key, a, b, na, nb in small:
            dn = abs(na - nb) / nb if nb > 0 else (0.0 if na == 0 else np.inf)
            bn = self.bound(f"{tag}gnorm_dev/{key}")
            dabs = np.linalg.norm(a - b) / gtot
            branch = "norm" if dn <= bn else ("absolute" if dabs <= 1e-3 else None)
            if verbose:
                print(f"  {tag}{key:62s} {b.size:7d} norm dev {dn:.2e} (bound {bn:.2e}, reference bf16 {self.ref(tag + 'gnorm_dev/' + key):.2e}) "
                      f"|diff| / |total gradient| {dabs:.2e} -> {branch or 'FAIL'}")
            if branch is None:
                fails.append((key, dn, bn, dabs, 1e-3))
            else:
                held[branch] += 1
        if small:
            print(f"  {tag} {len(small)} tensors under {min_size} elements: {held['norm']} held by their norm bound, {held['absolute']} by the absolute one")
        whole = np.sqrt(num / den)
        tot = abs(np.sqrt(t_got) - np.sqrt(t_ex)) / np.sqrt(t_ex)
        print(f"  {tag} whole gradient vector: elem {whole:.2e} (reference bf16 {self.ref(tag + 'whole_elem_dev'):.2e}), "
              f"total norm dev {tot:.2e} (reference bf16 {self.ref(tag + 'total_norm_dev'):.2e})")
        assert not fails, fails
        assert whole <= self.bound(f"{tag}whole_elem_dev"), whole
        assert tot <= self.bound(f"{tag}total_norm_dev"), tot
        return whole, tot


@pytest.fixture(scope="session")
def bf16_yardstick(golden):
    return Bf16Yardstick(golden("bf16_autocast.npz"))


@pytest.fixture
def devlib():
    """Tests that flip a developer switch (alternative launch plans) need the DEVELOPER library: the product library reads
    nothing from the environment.  In a default run they are skipped here and executed by
    tests/test_kernel_variants_gpu.py::test_developer_switch_variants_in_the_developer_library, which re-runs them in a
    process that loads libdta_hip_dev.so (DTA_DEV_LIB=1)."""
    from deeptreeattention_amd import _lib
    L = _lib.lib()
    if not L.dta_dev_switches_enabled():
        pytest.skip("developer-switch variant: runs in libdta_hip_dev.so (test_developer_switch_variants_in_the_developer_library)")
    return L
