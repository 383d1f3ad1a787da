"""deeptreeattention_amd: MI355X-native (gfx950) implementation of DeepTreeAttention's Hang2020 hot path.

The package mirrors the reference's module names for this path only:
    deeptreeattention_amd.Hang2020   <->  src/models/Hang2020.py
    deeptreeattention_amd.year       <->  src/models/year.py        (learned_ensemble)
    deeptreeattention_amd.engine     fused train step (forward + weighted CE + backward + Adam, optional RCCL DDP)
All arithmetic runs in libdta_hip.so (HIP, C ABI in include/dta_hip.h); there is no CPU fallback.
"""
from . import Hang2020  # noqa: F401
from .Hang2020 import set_default_precision, get_default_precision  # noqa: F401

__all__ = ["Hang2020", "set_default_precision", "get_default_precision"]
