"""tagrec_amd -- MI355X-native hot path for tag-aware graph recommenders.

Python host code (this package) mirrors the reference's operator / model / step
surface and calls a C-ABI HIP library
(`csrc/` -> `libtagrec_hip.so`, declared in `include/tagrec.h`) through ctypes.
There is no CPU fallback: every op raises if the library or a GPU is missing.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
