"""dfmir_amd -- MI355X-native (gfx950) training hot path of heyblackC/DFMIR `--model registration`.

Python host code on PyTorch-ROCm over a C-ABI HIP library (libdfmir_hip.so, include/dfmir_hip.h).
See DESIGN.md for the path, boundary and kernels; INTEGRATION.md for how it drops into the
reference's train.py.
"""
from ._lib import DfmirHipError, LIB_PATH, lib  # noqa: F401

__all__ = ["DfmirHipError", "LIB_PATH", "lib"]
