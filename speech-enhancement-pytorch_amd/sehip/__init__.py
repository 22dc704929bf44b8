"""sehip -- MI355X-native (gfx950) train-step path for ooshyun/Speech-Enhancement-Pytorch.

The package mirrors the reference's Python surface for the hot path (model registry, Solver train step,
loss, optimizer) and runs it on hand-written HIP kernels in libsehip.so through a C ABI (include/sehip.h).
There is no CPU or stock-PyTorch fallback: on a machine without the library or without a gfx950 GPU the
compute entry points raise SehipError.
"""
from ._lib import SehipError, lib, LIB_PATH  # noqa: F401

__version__ = "0.1.0"
