"""kvazaar_amd -- MI355X-native `hip` strategy for kvazaar's per-CTU hot path.

The product is the C-ABI shared library kvazaar_amd/lib/libkvz_hip.so (include/kvz_hip.h, include/kvz_hip_batch.h),
built from the HIP sources in kvazaar_amd/csrc.  This Python package only builds and loads it (ctypes) for the
tests, bench.py and __graft_entry__; kvazaar itself binds the library from C.

There is no CPU path here: loading fails loudly when the library has not been built, and the library aborts
when no gfx950 device is usable.
"""
from .build import LIB_PATH, build_library  # noqa: F401
from .library import load_library  # noqa: F401

__version__ = "0.1"
