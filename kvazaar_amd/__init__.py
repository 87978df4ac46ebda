"""kvazaar_amd -- MI355X-native `hip` strategy for kvazaar's per-CTU hot path.

The product is the C-ABI shared library kvazaar_amd/lib/libkvz_hip.so (include/kvz_hip.h, include/kvz_hip_batch.h),
built from the HIP sources in kvazaar_amd/csrc.  This Python package builds and loads it and holds the ctypes views of its
three API groups -- capi (flat per-call strategy API), dev (device-resident primitives), batch (the batched CTU pass) -- plus the
workload generator (synth) and the rank / tile sharding helpers (sharding) that bench.py, __graft_entry__ and the tests use;
kvazaar itself binds the library from C.

There is no CPU path here: loading fails loudly when the library has not been built, and the library aborts
when no gfx950 device is usable.
"""
from .build import LIB_PATH, build_library, build_tools  # noqa: F401
from .library import load_library  # noqa: F401

__version__ = "0.1"
