"""ctypes loader for libkvz_hip.so."""
import ctypes
import os

from .build import LIB_PATH

_lib = None


def load_library():
    """Load the built HIP library.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); the hip strategy has no CPU fallback")
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # see kvz_runtime.hpp HwQueuesDefault: before the HIP runtime initialises (also when torch brought it in)
        _lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _lib.kvz_hip_version.restype = ctypes.c_char_p
    return _lib
