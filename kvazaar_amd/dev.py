"""ctypes view of include/kvz_hip_dev.h (device-resident batch entry points) for the GPU tests and bench_kernels.py."""
import ctypes as C

import numpy as np

TRANSFORM_KINDS = {"dct4": 0, "dct8": 1, "dct16": 2, "dct32": 3, "dst4": 4, "idct4": 5, "idct8": 6, "idct16": 7, "idct32": 8, "idst4": 9}
TRANSFORM_SIZE = {0: 4, 1: 8, 2: 16, 3: 32, 4: 4, 5: 4, 6: 8, 7: 16, 8: 32, 9: 4}


class Dev:
    def __init__(self, cdll):
        self.lib = l = cdll
        vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
        l.kvz_hip_dev_alloc.restype = vp; l.kvz_hip_dev_alloc.argtypes = [sz]
        l.kvz_hip_dev_free.restype = None; l.kvz_hip_dev_free.argtypes = [vp]
        l.kvz_hip_dev_upload.restype = None; l.kvz_hip_dev_upload.argtypes = [vp, vp, sz]
        l.kvz_hip_dev_download.restype = None; l.kvz_hip_dev_download.argtypes = [vp, vp, sz]
        l.kvz_hip_dev_sync.restype = None; l.kvz_hip_dev_sync.argtypes = []
        l.kvz_hip_dev_timer_start.restype = None; l.kvz_hip_dev_timer_start.argtypes = []
        l.kvz_hip_dev_timer_stop.restype = C.c_float; l.kvz_hip_dev_timer_stop.argtypes = []
        for f in (l.kvz_hip_dev_sad_nxn, l.kvz_hip_dev_satd_nxn):
            f.restype = None; f.argtypes = [ci, vp, vp, ci, vp]
        l.kvz_hip_dev_transform.restype = None; l.kvz_hip_dev_transform.argtypes = [ci, vp, vp, vp, ci, ci]
        l.kvz_hip_dev_angular_pred.restype = None; l.kvz_hip_dev_angular_pred.argtypes = [ci, ci, vp, vp, ci, vp]

    def put(self, a):
        a = np.ascontiguousarray(a)
        p = self.lib.kvz_hip_dev_alloc(a.nbytes)
        self.lib.kvz_hip_dev_upload(p, a.ctypes.data, a.nbytes)
        return p

    def copy_in(self, p, a):
        a = np.ascontiguousarray(a)
        self.lib.kvz_hip_dev_upload(p, a.ctypes.data, a.nbytes)

    def empty(self, nbytes):
        return self.lib.kvz_hip_dev_alloc(nbytes)

    def get(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self.lib.kvz_hip_dev_download(out.ctypes.data, p, out.nbytes)
        return out

    def free(self, *ps):
        for p in ps:
            self.lib.kvz_hip_dev_free(p)
