"""Shared helpers of the frame-level SAO tests: random per-CTU parameters and the ctypes calls."""
import ctypes as C

import numpy as np

import flatapi


def random_params(rng, n_ctus, chroma):
    arr = (flatapi.SaoParams * n_ctus)()
    for p in arr:
        p.type = int(rng.choice([0, 1, 2, 2]))          # none / band / edge
        p.eo_class = int(rng.integers(0, 4))
        p.band_position[0], p.band_position[1] = int(rng.integers(0, 29)), int(rng.integers(0, 29))
        for i in range(10):
            p.offsets[i] = int(rng.integers(-7, 8))
        p.offsets[0] = 0                                  # category 0 of the edge classes carries no offset (sao.h:51)
        if chroma:
            p.offsets[5] = 0
        p.bitdepth = 8
    return arr


def run_cpu(func, width, height, frame, luma, chroma):
    out = np.zeros_like(frame)
    func.restype = None
    func.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    func(width, height, frame.ctypes.data, out.ctypes.data, C.addressof(luma), C.addressof(chroma))
    return out
