#!/usr/bin/env python3
"""Developer tool: ONE 16x16 luma block through kvz_hip_rdoq_blocks (a single active lane), for `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU`:
instructions per coefficient of kvz_rdoq on the device = counter / 256."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kvazaar_amd, flatapi

def main():
    lib = kvazaar_amd.load_library()
    rng = np.random.default_rng(3)
    w, qp = 16, 27
    coef = np.ascontiguousarray(rng.integers(-400, 401, w * w).astype(np.int16))  # dense: every coefficient is above the threshold of QP 27
    dest = np.zeros(w * w, np.int16)
    ctx = np.ascontiguousarray(rng.integers(2, 120, 160).astype(np.uint8))
    lib.kvz_hip_rdoq_blocks.restype = None
    lib.kvz_hip_rdoq_blocks.argtypes = [C.c_int, C.c_double, flatapi.u8p, flatapi.i16p, flatapi.i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lam = 0.57 * 2 ** ((qp - 12) / 3.0)
    for _ in range(3):
        lib.kvz_hip_rdoq_blocks(qp, lam, flatapi.ptr(ctx), flatapi.ptr(coef), flatapi.ptr(dest), w, 0, 0, 0, 1)
    print("nonzero levels:", int(np.count_nonzero(dest)), "of", w * w)

if __name__ == "__main__":
    main()
