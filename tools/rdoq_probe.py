#!/usr/bin/env python3
"""Developer tool: ONE transform block through kvz_hip_rdoq_blocks (one wavefront: rdoq_block_wave), for `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS`:
the instruction count of the routine on a block of the workload's statistics.  usage: tools/rdoq_probe.py <width> [type 0|2] [scale]   (tools/rdoq_insts.sh runs it under the counters)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kvazaar_amd, flatapi

def main():
    lib = kvazaar_amd.load_library()
    rng = np.random.default_rng(3)
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    typ = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    qp = 22
    scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5 * 2 ** (9 - int(np.log2(w)))  # 1.5 x the magnitude that quantises to a level at QP 22
    # the transform of a noisy residual: Laplacian coefficients with a mild decay towards the high frequencies -- at QP 22 about a third of them can code a level
    decay = np.exp(-np.add.outer(np.arange(w), np.arange(w)) / (1.5 * w))
    coef = np.ascontiguousarray(np.clip(rng.laplace(0, scale, (w, w)) * decay, -32000, 32000).astype(np.int16).reshape(-1))
    dest = np.zeros(w * w, np.int16)
    ctx = np.ascontiguousarray(rng.integers(2, 120, 160).astype(np.uint8))
    lib.kvz_hip_rdoq_blocks.restype = None
    lib.kvz_hip_rdoq_blocks.argtypes = [C.c_int, C.c_double, flatapi.u8p, flatapi.i16p, flatapi.i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lam = 0.57 * 2 ** ((qp - 12) / 3.0)
    for _ in range(3):
        lib.kvz_hip_rdoq_blocks(qp, lam, flatapi.ptr(ctx), flatapi.ptr(coef), flatapi.ptr(dest), w, typ, 0, 0, 1)
    print("width", w, "type", typ, "nonzero levels:", int(np.count_nonzero(dest)), "of", w * w)

if __name__ == "__main__":
    main()
