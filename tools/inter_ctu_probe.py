#!/usr/bin/env python3
"""Development probe: the inter CTU pass on the pictures of a tests/inter_common.py case, fed from the oracle's references, timed and compared.
usage: tools/inter_ctu_probe.py <case name> [copies]   (copies: the same picture as that many independent sequences in one launch)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flatapi, inter_common as ic
import kvazaar_amd
from kvazaar_amd.dev import Dev
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_inter_ctu import InterParams, params_of

name = sys.argv[1]
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 1
oracle = flatapi.load_oracle()
lib = kvazaar_amd.load_library()
dev = Dev(lib)
case = [c for c in ic.CASES if c[0] == name][0]
_, w, h, n, qp, preset, dbk, sao, owf, src = case
frames = ic.case_frames(case)
t0 = time.time()
rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
print("oracle %.1fs" % (time.time() - t0), flush=True)
lib.kvz_hip_dev_inter_ctu_pass.restype = C.c_int
lib.kvz_hip_dev_inter_ctu_pass.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]
fs, cells = w * h * 3 // 2, (w // 4) * (h // 4)
ctus = ((w + 63) // 64) * ((h + 63) // 64)
for k in range(1, n):
    prm = params_of(case, qps[k], k)
    d_src, d_ref, d_rcu = dev.put(np.tile(frames[k], copies)), dev.put(np.tile(rf[k - 1], copies)), dev.put(np.tile(cu[k - 1].reshape(-1), copies))
    d_rec, d_cu = dev.empty(copies * fs), dev.empty(copies * cells * ic.CU_DTYPE.itemsize)
    best = 1e9
    for rep in range(3):
        t0 = time.time()
        rc = lib.kvz_hip_dev_inter_ctu_pass(d_src, d_ref, d_rcu, d_rec, d_cu, None, w, h, copies, C.addressof(prm))
        best = min(best, time.time() - t0)
    rec = dev.get(d_rec, (copies, fs), np.uint8)
    got = dev.get(d_cu, (copies, h // 4, w // 4), ic.CU_DTYPE)
    same = all(ic.first_difference(got[i][None], cu[k][None]) is None and np.array_equal(rec[i], rs[k]) for i in range(copies))
    print("picture %d rc %d: %.1f ms for %d CTUs = %.0f CTUs/s, equal to the oracle: %s" % (k, rc, best * 1e3, ctus * copies, ctus * copies / best, same), flush=True)
    dev.free(d_src, d_ref, d_rcu, d_rec, d_cu)
