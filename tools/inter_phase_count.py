#!/usr/bin/env python3
"""Developer tool: phases (workgroup barriers) per stage and per CTU of the inter CTU pass, counted by the host simulation built with -DKVZ_ICTU_COUNT_PHASES -- with one
wavefront per CTU the pass is a chain of dependent phases, so this is what its latency is made of.  usage: tools/inter_phase_count.py [case]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flatapi, ctu_common as cc, inter_common as ic
so = "/tmp/libkvz_hostsim_phases.so"
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DKVZ_ICTU_COUNT_PHASES", "-o", so, os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")])
sim = C.CDLL(so)
f = sim.kvz_hostsim_inter_frame
f.restype = None
f.argtypes = [C.c_int] * 4 + [C.c_uint64, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 5
name = sys.argv[1] if len(sys.argv) > 1 else "survey-416x240"
case = [c for c in ic.CASES if c[0] == name][0]
_, w, h, n, qp, preset, dbk, sao, owf, src = case
oracle = flatapi.load_oracle()
frames = ic.case_frames(case)
rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
mc = cc.model_constants()
fb = np.array(mc["entropy_fbits"], np.float32)
p = ic.PRESETS[preset]
names = ["merge MC+SATD", "early skip", "integer ME", "fractional ME", "candidates", "intra search", "intra recon", "inter quant/recon", "mock+rd cost", "copies", "io", "total (unattributed)"] + ["cat%d" % i for i in range(12, 16)] + ["EVENTS price()", "EVENTS cell_at", "EVENTS candidate gathers", "EVENTS mv_candidates calls", "EVENTS probe groups", "EVENTS staged windows", "EVENTS hor_pass_ref", "EVENTS predict_tile", "EVENTS   of those: integer vector", "EVENTS   one direction integer", "EVENTS   of two lists", "EVENTS tiles of two lists with one vector"] + ["cat%d" % i for i in range(28, 31)] + ["outside"]
tot = np.zeros(32, np.int64)
ctus = ((w + 63) // 64) * ((h + 63) // 64)
for k in range(1, n):
    rec = np.zeros(w * h * 3 // 2, np.uint8)
    out = np.zeros((h // 4, w // 4), ic.CU_DTYPE)
    f(w, h, int(qps[k]), k, int(mc["coeff_weights"][str(int(qps[k]))]), fb.ctypes.data, int(owf > 0), sao, dbk, p["fme_level"], p["pu_depth_inter_max"], 0, p["fast_residual_cost"],
      np.ascontiguousarray(frames[k]).ctypes.data, np.ascontiguousarray(rf[k - 1]).ctypes.data, np.ascontiguousarray(cu[k - 1]).ctypes.data, rec.ctypes.data, out.ctypes.data)
    assert np.array_equal(rec, rs[k])
    ph = (C.c_long * 32)()
    sim.kvz_hostsim_inter_phases(ph)
    tot += np.array(ph[:], np.int64)
per = tot / float(ctus * (n - 1))
ev = per[16:28].copy(); per[16:28] = 0
print("%s: phases per CTU (%d pictures x %d CTUs)" % (name, n - 1, ctus))
for i in np.argsort(-per):
    if per[i] > 0:
        print("  %-24s %9.1f  %5.1f %%" % (names[i], per[i], 100 * per[i] / per.sum()))
print("  %-24s %9.1f" % ("sum", per.sum()))
for i in range(12):
    print("  %-28s %9.1f per CTU" % (names[16 + i], ev[i]))
b = cu[1:]
print("  CU records (4x4 units) per picture: intra %d, skipped %d, merged %d, amvp %d" % tuple(int(v) // (n - 1) for v in ((b["type"] == 1).sum(), ((b["type"] == 2) & (b["skipped"] == 1)).sum(), ((b["type"] == 2) & (b["merged"] == 1)).sum(), ((b["type"] == 2) & (b["merged"] == 0) & (b["skipped"] == 0)).sum())))
