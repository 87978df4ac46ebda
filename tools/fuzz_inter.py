#!/usr/bin/env python3
"""Fuzz of the inter CTU pass's device sources (host simulation, tests/hostsim) against the sequence oracle: random small clips (pan speed, noise, moving objects),
random picture sizes that cut CTUs, --qp 10..44 (picture QPs on both sides of fast-residual-cost), the four presets of the low-delay configuration (ultrafast /
superfast / veryfast / faster: subme 0 / 2 / 2 / 4, PUs down to 16x16 / 16x16 / 8x8 / 8x8, fast-residual-cost 28 / 28 / 28 / 0), low-delay GOPs of 2, 3, 4 and 8 pictures,
slow and fast pans, loop filters and the overlapped-picture motion restriction on or off, --no-wpp.  Every B picture is searched by
the simulated device program from the oracle's reference picture and CU records; reconstruction and every CU decision must be the oracle's -- and, for the lp-g4d3t1
rounds with WPP, the device's entropy coder for B pictures must write the oracle coder's slice data from the oracle's records.
usage: tools/fuzz_inter.py [rounds] [seed]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flatapi, ctu_common as cc, inter_common as ic, entropy_common as ec

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = flatapi.load_oracle()
sim = C.CDLL(os.path.join(ROOT, "tests", "hostsim", "libkvz_hostsim.so"))
f = sim.kvz_hostsim_inter_frame
f.restype = None
f.argtypes = [C.c_int] * 4 + [C.c_uint64, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 5
mc = cc.model_constants()
fb = np.array(mc["entropy_fbits"], np.float32)
fe = sim.kvz_hostsim_entropy_code_inter  # the device's entropy coder for B pictures (kvz_entropy.hpp) on the oracle's records, levels and SAO decisions
fe.restype = C.c_long
fe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_uint32, C.c_void_p, C.c_void_p]

bad = 0
for r in range(rounds):
    w, h = int(rng.choice([64, 72, 136, 200])), int(rng.choice([64, 88, 136]))
    n = int(rng.integers(2, 4))
    qp = int(rng.integers(10, 45))
    preset = str(rng.choice(["ultrafast", "superfast", "veryfast", "faster"]))
    dbk, sao, no_wpp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 4) == 0)
    owf = int(rng.integers(0, 2)) if not no_wpp else 0  # the motion restriction is cfg.owf && cfg.wpp
    gop = [(4, 3), (4, 3), (8, 4), (2, 2), (3, 2)][int(rng.integers(0, 5))]
    speed = float(rng.choice([6, 6, 24]))
    src = ("motion", int(rng.integers(1, 1 << 30)), float(rng.uniform(0, 3)), (float(rng.uniform(-speed, speed)), float(rng.uniform(-speed, speed))))
    frames = ic.clip(w, h, n, src[1], src[2], src[3])
    ov = {}  # options that differ from the preset's: --subme 0..4, --fast-residual-cost
    if rng.integers(0, 3) == 0:
        ov["fme_level"] = int(rng.integers(0, 5))
    if rng.integers(0, 3) == 0:
        ov["fast_residual_cost"] = int(rng.choice([0, 20, 28, 35, 51]))
    rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=bool(owf), gop=gop, no_wpp=bool(no_wpp), overrides=ov)
    p = dict(ic.PRESETS[preset])
    p.update(ov)
    ok = True
    for k in range(1, n):
        rec = np.zeros(w * h * 3 // 2, np.uint8)
        out = np.zeros((h // 4, w // 4), ic.CU_DTYPE)
        f(w, h, int(qps[k]), k, int(mc["coeff_weights"][str(int(qps[k]))]), fb.ctypes.data, owf, sao, dbk, p["fme_level"], p["pu_depth_inter_max"], no_wpp, p["fast_residual_cost"],
          np.ascontiguousarray(frames[k]).ctypes.data, np.ascontiguousarray(rf[k - 1]).ctypes.data, np.ascontiguousarray(cu[k - 1]).ctypes.data, rec.ctypes.data, out.ctypes.data)
        ok = ok and ic.first_difference(out[None], cu[k][None]) is None and np.array_equal(rec, rs[k])
    if gop == (4, 3) and not no_wpp and not ov:  # (what tests/inter_common.py oracle_sequence_for_entropy covers) the slice data of every B picture: simulated device coder == the oracle's coder
        parts = ic.oracle_sequence_for_entropy(oracle, ("fuzz", w, h, n, qp, preset, dbk, sao, 2 * owf, src))
        bits = ic.oracle_encode_bits(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=bool(owf))
        hc = (h + 63) // 64
        for k in range(1, n):
            init = ic.b_slice_context_states(oracle, parts["qps"][k])
            recs = merge = None
            if sao:
                recs = ec.pack_sao_records(np.ascontiguousarray(parts["sao_luma"][k]), np.ascontiguousarray(parts["sao_chroma"][k]), parts["ctus"])
                merge = np.ascontiguousarray(parts["merge"][k])
            out, sizes = np.zeros(w * h * 4 + 4096, np.uint8), np.zeros(hc, np.uint32)
            c_k, c_ref, lv = np.ascontiguousarray(parts["cu"][k]), np.ascontiguousarray(parts["cu"][k - 1]), np.ascontiguousarray(parts["coeff"][k])
            total = fe(init.ctypes.data, w, h, k, 0, c_k.ctypes.data, c_ref.ctypes.data, lv.ctypes.data, recs.ctypes.data if recs is not None else None,
                       merge.ctypes.data if merge is not None else None, 49152, out.ctypes.data, sizes.ctypes.data)
            ok = ok and total >= 0 and out[:total].tobytes() == bits[k][0] and [int(v) for v in sizes] == list(bits[k][1])
    b = cu[1:]
    print("round %d: %dx%d x %d %s lp-g%dd%d qp %d (pictures %s) dbk %d sao %d owf %d no_wpp %d %s: intra %d skipped %d merged %d amvp %d -> %s" % (
        r, w, h, n, preset, gop[0], gop[1], qp, list(map(int, qps)), dbk, sao, owf, no_wpp, ov, int((b["type"] == 1).sum()), int(((b["type"] == 2) & (b["skipped"] == 1)).sum()),
        int(((b["type"] == 2) & (b["merged"] == 1)).sum()), int(((b["type"] == 2) & (b["merged"] == 0) & (b["skipped"] == 0)).sum()), "equal" if ok else "DIFFERENT"), flush=True)
    bad += not ok
print("%d of %d rounds differ" % (bad, rounds))
sys.exit(1 if bad else 0)
