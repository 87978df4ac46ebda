#!/usr/bin/env python3
"""Fuzz of the sequence oracle (oracle/kvz_oracle_inter.inc) against the REFERENCE ENCODER itself (oracle/_ref/kvazaar_ref with the ref_cudump.c interposer; only where
/root/reference was compiled, i.e. not on the GPU box): random small clips, picture sizes that cut CTUs, --qp 10..44, the presets ultrafast / superfast / veryfast / faster,
low-delay GOPs of 2, 3, 4 and 8 pictures, slow and fast pans, loop filters on / off, the overlapped-picture motion restriction (--owf 2) on / off, --no-wpp.  The oracle's final pictures must be the encoder's --debug output
every CU decision (type, depth, skip / merge, merge index, vectors, MVP indices, intra mode) the encoder's, and the slice data of every picture (the oracle's
entropy coder) the bytes behind the encoder's slice headers.  usage: tools/fuzz_inter_oracle.py [rounds] [seed]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import flatapi, inter_common as ic, entropy_common as ec

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = flatapi.load_oracle()
bad = 0
for r in range(rounds):
    w, h = int(rng.choice([64, 72, 136, 200, 264])), int(rng.choice([64, 88, 136, 200]))
    n = int(rng.integers(2, 6))
    qp = int(rng.integers(10, 45))
    preset = str(rng.choice(["ultrafast", "superfast", "veryfast", "faster"]))
    dbk, sao, owf, no_wpp = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([0, 2])), int(rng.integers(0, 4) == 0)
    gop = [(4, 3), (4, 3), (8, 4), (2, 2), (3, 2)][int(rng.integers(0, 5))]
    speed = float(rng.choice([6, 6, 24]))
    ov, extra = {}, []  # options that differ from the preset's: --subme (0..4), --fast-residual-cost
    if rng.integers(0, 3) == 0:
        ov["fme_level"] = int(rng.integers(0, 5)); extra += ["--subme", str(ov["fme_level"])]
    if rng.integers(0, 3) == 0:
        ov["fast_residual_cost"] = int(rng.choice([0, 20, 28, 35, 51])); extra += ["--fast-residual-cost", str(ov["fast_residual_cost"])]
    frames = ic.clip(w, h, n, int(rng.integers(1, 1 << 30)), float(rng.uniform(0, 3)), (float(rng.uniform(-speed, speed)), float(rng.uniform(-speed, speed))))
    rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0 and not no_wpp, gop=gop, no_wpp=bool(no_wpp), overrides=ov)
    with tempfile.TemporaryDirectory() as d:
        rrec, rcu = ic.reference_encode(w, h, frames, qp, d, preset=preset, deblock=bool(dbk), sao=bool(sao), owf=owf, gop="lp-g%dd%dt1" % gop, extra=extra + (["--no-wpp"] if no_wpp else []))
        payloads = ec.slice_payloads(open(os.path.join(d, "out.hevc"), "rb").read())
    diff = ic.first_difference(cu, rcu)
    ok = diff is None and np.array_equal(rf, rrec)
    # ... and the slice data the oracle's entropy coder writes for every picture (kvz_oracle_entropy.inc) must be the tail of the encoder's slice NAL payloads
    bits = ic.oracle_encode_bits(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0 and not no_wpp, gop=gop, no_wpp=bool(no_wpp), overrides=ov)
    for payload, (data, sizes) in zip(payloads, bits):
        ok = ok and payload[len(payload) - sum(sizes):] == data and ec.header_ends_with_entry_points(payload[:len(payload) - sum(sizes)], sizes, not no_wpp)
    print("round %d: %dx%d x %d %s lp-g%dd%d qp %d (pictures %s) dbk %d sao %d owf %d no_wpp %d %s -> %s" % (r, w, h, n, preset, gop[0], gop[1], qp, list(map(int, qps)), dbk, sao, owf, no_wpp, " ".join(extra), "equal" if ok else "DIFFERENT %s" % (diff,)), flush=True)
    bad += not ok
print("%d of %d rounds differ" % (bad, rounds))
sys.exit(1 if bad else 0)
