#!/usr/bin/env python3
"""Fuzz of the all-intra oracle (oracle/kvz_oracle_ctu.c + kvz_oracle_deblock.c) against the REFERENCE ENCODER itself (oracle/_ref/kvazaar_ref -p 1 --debug; only where
/root/reference was compiled): random pictures (the inter fuzz's textured clips, one picture of them, with random noise and contrast), sizes that cut CTUs in every way a
multiple of 8 can, QP 0..51, the searches of ultrafast / faster / fast / medium (CABAC coefficient cost, 32x32 CUs, RDOQ, NxN partitions), deblocking on / off, --no-wpp.
The oracle's picture (after deblocking when it is on) must be the encoder's --debug output.  usage: tools/fuzz_intra_oracle.py [rounds] [seed]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import flatapi, ctu_common as cc, deblock_common as dc, inter_common as ic
import make_golden as mg
from test_encoder_parity import oracle_model

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = flatapi.load_oracle()
bad = 0
for r in range(rounds):
    w, h = int(rng.choice([64, 72, 80, 136, 200, 264])), int(rng.choice([64, 72, 88, 136, 200]))
    qp = int(rng.integers(0, 52))
    preset = str(rng.choice(["ultrafast", "faster", "fast", "medium"]))
    dbk, no_wpp = int(rng.integers(0, 2)), int(rng.integers(0, 4) == 0)
    frame = ic.clip(w, h, 1, int(rng.integers(1, 1 << 30)), float(rng.uniform(0, 8)), (0.0, 0.0))[0]
    if rng.integers(0, 4) == 0:  # low contrast: large CUs, zero-coefficient blocks
        frame = (128 + (frame.astype(np.int32) - 128) // int(rng.integers(4, 16))).astype(np.uint8)
    model = oracle_model(oracle, qp)
    model.no_wpp = no_wpp
    if preset != "ultrafast":
        model.coeff_cabac = 1
    if preset in ("fast", "medium"):
        model.search_32x32 = 1
    if preset == "medium":
        model.rdoq = 1
        model.search_nxn = 1
    o = (cc.run_oracle_nxn if preset == "medium" else cc.run_oracle)(oracle, model, w, h, frame)
    got = o["rec"] if not dbk else dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, w, h, qp, 0, 0, o["rec"], o["depth"].reshape(h // 8, w // 8))
    with tempfile.TemporaryDirectory() as d:
        want = mg.reference_encoder_recon(w, h, [frame], qp, bool(dbk), d, no_wpp=bool(no_wpp), preset=preset)[0]
    ok = np.array_equal(np.asarray(got).reshape(-1), want)
    print("round %d: %dx%d %s qp %d dbk %d no_wpp %d -> %s" % (r, w, h, preset, qp, dbk, no_wpp, "equal" if ok else "DIFFERENT"), flush=True)
    bad += not ok
# ... and the presets as they are, through to the bitstream: the oracle's CTU pass, its SAO parameter decision (presets with --sao full) and its entropy coder must write the
# slice data behind the encoder's slice headers (tests/entropy_common.py's chain on random clips instead of the fixture's)
import entropy_common as ec
ref_exe = os.path.join(ROOT, "oracle", "_ref", "kvazaar_ref")
for r in range(rounds // 2):
    w, h = int(rng.choice([64, 72, 136, 200, 264])), int(rng.choice([64, 88, 136, 200]))
    case = ("fuzz", w, h, int(rng.integers(1, 3)), int(rng.integers(0, 1 << 20)), str(rng.choice(["small", "large"])), int(rng.integers(4, 48)),
            str(rng.choice(["ultrafast", "veryfast", "medium"])), ["--no-wpp"] if rng.integers(0, 4) == 0 else [])
    with tempfile.TemporaryDirectory() as d:
        payloads = ec.reference_slice_payloads(ref_exe, case, d)
    ok = True
    for payload, (data, sizes) in zip(payloads, ec.oracle_slice_data(oracle, case)):
        ok = ok and payload[len(payload) - sum(sizes):] == data and ec.header_ends_with_entry_points(payload[:len(payload) - sum(sizes)], sizes, "--no-wpp" not in case[8])
    print("bitstream round %d: %dx%d x %d %s qp %d %s -> %s" % (r, w, h, case[3], case[7], case[6], " ".join(case[8]), "equal" if ok else "DIFFERENT"), flush=True)
    bad += not ok
print("%d of %d rounds differ" % (bad, rounds + rounds // 2))
sys.exit(1 if bad else 0)
