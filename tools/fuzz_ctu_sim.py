#!/usr/bin/env python3
"""Fuzz of the all-intra CTU program's device sources (host simulation, tests/hostsim) against the oracle, without a GPU: tools/fuzz_ctu.py's random pictures (noise, flat,
two-level, ramps, waves, blocks) and sizes, QP 0..51, and every switch of the cost model -- CABAC coefficient cost, 32x32 CUs, RDOQ, NxN partitions, WPP, frozen contexts.
Every output (reconstruction, levels, CU depths / modes, NxN flags and PU modes, RD costs as doubles) must be identical.  usage: tools/fuzz_ctu_sim.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import ctu_common as cc, flatapi
from fuzz_ctu import picture
from test_encoder_parity import oracle_model

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle = flatapi.load_oracle()
sim = flatapi.FlatLib(os.path.join(ROOT, "tests", "hostsim", "libkvz_hostsim.so"), "kvz_hostsim_")
bad = 0
for i in range(cases):
    w, h = int(rng.integers(1, 18)) * 8, int(rng.integers(1, 14)) * 8
    qp = int(rng.integers(0, 52))
    model = oracle_model(oracle, qp)
    model.no_wpp = int(rng.integers(0, 4) == 0)
    model.adaptive = int(rng.integers(0, 8) > 0)
    search = str(rng.choice(["ultrafast", "faster", "fast", "medium-pu13", "medium"]))
    if search != "ultrafast":
        model.coeff_cabac = 1
    if search in ("fast", "medium-pu13", "medium"):
        model.search_32x32 = 1
    if search in ("medium-pu13", "medium"):
        model.rdoq = 1
    if search == "medium":
        model.search_nxn = 1
    f = picture(rng, w, h, int(rng.integers(0, 6)))
    if search == "medium":
        diff = cc.compare(cc.run_hostsim_nxn(sim.lib, model, w, h, f), cc.run_oracle_nxn(oracle, model, w, h, f))
    else:
        diff = cc.compare(cc.run_hostsim(sim.lib, model, w, h, f), cc.run_oracle(oracle, model, w, h, f))
    print("case %d: %dx%d %s qp %d no_wpp %d adaptive %d -> %s" % (i, w, h, search, qp, model.no_wpp, model.adaptive, "differs in %s" % diff if diff else "equal"), flush=True)
    bad += bool(diff)
print("%d of %d cases differ" % (bad, cases))
sys.exit(1 if bad else 0)
