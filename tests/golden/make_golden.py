#!/usr/bin/env python3
"""Generates the golden fixtures of this directory from the REFERENCE ITSELF (oracle/_ref, compiled from /root/reference by
oracle/Makefile) -- run in the build container, where /root/reference exists:

    make -C oracle ref && python tests/golden/make_golden.py

Fixtures (small, committed; the GPU box and any machine without the reference tree check against them):
  strategy_cases.json   sha256 of the reference's GENERIC strategy output for every seeded case of tests/cases.py
                        (label -> digest of repr(outputs)), incl. tests/cases.py cases_find_last_scanpos
  deblock.json          sha256 of kvz_filter_deblock_lcu's result (filter.c:783, all LCUs) for seeded pictures / CU quadtrees
  sao_frame.json        sha256 of kvz_sao_reconstruct's result (sao.c:302-361, every CTU and plane) for seeded pictures / parameters
(the bitstream md5s of the reference encoder are asserted by tests/test_e2e_dropin.py)
tests/test_oracle_golden.py holds the known answers of the reference's own unit tests; this file adds outputs of the
compiled reference for the functions those tests do not pin."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
import deblock_common as dc  # noqa: E402
import flatapi  # noqa: E402


def digest(outputs):
    return hashlib.sha256(repr(outputs).encode()).hexdigest()[:24]


def strategy_digests(lib, scan_table):
    out = {}
    for label, run in cases.all_cases():
        r = run(lib)
        if label.startswith("pixel_var"):
            r = tuple(float(np.float64(v)).hex() for v in r)  # the one floating-point output: exact bits of the generic result
        assert label not in out, label
        out[label] = digest(r)
    for label, run in cases.cases_find_last_scanpos(scan_table):
        out["find_last_scanpos/" + label] = digest(run(lib))
    return out


DEBLOCK_CASES = [(64, 64, 1), (192, 136, 2), (416, 240, 3)]


def deblock_digests(func):
    out = {}
    for (w, h, seed) in DEBLOCK_CASES:
        rng = np.random.default_rng(seed)
        for kind in ("smooth", "steps", "noise"):
            frame, _ = dc.test_picture(w, h, rng, kind)
            depth = dc.random_depth_map(w, h, rng)
            for qp, b_off, t_off in ((22, 0, 0), (37, 1, -2)):
                res = dc.run_cpu(func, w, h, qp, b_off, t_off, frame, depth)
                out[f"{w}x{h}/{kind}/qp{qp}/b{b_off}/t{t_off}"] = hashlib.sha256(res.tobytes()).hexdigest()[:24]
    return out


def sao_digests(func):
    import sao_common as sc
    out = {}
    for (w, h, seed) in ((64, 64, 11), (136, 72, 12), (416, 240, 13)):
        rng = np.random.default_rng(seed)
        n = ((w + 63) // 64) * ((h + 63) // 64)
        for trial in range(3):
            frame = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
            luma, chroma = sc.random_params(rng, n, False), sc.random_params(rng, n, True)
            out[f"{w}x{h}/{trial}"] = hashlib.sha256(sc.run_cpu(func, w, h, frame, luma, chroma).tobytes()).hexdigest()[:24]
    return out


def main():
    ref = flatapi.load_ref(0)  # generic strategies
    oracle = flatapi.load_oracle()

    def scan_table(scan_idx, l2):
        n = 1 << (2 * l2)
        return np.ctypeslib.as_array(oracle.lib.kvz_oracle_scan_table(scan_idx, l2), shape=(n,)).copy()

    json.dump(strategy_digests(ref, scan_table), open(os.path.join(HERE, "strategy_cases.json"), "w"), indent=0, sort_keys=True)
    json.dump(deblock_digests(ref.lib.kvz_ref_deblock_frame), open(os.path.join(HERE, "deblock.json"), "w"), indent=0, sort_keys=True)
    json.dump(sao_digests(ref.lib.kvz_ref_sao_frame), open(os.path.join(HERE, "sao_frame.json"), "w"), indent=0, sort_keys=True)
    print("wrote strategy_cases.json, deblock.json, sao_frame.json")


if __name__ == "__main__":
    main()
