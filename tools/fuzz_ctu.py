#!/usr/bin/env python3
"""Developer tool (GPU): randomised HIP-vs-oracle comparison of the batched CTU pass -- picture sizes, content classes and QPs
beyond what tests/test_gpu_ctu.py fixes.  Every output (reconstruction, coefficients, CU depths / modes, RD costs as doubles)
must be identical.  usage: tools/fuzz_ctu.py [cases=60] [seed=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ctu_common as cc  # noqa: E402
import flatapi  # noqa: E402


def picture(rng, w, h, kind):
    n = w * h * 3 // 2
    if kind == 0:
        return rng.integers(0, 256, n, dtype=np.uint8)
    if kind == 1:
        return np.full(n, rng.integers(0, 256), np.uint8)
    if kind == 2:
        return rng.choice(np.array([0, 255], np.uint8), n)
    yy, xx = np.mgrid[0:h, 0:w]
    a, b, c = rng.integers(-4, 5), rng.integers(-4, 5), rng.integers(0, 256)
    y = (a * xx + b * yy + c + rng.integers(-2, 3, (h, w))) % 256 if kind == 3 else \
        (128 + 100 * np.sin(xx / rng.uniform(3, 40)) * np.cos(yy / rng.uniform(3, 40)) + rng.normal(0, rng.uniform(0, 6), (h, w)))
    u = (xx[::2, ::2] * rng.integers(0, 4) + 90) % 256
    v = (yy[::2, ::2] * rng.integers(0, 4) + 140) % 256
    if kind == 5:  # blocky: sharp 8/16-aligned edges -> mixed CU sizes
        y = (rng.integers(0, 256, (h // 8 + 1, w // 8 + 1))[yy // 8, xx // 8] + rng.integers(-3, 4, (h, w)))
    return np.concatenate([np.clip(y, 0, 255).astype(np.uint8).reshape(-1), u.astype(np.uint8).reshape(-1), v.astype(np.uint8).reshape(-1)])


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    import kvazaar_amd
    lib = kvazaar_amd.load_library()
    oracle = flatapi.load_oracle()
    bad = 0
    for i in range(cases):
        w, h = int(rng.integers(1, 26)) * 8, int(rng.integers(1, 20)) * 8
        qp = int(rng.choice([0, 7, 12, 17, 22, 27, 32, 37, 42, 47, 51]))
        model = cc.hip_cost_model(lib, qp, cc.coeff_weights(qp))
        model.no_wpp = int(rng.integers(0, 2))      # with / without wavefront parallel processing (context hand-off between rows)
        model.adaptive = int(rng.integers(0, 8) > 0)  # mostly kvazaar's adaptive contexts, sometimes frozen ones
        frames = [picture(rng, w, h, int(rng.integers(0, 6))) for _ in range(int(rng.integers(1, 4)))]
        b = cc.HipBatch(lib, w, h, len(frames))
        for k, f in enumerate(frames):
            b.upload(k, f)
        b.run(model)
        for k, f in enumerate(frames):
            diff = cc.compare(b.download(k), cc.run_oracle(oracle, model, w, h, f))
            if diff:
                bad += 1
                print(f"case {i} frame {k}: {w}x{h} qp {qp} no_wpp {model.no_wpp} adaptive {model.adaptive} differs in {diff}", flush=True)
        b.close()
    print(f"fuzz: {cases} cases, {bad} mismatching frames")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
