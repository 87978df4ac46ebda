"""Randomised HIP-vs-oracle comparison of the batched CTU pass on the MI355X (-m gpu): random picture sizes, content classes, QPs and model switches, every output
identical -- reconstruction, levels, CU depths / modes, the double-precision RD costs, and for the `medium` kernel the NxN flags and 4x4 modes.  The CPU fuzz of
tests/hostsim runs the same sources as plain loops; only here do the DPP / v_sad / v_readlane / MFMA forms of the kernels see random content (tools/fuzz_ctu.py is
the open-ended version of the first test)."""
import os
import sys

import numpy as np
import pytest

import ctu_common as cc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.gpu
def test_hip_ctu_pass_fuzz_equals_oracle(oracle):
    """50 random cases through the kernels of `ultrafast` .. `faster` (fast and CABAC coefficient cost, WPP on / off, adaptive / frozen contexts)"""
    import kvazaar_amd
    import fuzz_ctu
    lib = kvazaar_amd.load_library()
    rng = np.random.default_rng(20260930)
    bad = []
    for i in range(50):
        w, h = int(rng.integers(1, 26)) * 8, int(rng.integers(1, 20)) * 8
        qp = int(rng.choice([0, 7, 12, 17, 22, 27, 32, 37, 42, 47, 51]))
        model = cc.hip_cost_model(lib, qp, cc.coeff_weights(qp))
        model.no_wpp = int(rng.integers(0, 2))
        model.adaptive = int(rng.integers(0, 8) > 0)
        frames = [fuzz_ctu.picture(rng, w, h, int(rng.integers(0, 6))) for _ in range(int(rng.integers(1, 4)))]
        b = cc.HipBatch(lib, w, h, len(frames))
        try:
            for k, f in enumerate(frames):
                b.upload(k, f)
            b.run(model)
            for k, f in enumerate(frames):
                diff = cc.compare(b.download(k), cc.run_oracle(oracle, model, w, h, f))
                if diff:
                    bad.append((i, k, w, h, qp, int(model.no_wpp), int(model.adaptive), diff))
        finally:
            b.close()
    assert not bad, bad[:6]


@pytest.mark.gpu
def test_hip_medium_ctu_pass_fuzz_equals_oracle(oracle):
    """24 random cases through the `medium` kernel (CtuProgramT<true, true, true>): kvz_rdoq in every quantisation, 32x32 CUs searched, 8x8 CUs tried as four 4x4
    PUs -- each switch also off at random (the instantiation's other positions)"""
    import kvazaar_amd
    import fuzz_ctu
    from kvazaar_amd.batch import HipBatch, cost_model
    from test_encoder_parity import oracle_model
    lib = kvazaar_amd.load_library()
    rng = np.random.default_rng(606)
    bad = []
    for i in range(24):
        w, h = int(rng.integers(1, 18)) * 8, int(rng.integers(1, 14)) * 8
        qp = int(rng.choice([7, 12, 17, 22, 27, 32, 37, 42]))
        rdoq, s32, nxn = int(rng.integers(0, 4) > 0), int(rng.integers(0, 4) > 0), int(rng.integers(0, 4) > 0)
        if not (rdoq or nxn):
            rdoq = 1  # at least one of the switches that select this kernel

        def switches(m):
            m.coeff_cabac, m.search_32x32, m.rdoq, m.search_nxn = 1, s32, rdoq, nxn
            return m
        model, omodel = switches(cost_model(lib, qp)), switches(oracle_model(oracle, qp))
        frames = [fuzz_ctu.picture(rng, w, h, int(rng.integers(0, 6))) for _ in range(int(rng.integers(1, 3)))]
        b = HipBatch(lib, w, h, len(frames))
        try:
            for k, f in enumerate(frames):
                b.upload(k, f)
            b.run(model)
            for k, f in enumerate(frames):
                o = b.download(k)
                if nxn:
                    o["part"], o["mode4"] = b.download_partitions(k)
                    diff = cc.compare(cc.run_oracle_nxn(oracle, omodel, w, h, f), o)
                else:
                    diff = cc.compare(cc.run_oracle(oracle, omodel, w, h, f), o)
                if diff:
                    bad.append((i, k, w, h, qp, rdoq, s32, nxn, diff))
        finally:
            b.close()
    assert not bad, bad[:6]
