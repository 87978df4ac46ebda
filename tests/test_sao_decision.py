"""The SAO parameter decision (sao.c:671 kvz_sao_search_lcu, `--sao full`) behind the batched pass.

  * oracle/kvz_oracle_sao.c -- LCU by LCU in the encoder's order, literally deblocking one LCU at a time -- chained behind the oracle's CTU
    pass must reproduce the reference ENCODER's final reconstruction (`kvazaar --preset ultrafast -p 1 --sao full --debug`, digests in
    tests/golden/encoder_recon.json): pins decisions, merge flags and the SAO contexts' flow, band SAO included (adversarial clips);
  * the device sources (kvz_sao.hpp: statistics on the R / V / D view, context-free candidates, chain) compiled for the host must decide
    exactly what the oracle decides -- parameters and merge flags, not only pixels;
  * under -m gpu: kvz_hip_batch_loop_filters on the MI355X == the reference encoder's picture, and its parameters == the oracle's.
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import ctu_common as cc
import flatapi
from flatapi import SaoParams, ptr
from test_encoder_parity import oracle_model
from test_hostsim import hostsim  # noqa: F401  (fixture)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

GOLDEN = json.load(open(os.path.join(HERE, "golden", "encoder_recon.json")))
IDS = lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}{'-nowpp' if c[6] else ''}"  # noqa: E731


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def params_tuple(p):
    return (p.type, p.eo_class, tuple(p.band_position), tuple(p.offsets))


def oracle_sao_chain(oracle, model, w, h, frame, pre=None):
    """CTU pass -> LCU-order deblocking + SAO search -> SAO reconstruction; returns (final picture, deblocked picture, luma, chroma, merge, pass outputs)"""
    o = pre or cc.run_oracle(oracle, model, w, h, frame)
    n = ((w + 63) // 64) * ((h + 63) // 64)
    luma, chroma, merge = (SaoParams * n)(), (SaoParams * n)(), np.zeros(n, np.uint8)
    rec = o["rec"].copy()
    f = oracle.lib.kvz_oracle_sao_search_frame
    f.restype = None
    f(C.byref(model), w, h, ptr(frame), ptr(rec), ptr(o["depth"]), 1, 0, 0, luma, chroma, ptr(merge))
    out = rec.copy()
    g = oracle.lib.kvz_oracle_sao_frame
    g.restype = None
    g(w, h, ptr(rec), ptr(out), luma, chroma)
    return out, rec, luma, chroma, merge, o


@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_SAO, ids=IDS)
def test_oracle_sao_chain_reproduces_reference_encoder(oracle, clip):
    w, h, n, seed, kind, qp, no_wpp = clip
    model = oracle_model(oracle, qp)
    model.no_wpp = 1 if no_wpp else 0
    got = []
    for f in cc.yuv_frames(w, h, n, seed, kind):
        out, dbk, *_ = oracle_sao_chain(oracle, model, w, h, f)
        got.append(_sha(out))
        # deblocking LCU by LCU (the encoder's order) ends at the picture-level result the device computes
        o = cc.run_oracle(oracle, model, w, h, f)
        import deblock_common as dc
        assert np.array_equal(dbk, dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, w, h, qp, 0, 0, o["rec"], o["depth"].reshape(h // 8, w // 8)))
    assert got == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1, no_wpp) + "/sao"]


def _passes(oracle, w, h, qp, rec, depth, passes):
    out = rec.copy()
    ys, cs = w * h, w * h // 4
    f = oracle.lib.kvz_oracle_deblock_frame_passes
    f.restype = None
    f(w, h, qp, 0, 0, ptr(out), ptr(out, offset=ys), ptr(out, offset=ys + cs), ptr(depth), passes)
    return out


@pytest.mark.parametrize("clip", [c for c in mg.ENCODER_CLIPS_SAO if c[0] * c[1] <= 832 * 480], ids=IDS)
def test_hostsim_sao_decision_equals_oracle(oracle, hostsim, clip):
    """device sources on the host: the R / V / D view + statistics + candidates + chain decide what the LCU-order oracle decides"""
    w, h, n, seed, kind, qp, no_wpp = clip
    model = oracle_model(oracle, qp)
    model.no_wpp = 1 if no_wpp else 0
    nl = ((w + 63) // 64) * ((h + 63) // 64)
    for f in cc.yuv_frames(w, h, n, seed, kind):
        _, _, luma, chroma, merge, o = oracle_sao_chain(oracle, model, w, h, f)
        R = o["rec"]
        V = _passes(oracle, w, h, qp, R, o["depth"], 1)
        D = _passes(oracle, w, h, qp, V, o["depth"], 2)
        gl, gc, gm = (SaoParams * nl)(), (SaoParams * nl)(), np.zeros(nl, np.uint8)
        fn = hostsim.lib.kvz_hostsim_sao_decide
        fn.restype = None
        fn(C.byref(model), w, h, ptr(f), ptr(R), ptr(V), ptr(D), gl, gc, ptr(gm))
        assert list(gm) == list(merge)
        assert [params_tuple(p) for p in gl] == [params_tuple(p) for p in luma]
        assert [params_tuple(p) for p in gc] == [params_tuple(p) for p in chroma]


@pytest.mark.gpu
@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_SAO, ids=IDS)
def test_hip_loop_filters_reproduce_reference_encoder(oracle, clip):
    """the product on the MI355X: CTU pass -> kvz_hip_batch_loop_filters(deblock, sao) == the reference encoder's --sao full picture, and the
    parameters / merge flags it decided == the oracle's"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    lib = kvazaar_amd.load_library()
    w, h, n, seed, kind, qp, no_wpp = clip
    model = cost_model(lib, qp)
    model.no_wpp = 1 if no_wpp else 0
    frames = cc.yuv_frames(w, h, n, seed, kind)
    b = HipBatch(lib, w, h, n)
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run(model)
        b.loop_filters(model, deblock=True, sao=True)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1, no_wpp) + "/sao"]
        omodel = oracle_model(oracle, qp)
        omodel.no_wpp = model.no_wpp
        for i, f in enumerate(frames):
            _, _, luma, chroma, merge, _ = oracle_sao_chain(oracle, omodel, w, h, f)
            gl, gc, gm = b.sao_params(i)
            assert list(gm) == list(merge), i
            assert [params_tuple(p) for p in gl] == [params_tuple(p) for p in luma], i
            assert [params_tuple(p) for p in gc] == [params_tuple(p) for p in chroma], i
    finally:
        b.close()
