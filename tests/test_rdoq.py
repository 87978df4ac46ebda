"""Rate-distortion optimised quantisation (kvz_rdoq, rdo.c:661-1000): the oracle restatement against the compiled reference's own function on
random transform blocks x context states (CPU, where oracle/_ref exists); the device sources on the host (hostsim) against the oracle; and
under -m gpu the per-call / batched device entry points against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import ctu_common as cc
import flatapi
from flatapi import A, ptr
from test_hostsim import hostsim  # noqa: F401  (fixture)

RDOQ_ARGS = [C.c_int, C.c_double, flatapi.u8p, C.POINTER(C.c_float), flatapi.i16p, flatapi.i16p, C.c_int, C.c_int, C.c_int, C.c_int]


def rdoq_cases(n_per_shape=14):
    """(qp, lambda, ctx states, coefficients, width, type, scan_mode, tr_depth): the shapes kvazaar produces for intra blocks -- 4x4 / 8x8 luma with the three scans,
    16x16 / 32x32 diagonal, chroma 4..16 -- on coefficient statistics from sparse to dense, context states from slice start to heavily adapted"""
    rng = np.random.default_rng(4242)
    out = []
    for (w, typ, scans) in ((4, 0, (0, 1, 2)), (8, 0, (0, 1, 2)), (16, 0, (0,)), (32, 0, (0,)), (4, 2, (0, 1, 2)), (8, 2, (0,)), (16, 2, (0,))):
        for scan in scans:
            for k in range(n_per_shape):
                qp = int(rng.choice([12, 22, 27, 32, 37, 45]))
                lam = 0.57 * 2 ** ((qp - 12) / 3.0) * float(rng.choice([1.0, 0.7, 1.6]))
                ctx = rng.integers(0, 126, 160).astype(np.uint8) if k % 3 else np.full(160, int(rng.integers(0, 126)), np.uint8)
                scale = float(rng.choice([4, 30, 200, 1500]))
                decay = np.exp(-np.add.outer(np.arange(w), np.arange(w)) / float(rng.choice([1.5, 4, 12, 100])))
                coef = np.clip(rng.laplace(0, scale, (w, w)) * decay, -32000, 32000).astype(np.int16)
                if k == 0:
                    coef[:] = 0
                if k == 1:
                    coef[:] = 0
                    coef[w - 1, w - 1] = 900
                out.append((qp, lam, A(ctx), A(coef.reshape(-1)), w, typ, scan, int(rng.integers(0, 2))))
    return out


def rdoq_cases_nxn(n=40):
    """the 4x4 blocks of an NxN CU: kvz_rdoq gets tr_depth 2 for them (quant-generic.c:237-238), and a chroma block then prices its coded-block flag on
    qt_cbf_model_chroma[2], which still has its slice-start state"""
    rng = np.random.default_rng(77)
    out = []
    for k in range(n):
        typ = 2 if k % 2 else 0
        qp = int(rng.choice([12, 17, 22, 27, 32, 37]))
        lam = 0.57 * 2 ** ((qp - 12) / 3.0)
        ctx = rng.integers(0, 126, 160).astype(np.uint8)
        coef = np.clip(rng.laplace(0, float(rng.choice([4, 30, 200, 1500])), (4, 4)), -32000, 32000).astype(np.int16)
        out.append((qp, lam, A(ctx), A(coef.reshape(-1)), 4, typ, int(rng.integers(0, 3)), 2))
    return out


def run_rdoq(fn, fbits, case):
    qp, lam, ctx, coef, w, typ, scan, trd = case
    dest = A(np.full(w * w, 77, np.int16))  # the function leaves positions past the last significant one untouched only when nothing is coded
    fn(qp, lam, ptr(ctx), fbits, ptr(coef), ptr(dest), w, typ, scan, trd)
    return dest.tobytes()


def _fbits():
    return (C.c_float * 128)(*cc.model_constants()["entropy_fbits"])


def test_oracle_rdoq_equals_compiled_reference(oracle):
    if not os.path.exists(flatapi.refshim_path()):
        pytest.skip("oracle/_ref not built")
    ref = flatapi.load_ref(0)
    fo, fr = oracle.lib.kvz_oracle_rdoq, ref.lib.kvz_ref_rdoq
    fo.restype = fr.restype = None
    fo.argtypes = fr.argtypes = RDOQ_ARGS
    fb = _fbits()
    cases = rdoq_cases()
    bad = [i for i, c in enumerate(cases) if run_rdoq(fo, fb, c) != run_rdoq(fr, fb, c)]
    assert not bad, f"{len(bad)}/{len(cases)} blocks differ: {[(cases[i][0], cases[i][4], cases[i][5], cases[i][6]) for i in bad[:8]]}"
    changed = sum(1 for c in cases if np.frombuffer(run_rdoq(fo, fb, c), np.int16).any())
    assert changed > len(cases) // 3  # the rest quantise to nothing at their QP


def test_oracle_rdoq_nxn_blocks_equal_compiled_reference(oracle):
    if not os.path.exists(flatapi.refshim_path()):
        pytest.skip("oracle/_ref not built")
    ref = flatapi.load_ref(0)
    fo, fr = oracle.lib.kvz_oracle_rdoq, ref.lib.kvz_ref_rdoq
    fo.restype = fr.restype = None
    fo.argtypes = fr.argtypes = RDOQ_ARGS
    fb = _fbits()
    cases = rdoq_cases_nxn()
    bad = [i for i, c in enumerate(cases) if run_rdoq(fo, fb, c) != run_rdoq(fr, fb, c)]
    assert not bad, f"{len(bad)}/{len(cases)} blocks differ: {[(cases[i][0], cases[i][5], cases[i][6]) for i in bad[:8]]}"
    # ... and the deep context matters: with tr_depth 1 some of the chroma blocks quantise differently
    rng = np.random.default_rng(5)
    found = 0
    for _ in range(3000):  # blocks around the code-it-or-not threshold: a single small coefficient
        qp = int(rng.choice([22, 27, 32, 37]))
        coef = np.zeros(16, np.int16)
        coef[int(rng.integers(0, 16))] = int(rng.integers(4, 400))
        c = (qp, 0.57 * 2 ** ((qp - 12) / 3.0), A(rng.integers(0, 126, 160).astype(np.uint8)), A(coef), 4, 2, 0, 2)
        a = run_rdoq(fo, fb, c)
        assert a == run_rdoq(fr, fb, c)
        found += a != run_rdoq(fo, fb, c[:7] + (1,))
    assert found > 0


def test_hostsim_rdoq_equals_oracle(oracle, hostsim):
    """the device sources (kvz_rdoq.hpp through the per-call sequence) on the host == the oracle, block by block"""
    fo, fh = oracle.lib.kvz_oracle_rdoq, hostsim.lib.kvz_hostsim_rdoq
    fo.restype = fh.restype = None
    fo.argtypes = fh.argtypes = RDOQ_ARGS
    fb = _fbits()
    cases = rdoq_cases() + rdoq_cases_nxn()  # the latter: tr_depth 2, chroma coded-block flag priced on qt_cbf_model_chroma[2]
    bad = [i for i, c in enumerate(cases) if run_rdoq(fo, fb, c) != run_rdoq(fh, fb, c)]
    assert not bad, f"{len(bad)}/{len(cases)} blocks differ: {[(cases[i][0], cases[i][4], cases[i][5], cases[i][6]) for i in bad[:8]]}"


def test_hostsim_rdoq_dense_blocks_and_every_class(oracle, hostsim):
    """the chain of kvz_rdoq.hpp moves from event to event and decides its class sets on demand: a larger draw of blocks, a good part of them with more than half of their
    positions coded (groups that pass their eighth level, rice parameters that climb to the cap), must still equal the oracle block for block"""
    fo, fh = oracle.lib.kvz_oracle_rdoq, hostsim.lib.kvz_hostsim_rdoq
    fo.restype = fh.restype = None
    fo.argtypes = fh.argtypes = RDOQ_ARGS
    fb = _fbits()
    cases = rdoq_cases(60) + rdoq_cases_nxn(200)
    rng = np.random.default_rng(99)
    for w, typ in ((4, 0), (8, 0), (16, 0), (32, 0), (8, 2)):  # very large levels: escape codes at every rice parameter
        for _ in range(6):
            coef = np.clip(rng.laplace(0, 6000, (w, w)), -32000, 32000).astype(np.int16)
            cases.append((int(rng.choice([12, 22])), 4.0, A(rng.integers(0, 126, 160).astype(np.uint8)), A(coef.reshape(-1)), w, typ, 0, 0))
    dense, bad = 0, []
    for i, c in enumerate(cases):
        a = run_rdoq(fo, fb, c)
        dense += np.count_nonzero(np.frombuffer(a, np.int16)) > c[4] * c[4] // 2
        if a != run_rdoq(fh, fb, c):
            bad.append(i)
    assert not bad, f"{len(bad)}/{len(cases)} blocks differ: {[(cases[i][0], cases[i][4], cases[i][5], cases[i][6]) for i in bad[:8]]}"
    assert dense > 100


@pytest.mark.gpu
def test_hip_rdoq_equals_oracle(oracle):
    """on the MI355X: one block per call, and the same blocks grouped by shape through the batched entry point"""
    import kvazaar_amd
    lib = kvazaar_amd.load_library()
    fo, fh = oracle.lib.kvz_oracle_rdoq, lib.kvz_hip_rdoq
    fo.restype = fh.restype = None
    fo.argtypes = fh.argtypes = RDOQ_ARGS
    fb = _fbits()
    cases = rdoq_cases(6) + rdoq_cases_nxn(24)
    want = [run_rdoq(fo, fb, c) for c in cases]
    assert [run_rdoq(fh, fb, c) for c in cases] == want
    # dense blocks and very large levels: groups that pass their eighth level, every rice parameter (the class sets the chain decides on demand)
    rng = np.random.default_rng(99)
    big = []
    for w, typ in ((4, 0), (8, 0), (16, 0), (32, 0), (8, 2)):
        for scale in (6000, 800):
            coef = np.clip(rng.laplace(0, scale, (w, w)), -32000, 32000).astype(np.int16)
            big.append((int(rng.choice([12, 22])), 4.0, A(rng.integers(0, 126, 160).astype(np.uint8)), A(coef.reshape(-1)), w, typ, 0, 0))
    assert [run_rdoq(fh, fb, c) for c in big] == [run_rdoq(fo, fb, c) for c in big]
    lib.kvz_hip_rdoq_blocks.restype = None
    lib.kvz_hip_rdoq_blocks.argtypes = [C.c_int, C.c_double, flatapi.u8p, flatapi.i16p, flatapi.i16p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    # batches share qp / lambda / contexts / shape: re-run groups of cases with the first member's parameters
    by_shape = {}
    for c in cases:
        by_shape.setdefault((c[4], c[5], c[6], c[7] >= 2), []).append(c)
    for (w, typ, scan, _), group in by_shape.items():
        qp, lam, ctx, _, _, _, _, trd = group[0]
        coef = A(np.concatenate([g[3] for g in group]))
        dest = A(np.full(coef.size, 77, np.int16))
        lib.kvz_hip_rdoq_blocks(qp, lam, ptr(ctx), ptr(coef), ptr(dest), w, typ, scan, trd, len(group))
        exp = b"".join(run_rdoq(fo, fb, (qp, lam, ctx, g[3], w, typ, scan, trd)) for g in group)
        assert dest.tobytes() == exp, (w, typ, scan)


QR_RDOQ_ARGS = [C.POINTER(flatapi.QuantParams), C.c_double, flatapi.u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, flatapi.u8p, flatapi.u8p, flatapi.u8p, flatapi.i16p, C.c_int]


def _fused_cases(n=36):
    """(qp, lambda, ctx, width, color, scan, tr_depth, ref block, pred block with stride, early_skip)"""
    rng = np.random.default_rng(99)
    out = []
    for k in range(n):
        color = int(rng.integers(0, 3))
        w = int(rng.choice([4, 8, 16, 32] if color == 0 else [4, 8, 16]))
        qp = int(rng.choice([17, 22, 27, 32, 37]))
        stride = w + int(rng.choice([0, 8, 24]))
        pred = rng.integers(0, 256, (w, stride)).astype(np.uint8)
        noise = rng.normal(0, float(rng.choice([1.5, 6, 25])), (w, stride))
        ref = np.clip(pred.astype(np.float64) + noise + (8 if k % 5 == 0 else 0), 0, 255).astype(np.uint8)
        scan = int(rng.integers(0, 3)) if w <= 8 else 0
        trd = 2 if (w == 4 and k % 2) else int(rng.integers(0, 2))
        out.append((qp, 0.57 * 2 ** ((qp - 12) / 3.0), A(rng.integers(0, 126, 160).astype(np.uint8)), w, color, scan, trd, A(ref.reshape(-1)), A(pred.reshape(-1)), stride, int(k % 7 == 3)))
    return out


def _fused_expected(oracle, fb, case):
    """quant-generic.c:198-292 with the rdoq leg, from the oracle's pieces: residual, transform, kvz_oracle_rdoq, dequant, inverse, reconstruction"""
    qp, lam, ctx, w, color, scan, trd, ref, pred, stride, early_skip = case
    r2, p2 = ref.reshape(w, stride)[:, :w].astype(np.int16), pred.reshape(w, stride)[:, :w].astype(np.int16)
    resid = A((r2 - p2).reshape(-1))
    idx = (4 if color == 0 else 0) if w == 4 else {8: 1, 16: 2, 32: 3}[w]
    coeff, levels = A(np.zeros(w * w, np.int16)), A(np.full(w * w, 77, np.int16))
    oracle.transform(idx, 8, ptr(resid), ptr(coeff))
    fo = oracle.lib.kvz_oracle_rdoq
    fo.restype, fo.argtypes = None, RDOQ_ARGS
    fo(qp, lam, ptr(ctx), fb, ptr(coeff), ptr(levels), w, 0 if color == 0 else 2, scan, trd)
    rec = p2.astype(np.uint8).copy()
    has = bool(levels.any())
    if has and not early_skip:
        qpar = flatapi.QuantParams(qp=qp, bitdepth=8, slice_is_intra=1, cu_is_intra=1)
        deq, res2 = A(np.zeros(w * w, np.int16)), A(np.zeros(w * w, np.int16))
        oracle.dequant(C.byref(qpar), ptr(levels), ptr(deq), w, w, 0 if color == 0 else (2 if color == 1 else 3), 1)
        oracle.transform(5 + idx, 8, ptr(deq), ptr(res2))
        rec = np.clip((res2.reshape(w, w) + p2).astype(np.int16), 0, 255).astype(np.uint8)
    return int(has), levels.tobytes(), rec.tobytes()


def _run_fused(fn, case):
    qp, lam, ctx, w, color, scan, trd, ref, pred, stride, early_skip = case
    qpar = flatapi.QuantParams(qp=qp, bitdepth=8, slice_is_intra=1, cu_is_intra=1)
    rec, levels = A(np.zeros(w * stride, np.uint8)), A(np.zeros(w * w, np.int16))
    fn.restype, fn.argtypes = C.c_int, QR_RDOQ_ARGS
    has = fn(C.byref(qpar), lam, ptr(ctx), trd, w, color, scan, stride, stride, ptr(ref), ptr(pred), ptr(rec), ptr(levels), early_skip)
    return int(has), levels.tobytes(), rec.reshape(w, stride)[:, :w].tobytes()


def test_hostsim_fused_quantize_residual_rdoq(oracle, hostsim):
    fb = _fbits()
    for case in _fused_cases():
        assert _run_fused(hostsim.lib.kvz_hostsim_quantize_residual_rdoq, case) == _fused_expected(oracle, fb, case), case[:7]


@pytest.mark.gpu
def test_hip_fused_quantize_residual_rdoq(oracle):
    """kvz_hip_quantize_residual_rdoq (what the drop-in's quantize_residual calls with --rdoq): one round trip == the oracle's chain of the same steps"""
    import kvazaar_amd
    lib = kvazaar_amd.load_library()
    fb = _fbits()
    for case in _fused_cases():
        assert _run_fused(lib.kvz_hip_quantize_residual_rdoq, case) == _fused_expected(oracle, fb, case), case[:7]
