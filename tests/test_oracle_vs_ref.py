"""Pin the oracle against the compiled reference itself (oracle/_ref built from /root/reference by
oracle/Makefile): every function of the flat API, generic AND AVX2 function pointers, bit-exact on the
seeded cases of tests/cases.py.  Skipped when oracle/_ref has not been built."""
import numpy as np
import pytest

import cases
import flatapi
from flatapi import ptr

# Inputs on which the reference's own AVX2 and generic strategies disagree (both pass the reference's parity
# bar because the encoder never produces them -- SURVEY.md section 8a "AVX2-vs-generic behavioural
# differences").  The oracle follows GENERIC; these labels are only skipped for the AVX2 comparison.
AVX2_OUT_OF_DOMAIN = ()


def test_tables_match_reference(oracle, reflib):
    for n in (4, 8, 16, 32):
        a = np.ctypeslib.as_array(oracle.lib.kvz_oracle_dct_matrix(n), shape=(n * n,))
        b = np.ctypeslib.as_array(reflib.lib.kvz_ref_dct_matrix(n), shape=(n * n,))
        assert np.array_equal(a, b), f"dct matrix {n}"
    a = np.ctypeslib.as_array(oracle.lib.kvz_oracle_dst_matrix(), shape=(16,))
    b = np.ctypeslib.as_array(reflib.lib.kvz_ref_dst_matrix(), shape=(16,))
    assert np.array_equal(a, b)
    for scan in range(3):
        for l2 in range(1, 6):
            n = 1 << (2 * l2)
            a = np.ctypeslib.as_array(oracle.lib.kvz_oracle_scan_table(scan, l2), shape=(n,))
            b = np.ctypeslib.as_array(reflib.lib.kvz_ref_scan_table(scan, l2), shape=(n,))
            assert np.array_equal(a, b), f"scan {scan} log2 {l2}"


@pytest.mark.parametrize("name", ["default", "custom"])
def test_scaling_lists_match_reference(reflib, name):
    """tests/scaling_lists.py (what the scaling-list cases hand the oracle and the product) against the tables of the reference's own encoder control after
    kvz_scalinglist_process (scalinglist.c:407-425): every size, list and QP remainder, forward and inverse"""
    import scaling_lists
    lists = scaling_lists.get(name)
    reflib.set_scaling_list(lists)
    try:
        for l2 in (2, 3, 4, 5):
            n = 1 << (2 * l2)
            for lst in range(6):
                if l2 == 5 and lst not in (0, 1, 3):
                    continue
                for rem in range(6):
                    q, d = lists.tables(l2, lst, rem)
                    a = np.ctypeslib.as_array(reflib.lib.kvz_ref_quant_coeff(l2, lst, rem), shape=(n,))
                    b = np.ctypeslib.as_array(reflib.lib.kvz_ref_dequant_coeff(l2, lst, rem), shape=(n,))
                    assert np.array_equal(q, a) and np.array_equal(d, b), (l2, lst, rem)
    finally:
        reflib.set_scaling_list(None)


def _scan_table(oracle):
    def f(scan_idx, l2):
        n = 1 << (2 * l2)
        return np.ctypeslib.as_array(oracle.lib.kvz_oracle_scan_table(scan_idx, l2), shape=(n,)).copy()
    return f


@pytest.mark.parametrize("gen", cases.ALL_GENERATORS, ids=lambda g: g.__name__)
def test_oracle_equals_reference(oracle, reflib, ref, gen):
    bad = []
    n = 0
    for label, run in gen():
        if ref == 1 and label in AVX2_OUT_OF_DOMAIN:
            continue
        n += 1
        a, b = run(oracle), run(reflib)
        if label.startswith("pixel_var") and ref == 1:
            # the one floating-point function of the path: the reference's own AVX2 version accumulates in
            # float32 (picture-avx2.c), so only generic is compared exactly (SURVEY.md App. A "floating point")
            if abs(a[0] - b[0]) > 1e-6 * max(1.0, abs(a[0])):
                bad.append(label)
        elif a != b:
            bad.append(label)
    assert not bad, f"{len(bad)}/{n} cases differ: {bad[:12]}"
    assert n > 0


def test_find_last_scanpos(oracle, reflib):
    bad = [label for label, run in cases.cases_find_last_scanpos(_scan_table(oracle)) if run(oracle) != run(reflib)]
    assert not bad, bad[:10]


def test_optimized_sad_matches_reg_sad(oracle, reflib, ref):
    """get_optimized_sad (picture-generic.c:671 returns NULL; AVX2 returns width-specialised kernels): whatever
    the strategy hands back must equal reg_sad of that width."""
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, 80 * 70, dtype=np.uint8)
    b = rng.integers(0, 256, 90 * 70, dtype=np.uint8)
    for w in (4, 8, 12, 16, 24, 32, 48, 64):
        for h in (4, 8, 16, 64):
            got = reflib.lib.kvz_ref_optimized_sad(w, ptr(a), ptr(b), h, 80, 90)
            if got == 0xFFFFFFFF:
                continue
            assert got == oracle.reg_sad(ptr(a), ptr(b), w, h, 80, 90)


@pytest.mark.parametrize("size", [(64, 64), (192, 136), (416, 240)])
def test_deblock_frame_matches_reference(oracle, reflib, size):
    """kvz_oracle_deblock_frame (picture-level: all vertical edges, then all horizontal ones) == the reference's LCU-by-LCU
    kvz_filter_deblock_lcu (filter.c:783) on random CU quadtrees, QPs and offsets"""
    import deblock_common as dc
    w, h = size
    rng = np.random.default_rng(w * 7 + h)
    for trial, kind in enumerate(("smooth", "steps", "noise", "steps", "smooth")):
        frame, _ = dc.test_picture(w, h, rng, kind)
        depth = dc.random_depth_map(w, h, rng)
        qp = int(rng.choice([17, 22, 27, 32, 37, 45, 51]))
        b_off, t_off = (0, 0) if trial < 3 else (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
        a = dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, w, h, qp, b_off, t_off, frame, depth)
        b = dc.run_cpu(reflib.lib.kvz_ref_deblock_frame, w, h, qp, b_off, t_off, frame, depth)
        assert np.array_equal(a, b), (kind, qp, b_off, t_off, np.flatnonzero(a != b)[:8])
        if kind != "noise" and qp >= 22:
            assert not np.array_equal(a, frame), "the filter should have changed something"


@pytest.mark.parametrize("size", [(64, 64), (136, 72), (416, 240)])
def test_sao_frame_matches_reference(oracle, reflib, size):
    """kvz_oracle_sao_frame == the reference's kvz_sao_reconstruct (sao.c:302-361) called per CTU and plane with a separate copy
    of the deblocked picture as input: random per-CTU types, classes, band positions and offsets"""
    import sao_common as sc
    w, h = size
    rng = np.random.default_rng(w + 3 * h)
    n = ((w + 63) // 64) * ((h + 63) // 64)
    for _ in range(4):
        frame = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
        luma, chroma = sc.random_params(rng, n, False), sc.random_params(rng, n, True)
        a = sc.run_cpu(oracle.lib.kvz_oracle_sao_frame, w, h, frame, luma, chroma)
        b = sc.run_cpu(reflib.lib.kvz_ref_sao_frame, w, h, frame, luma, chroma)
        assert np.array_equal(a, b), np.flatnonzero(a != b)[:8]
        assert not np.array_equal(a, frame)


def test_residual_coder_bits_match_reference(oracle, reflib):
    """the oracle's restatement of kvz_encode_coeff_nxn in counting mode (what get_coeff_cabac_cost runs, rdo.c:220-263) against
    the reference function on random blocks, random context states, all scans, with and without context updates: same bits,
    same states afterwards"""
    import ctypes as C
    fb = (C.c_float * 128)(*[reflib.lib.kvz_ref_entropy_fbits(i) for i in range(128)])
    reflib.lib.kvz_ref_coeff_cabac_bits.restype = C.c_double
    oracle.lib.kvz_oracle_coeff_cabac_bits.restype = C.c_double
    rng = np.random.default_rng(1)
    for trial in range(1500):
        typ = int(rng.choice([0, 2]))
        w = int(rng.choice([4, 8, 16, 32] if typ == 0 else [4, 8, 16]))
        scan = int(rng.choice([0, 1, 2])) if w <= 8 else 0
        dens, mag = rng.choice([0.02, 0.1, 0.4, 0.9]), int(rng.choice([1, 2, 4, 40, 3000]))
        blk = (rng.random(w * w) < dens) * rng.integers(-mag, mag + 1, w * w)
        if rng.random() < 0.5:
            yy, xx = np.divmod(np.arange(w * w), w)
            blk = blk * ((yy + xx) < w // 2 + 1)
        blk = blk.astype(np.int16)
        if not blk.any():
            blk[int(rng.integers(0, w * w))] = 1
        upd = int(rng.integers(0, 2))
        ctx = rng.integers(0, 126, 136, dtype=np.uint8)
        a, b = ctx.copy(), ctx.copy()
        ra = reflib.lib.kvz_ref_coeff_cabac_bits(blk.ctypes.data_as(C.c_void_p), w, typ, scan, upd, a.ctypes.data_as(C.c_void_p))
        rb = oracle.lib.kvz_oracle_coeff_cabac_bits(fb, blk.ctypes.data_as(C.c_void_p), w, typ, scan, upd, b.ctypes.data_as(C.c_void_p))
        assert ra == rb and np.array_equal(a, b), (trial, w, typ, scan, upd, ra, rb)
        assert upd or np.array_equal(a, ctx)
