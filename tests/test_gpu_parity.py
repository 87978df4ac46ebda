"""Parity tests proper (-m gpu): every per-call entry point of libkvz_hip.so (include/kvz_hip.h), running its HIP
kernels on the MI355X, against the oracle on the seeded cases of tests/cases.py -- bit-exact, no tolerances
(pixel_var included: the summation order is part of the contract)."""
import os

import numpy as np
import pytest

import cases
import flatapi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import kvazaar_amd
    kvazaar_amd.load_library()  # raises if the HIP library is not built: no fallback
    lib = flatapi.FlatLib(kvazaar_amd.LIB_PATH, "kvz_hip_")
    assert lib.lib.kvz_hip_device_count() >= 1
    return lib


@pytest.mark.parametrize("gen", cases.ALL_GENERATORS, ids=lambda g: g.__name__)
def test_hip_equals_oracle(oracle, hip, gen):
    bad, n = [], 0
    for label, run in gen():
        n += 1
        if run(oracle) != run(hip):
            bad.append(label)
    assert not bad, f"{len(bad)}/{n} cases differ: {bad[:12]}"


def test_hip_find_last_scanpos(oracle, hip):
    def st(scan_idx, l2):
        n = 1 << (2 * l2)
        return np.ctypeslib.as_array(oracle.lib.kvz_oracle_scan_table(scan_idx, l2), shape=(n,)).copy()
    bad = [label for label, run in cases.cases_find_last_scanpos(st) if run(oracle) != run(hip)]
    assert not bad, bad[:10]


def test_hip_get_optimized_sad(oracle, hip):
    """get_optimized_sad (strategies-picture.h:128, picture-generic.c:671, AVX2: picture-avx2.c): the pointer the strategy hands back for every width it serves is
    CALLED here -- against reg_sad of that width and, where oracle/_ref is built, against whatever the compiled reference's own AVX2 strategy returns for it -- and a
    width no PU can have must be refused with NULL (search_inter.c:1656 then falls back to reg_sad)."""
    import ctypes as C
    from flatapi import ptr, u8p
    get = hip.lib.kvz_hip_get_optimized_sad
    get.restype = C.c_void_p
    get.argtypes = [C.c_int32]
    proto = C.CFUNCTYPE(C.c_uint32, u8p, u8p, C.c_int32, C.c_uint32, C.c_uint32)
    ref = None
    if os.path.exists(flatapi.refshim_path()):
        ref = flatapi.load_ref(1)
    rng = np.random.default_rng(7)
    a = rng.integers(0, 256, 80 * 70, dtype=np.uint8)
    b = rng.integers(0, 256, 90 * 70, dtype=np.uint8)
    served = 0
    for w in (4, 8, 12, 16, 24, 32, 48, 64):
        addr = get(w)
        assert addr, w
        fn = proto(addr)
        served += 1
        for h in (1, 4, 8, 16, 63, 64):
            for (oa, ob) in ((0, 0), (3, 5), (81, 7)):
                pa, pb = ptr(a[oa:]), ptr(b[ob:])
                got = fn(pa, pb, h, 80, 90)
                assert got == oracle.reg_sad(pa, pb, w, h, 80, 90), (w, h, oa, ob)
                if ref is not None:
                    r = ref.lib.kvz_ref_optimized_sad(w, pa, pb, h, 80, 90)
                    assert r == 0xFFFFFFFF or r == got, (w, h, oa, ob)
    assert served == 8
    for w in (0, 1, 2, 6, 20, 40, 65, 128, -8):
        assert not get(w), w


def test_hip_golden_satd(hip):
    """the reference's own known answers (tests/satd_tests.c:122,140,159) straight through the HIP library"""
    from test_oracle_golden import SATD_GOLDEN, _satd_bufs
    from flatapi import ptr
    for test in (0, 1, 2):
        for w in (2, 3, 4, 5, 6):
            a, b = _satd_bufs(test, w)
            assert hip.satd_nxn(1 << w, ptr(a), ptr(b)) == SATD_GOLDEN[test][w - 2]


def test_hip_threads(oracle, hip):
    """re-entrancy (threadqueue.c:275: strategies are called from N pthread workers): one stream + arena per thread"""
    import threading
    from flatapi import ptr
    rng = np.random.default_rng(5)
    blocks = [(rng.integers(0, 256, 1024, dtype=np.uint8), rng.integers(0, 256, 1024, dtype=np.uint8)) for _ in range(64)]
    want = [oracle.satd_nxn(32, ptr(a), ptr(b)) for a, b in blocks]
    errs = []

    def work(tid):
        for rep in range(20):
            for i, (a, b) in enumerate(blocks):
                if hip.satd_nxn(32, ptr(a), ptr(b)) != want[i]:
                    errs.append((tid, i))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs
