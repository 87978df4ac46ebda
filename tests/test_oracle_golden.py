"""Pin the oracle against the golden values of the reference's OWN unit tests.

No GPU, no reference tree needed: the numbers below are the known answers hard-coded in
/root/reference/tests/{satd,sad,intra_sad,coeff_sum}_tests.c; the buffers are rebuilt exactly as those
tests build them."""
import math

import numpy as np
import pytest

from flatapi import ptr


def _satd_bufs(test, w):
    n = 1 << w
    size = n * n
    i = np.arange(size)
    if test == 0:      # satd_tests.c:80-85 black / white
        return np.zeros(size, np.uint8), np.full(size, 255, np.uint8)
    if test == 1:      # satd_tests.c:87-95 checkers; buffer 1 = (buffer 0 + 1) % 2
        a = (255 * ((((i >> w) % 2) + (i % 2)) % 2)).astype(np.uint8)
        return a, ((a.astype(np.int32) + 1) % 2).astype(np.uint8)
    col, row = i % n, i // n   # satd_tests.c:97-108 gradient, r = (int)sqrt(row^2+col^2)
    r = np.array([int(math.sqrt(int(rr) * int(rr) + int(cc) * int(cc))) for rr, cc in zip(row, col)])
    a = (255 // (r + 1)).astype(np.uint8)
    return a, (255 - 255 // (r + 1)).astype(np.uint8)


SATD_GOLDEN = {0: [2040, 4080, 16320, 65280, 261120],    # satd_tests.c:122
               1: [2040, 4080, 16320, 65280, 261120],    # satd_tests.c:140
               2: [3140, 9004, 20481, 67262, 258672]}    # satd_tests.c:159


@pytest.mark.parametrize("test", [0, 1, 2])
@pytest.mark.parametrize("w", [2, 3, 4, 5, 6])
def test_satd_known_answers(oracle, test, w):
    a, b = _satd_bufs(test, w)
    n = 1 << w
    r1 = oracle.satd_nxn(n, ptr(a), ptr(b))
    r2 = oracle.satd_nxn(n, ptr(b), ptr(a))
    assert r1 == r2 == SATD_GOLDEN[test][w - 2]


@pytest.mark.parametrize("w", [2, 3, 4, 5, 6])
def test_intra_sad_known_answers(oracle, w):
    """tests/intra_sad_tests.c:151 black vs white = 255*N*N; :160-179 gradient vs constant; symmetry."""
    n = 1 << w
    a, b = _satd_bufs(0, w)
    assert oracle.sad_nxn(n, ptr(a), ptr(b)) == 255 * n * n == oracle.sad_nxn(n, ptr(b), ptr(a))
    g, _ = _satd_bufs(2, w)
    c = np.full(n * n, 128, np.uint8)
    expect = int(np.abs(g.astype(np.int32) - 128).sum())
    assert oracle.sad_nxn(n, ptr(g), ptr(c)) == expect == oracle.sad_nxn(n, ptr(c), ptr(g))


# tests/sad_tests.c:134-272 -- 8x8 hand-made picture/reference, MVs overlapping / outside the frame
REF8 = np.array([1, 2, 2, 2, 2, 2, 2, 3] + [4, 5, 5, 5, 5, 5, 5, 6] * 6 + [7, 8, 8, 8, 8, 8, 8, 9], np.uint8) + 48
PIC8 = np.full(64, 1 + 48, np.uint8)
D = 10
SAD_GOLDEN = [
    ((-3, -3), 1 * 16 + (2 + 4) * 16 + 5 * 16 - 64), ((0, -3), (1 + 3) * 4 + 2 * 24 + (4 + 6) * 4 + 5 * 24 - 64),
    ((3, -3), 3 * 16 + (2 + 6) * 16 + 5 * 16 - 64), ((-3, 0), (1 + 7) * 4 + 4 * 24 + (2 + 8) * 4 + 5 * 24 - 64),
    ((0, 0), (1 + 3 + 7 + 9) + (2 + 4 + 6 + 8) * 6 + 5 * 36 - 64), ((3, 0), (3 + 9) * 4 + 6 * 24 + (2 + 8) * 4 + 5 * 24 - 64),
    ((-3, 3), 7 * 16 + (4 + 8) * 16 + 5 * 16 - 64), ((0, 3), (7 + 9) * 4 + 8 * 24 + (4 + 6) * 4 + 5 * 24 - 64),
    ((3, 3), 9 * 16 + (6 + 8) * 16 + 5 * 16 - 64),
    ((-D, -D), 1 * 64 - 64), ((0, -D), (1 + 3) * 8 + 2 * 48 - 64), ((D, -D), 3 * 64 - 64),
    ((-D, 0), (1 + 7) * 8 + 4 * 48 - 64), ((D, 0), (3 + 9) * 8 + 6 * 48 - 64),
    ((-D, D), 7 * 64 - 64), ((0, D), (7 + 9) * 8 + 8 * 48 - 64), ((D, D), 9 * 64 - 64),
]


@pytest.mark.parametrize("mv,expected", SAD_GOLDEN)
def test_image_calc_sad_known_answers(oracle, mv, expected):
    assert oracle.image_calc_sad(ptr(PIC8), 8, ptr(REF8), 8, 8, 8, 0, 0, mv[0], mv[1], 8, 8) == expected


REG_DIMS = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16), (8, 4), (4, 8),
            (48, 16), (16, 48), (24, 16), (16, 24), (12, 4), (4, 12)]


@pytest.mark.parametrize("dim", REG_DIMS)
def test_reg_sad_patterns(oracle, dim):
    """tests/sad_tests.c:103-111,283-333: (i*i/32+i)%255 vs (i*i/16+i)%255 and all-0 vs all-255, stride 64."""
    w, h = dim
    i = np.arange(64 * 64, dtype=np.int64)
    pic = ((i * i // 32 + i) % 255).astype(np.uint8)
    ref = ((i * i // 16 + i) % 255).astype(np.uint8)
    exp = int(np.abs(pic.reshape(64, 64)[:h, :w].astype(np.int32) - ref.reshape(64, 64)[:h, :w]).sum())
    assert oracle.reg_sad(ptr(pic), ptr(ref), w, h, 64, 64) == exp
    z, m = np.zeros(4096, np.uint8), np.full(4096, 255, np.uint8)
    assert oracle.reg_sad(ptr(z), ptr(m), w, h, 64, 64) == 255 * w * h


def test_coeff_abs_sum_series(oracle):
    """tests/coeff_sum_tests.c:39-62: INT16_MIN .. step 16, closed form."""
    c = np.arange(-32768, 32768, 16).astype(np.int16)
    assert oracle.coeff_abs_sum(ptr(c), len(c)) == int(np.abs(c.astype(np.int64)).sum())


def test_oracle_md5_known_answers(oracle):
    """RFC 1321 appendix A.5 test suite + python's hashlib on plane-sized messages: pins kvz_oracle_plane_md5 (nal-generic.c:41-55)"""
    import hashlib
    import numpy as np
    from flatapi import A, ptr
    known = {b"": "d41d8cd98f00b204e9800998ecf8427e", b"a": "0cc175b9c0f1b6a831c399e269772661", b"abc": "900150983cd24fb0d6963f7d28e17f72",
             b"message digest": "f96b697d7cb7938d525a2f31aaf161d0", b"abcdefghijklmnopqrstuvwxyz": "c3fcd3d76192e4007dfb496cca67e13b",
             b"12345678901234567890123456789012345678901234567890123456789012345678901234567890": "57edf4a22be3c955ac49da2e2107b67a"}
    rng = np.random.default_rng(2)
    msgs = list(known) + [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in (55, 56, 64, 416 * 240, 960 * 540)]
    for m in msgs:
        d = A(np.frombuffer(m or b"\0", np.uint8).copy())
        out = A(np.zeros(16, np.uint8))
        oracle.plane_md5(ptr(d), 1, len(m), len(m), ptr(out))
        assert out.tobytes().hex() == hashlib.md5(m).hexdigest()
        if m in known:
            assert out.tobytes().hex() == known[m]
