"""Seeded test cases for every function of the flat strategy API.

Each case is (label, run) where run(lib) calls the function on a FlatLib (oracle / reference / HIP) with
identical inputs and returns a tuple of comparable outputs (ints, floats, bytes).  A parity test is
`run(lib_a) == run(lib_b)` -- bit-exact, no tolerance anywhere (pixel_var is compared exactly as well:
the operation order is part of the contract).

Input families follow SURVEY.md section 7 step 1: seeded random + adversarial (all-0 / all-255, +-32768
coefficients, every intra mode x size, every fractional MV, every SAO class).
"""
import ctypes as C

import numpy as np

import scaling_lists
from flatapi import (A, EpolParams, IPOL_COL_LEN, IPOL_IM_PLANE, QuantParams, SaoParams, i16p, ptr, u8p)

SIZES = (4, 8, 16, 32, 64)


def _rng(seed):
    return np.random.default_rng(seed)


def _pix_families(rng, n):
    """pixel arrays of length n: random, smooth+noise, extremes"""
    out = [("rand", rng.integers(0, 256, n, dtype=np.uint8)),
           ("low", rng.integers(0, 8, n, dtype=np.uint8)),
           ("zero", np.zeros(n, np.uint8)), ("max", np.full(n, 255, np.uint8))]
    ramp = A((np.arange(n) * 7 % 251).astype(np.uint8))
    out.append(("ramp", ramp))
    return [(l, A(a)) for (l, a) in out]


# ----------------------------------------------------------------------------------------------- picture
def cases_sad_satd_nxn():
    rng = _rng(100)
    for n in SIZES:
        fam_a = _pix_families(rng, n * n)
        fam_b = _pix_families(rng, n * n)
        for (la, a) in fam_a:
            for (lb, b) in fam_b:
                for fn in ("sad_nxn", "satd_nxn"):
                    def run(lib, fn=fn, n=n, a=a, b=b):
                        return (getattr(lib, fn)(n, ptr(a), ptr(b)),)
                    yield (f"{fn}{n}-{la}-{lb}", run)


def cases_dual():
    rng = _rng(101)
    for n in SIZES:
        for rep in range(3):
            # pred_buffer is kvz_pixel(*)[32*32]: candidate 1 starts 1024 bytes after candidate 0 whatever n is, so the
            # 64x64 variants (never called by the encoder) read overlapping 4096-byte windows
            preds = A(rng.integers(0, 256, 1024 + max(1024, n * n), dtype=np.uint8))
            orig = A(rng.integers(0, 256, n * n, dtype=np.uint8))
            if rep == 2:
                preds[:1024] = 0
                preds[1024:] = 255
            for fn in ("sad_nxn_dual", "satd_nxn_dual"):
                def run(lib, fn=fn, n=n, preds=preds, orig=orig):
                    costs = A(np.zeros(2, np.uint32))
                    getattr(lib, fn)(n, ptr(preds), ptr(orig), 2, ptr(costs))
                    return (costs.tobytes(),)
                yield (f"{fn}{n}-{rep}", run)


REG_SAD_DIMS = [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16),
                (8, 4), (4, 8), (48, 16), (16, 48), (24, 16), (16, 24), (12, 4), (4, 12), (1, 1), (64, 63), (5, 3)]


def cases_reg_sad():
    rng = _rng(102)
    for (w, h) in REG_SAD_DIMS:
        for rep in range(2):
            s1, s2 = (64, 64) if rep == 0 else (int(rng.integers(w, 100)), int(rng.integers(w, 100)))
            a = A(rng.integers(0, 256, s1 * h + 64, dtype=np.uint8))
            b = A(rng.integers(0, 256, s2 * h + 64, dtype=np.uint8))

            def run(lib, w=w, h=h, s1=s1, s2=s2, a=a, b=b):
                return (lib.reg_sad(ptr(a), ptr(b), w, h, s1, s2),)
            yield (f"reg_sad{w}x{h}-{rep}", run)


def cases_any_size():
    rng = _rng(103)
    dims = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (4, 4), (4, 8), (8, 4), (12, 16), (16, 12),
            (12, 12), (24, 32), (32, 24), (64, 16), (16, 64), (48, 64), (4, 16), (16, 4), (20, 12)]
    for (w, h) in dims:
        s1, s2 = int(rng.integers(w, 80)), int(rng.integers(w, 80))
        a = A(rng.integers(0, 256, s1 * h + 64, dtype=np.uint8))
        b = A(rng.integers(0, 256, s2 * h + 64, dtype=np.uint8))

        def run(lib, w=w, h=h, s1=s1, s2=s2, a=a, b=b):
            return (lib.satd_any_size(w, h, ptr(a), s1, ptr(b), s2),)
        yield (f"satd_any{w}x{h}", run)
        # quad: 4 candidate planes of stride 64 (LCU_WIDTH, search_inter.c:1118), orig with its own stride
        planes = A(rng.integers(0, 256, (4, 64 * 64), dtype=np.uint8))
        orig = A(rng.integers(0, 256, 64 * 64, dtype=np.uint8))
        os_ = 64

        def runq(lib, w=w, h=h, planes=planes, orig=orig, os_=os_):
            arr = (u8p * 4)(*[ptr(planes[i]) for i in range(4)])
            costs = A(np.zeros(4, np.uint32))
            valid = np.ones(4, np.int8)
            lib.satd_any_size_quad(w, h, arr, 64, ptr(orig), os_, 4, ptr(costs), ptr(valid))
            return (costs.tobytes(),)
        yield (f"satd_quad{w}x{h}", runq)


def cases_ssd_versad_horsad_var():
    rng = _rng(104)
    for w in (4, 8, 16, 32, 64):
        rs, cs = int(rng.integers(w, 80)), int(rng.integers(w, 80))
        a = A(rng.integers(0, 256, rs * w + 64, dtype=np.uint8))
        b = A(rng.integers(0, 256, cs * w + 64, dtype=np.uint8))
        yield (f"ssd{w}", lambda lib, w=w, rs=rs, cs=cs, a=a, b=b: (lib.pixels_calc_ssd(ptr(a), ptr(b), rs, cs, w),))
        zero, mx = A(np.zeros(64 * 64, np.uint8)), A(np.full(64 * 64, 255, np.uint8))
        yield (f"ssd{w}-extreme", lambda lib, w=w, zero=zero, mx=mx: (lib.pixels_calc_ssd(ptr(zero), ptr(mx), 64, 64, w),))
    for (w, h) in [(8, 8), (16, 4), (64, 64), (12, 5), (32, 1)]:
        ps = 80
        pic = A(rng.integers(0, 256, ps * h + 64, dtype=np.uint8))
        ref = A(rng.integers(0, 256, 256 * 70, dtype=np.uint8))
        yield (f"ver_sad{w}x{h}", lambda lib, w=w, h=h, ps=ps, pic=pic, ref=ref: (lib.ver_sad(ptr(pic), ptr(ref), w, h, ps),))
        # image.c:322-395 only calls hor_sad with left or right non-zero
        for (left, right) in [(3, 0), (0, 3), (w - 1, 0), (0, w - 1), (1, 0)]:
            if left >= w or right >= w or (left == 0 and right == 0):
                continue

            def run(lib, w=w, h=h, ps=ps, pic=pic, ref=ref, left=left, right=right):
                return (lib.hor_sad(ptr(pic), ptr(ref, offset=8), w, h, ps, 256, left, right),)
            yield (f"hor_sad{w}x{h}-{left}-{right}", run)
    for ln in (1, 64, 4096, 1000):
        buf = A(rng.integers(0, 256, ln, dtype=np.uint8))
        yield (f"pixel_var{ln}", lambda lib, buf=buf, ln=ln: (lib.pixel_var(ptr(buf), ln),))


def cases_image_calc_sad():
    """image.c frame-edge glue; the first 17 cases are tests/sad_tests.c:134-272 (golden values checked separately)."""
    ref8 = A(np.array([1, 2, 2, 2, 2, 2, 2, 3] + [4, 5, 5, 5, 5, 5, 5, 6] * 6 + [7, 8, 8, 8, 8, 8, 8, 9], np.uint8) + 48)
    pic8 = A(np.full(64, 49, np.uint8))
    for dx in (-10, -3, 0, 3, 10):
        for dy in (-10, -3, 0, 3, 10):
            def run(lib, dx=dx, dy=dy):
                return (lib.image_calc_sad(ptr(pic8), 8, ptr(ref8), 8, 8, 8, 0, 0, dx, dy, 8, 8),)
            yield (f"calc_sad8-{dx}-{dy}", run)
    rng = _rng(105)
    W, H = 96, 80
    pic = A(rng.integers(0, 256, W * H, dtype=np.uint8))
    ref = A(rng.integers(0, 256, W * H, dtype=np.uint8))
    for (bw, bh) in [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (24, 16), (12, 4)]:
        for _ in range(6):
            px, py = int(rng.integers(0, W - bw + 1)), int(rng.integers(0, H - bh + 1))
            rx, ry = int(rng.integers(-bw - 20, W + 20)), int(rng.integers(-bh - 20, H + 20))

            def run(lib, bw=bw, bh=bh, px=px, py=py, rx=rx, ry=ry):
                return (lib.image_calc_sad(ptr(pic), W, ptr(ref), W, H, W, px, py, rx, ry, bw, bh),)
            yield (f"calc_sad{bw}x{bh}-{px},{py}-{rx},{ry}", run)


def cases_bipred():
    rng = _rng(106)
    for (w, h) in [(8, 8), (16, 16), (64, 64), (32, 16), (4, 4), (8, 4), (12, 16)]:
        px0 = A(rng.integers(0, 256, w * h, dtype=np.uint8))
        px1 = A(rng.integers(0, 256, w * h, dtype=np.uint8))
        im0 = A(rng.integers(-2000, 18000, w * h).astype(np.int16))
        im1 = A(rng.integers(-2000, 18000, w * h).astype(np.int16))
        for kind in ("pp", "ii", "pi", "ip"):
            def run(lib, w=w, h=h, kind=kind, px0=px0, px1=px1, im0=im0, im1=im1):
                dst = A(np.zeros(64 * h, np.uint8))
                a = (ptr(px0), None) if kind[0] == "p" else (None, ptr(im0))
                b = (ptr(px1), None) if kind[1] == "p" else (None, ptr(im1))
                lib.bipred_average_plane(ptr(dst), 64, a[0], a[1], b[0], b[1], w, h)
                return (dst.tobytes(),)
            yield (f"bipred{w}x{h}-{kind}", run)


# ----------------------------------------------------------------------------------------------- dct
TR_N = [4, 8, 16, 32, 4, 4, 8, 16, 32, 4]


def gradient_block(n):
    """tests/dct_tests.c:68-92: 64x64 radial gradient, top-left n x n window, as int16"""
    yy, xx = np.mgrid[0:n, 0:n]
    g = (255.0 / 64.0) * np.sqrt(xx * xx + yy * yy) + 0.5
    return np.clip(g.astype(np.int64), 0, 255).astype(np.int16).reshape(-1)


def cases_transform():
    rng = _rng(107)
    for kind in range(10):
        n = TR_N[kind]
        inverse = kind >= 5
        blocks = [("grad", gradient_block(n))]
        if not inverse:
            blocks += [("resid", rng.integers(-255, 256, n * n).astype(np.int16)),
                       ("+255", np.full(n * n, 255, np.int16)), ("-255", np.full(n * n, -255, np.int16)),
                       ("alt", (255 * (1 - 2 * (np.arange(n * n) % 2))).astype(np.int16)),
                       ("chk", (255 * (1 - 2 * ((np.arange(n * n) // n + np.arange(n * n)) % 2))).astype(np.int16)),
                       ("zero", np.zeros(n * n, np.int16))]
        else:
            blocks += [("coef", rng.integers(-2000, 2001, n * n).astype(np.int16)),
                       ("big", rng.integers(-32768, 32768, n * n).astype(np.int16)),
                       ("min", np.full(n * n, -32768, np.int16)), ("max", np.full(n * n, 32767, np.int16)),
                       ("dc", np.concatenate([[1000], np.zeros(n * n - 1)]).astype(np.int16)),
                       ("zero", np.zeros(n * n, np.int16))]
        for (lbl, blk) in blocks:
            blk = A(blk)
            def run(lib, kind=kind, n=n, blk=blk):
                out = A(np.zeros(n * n, np.int16))
                lib.transform(kind, 8, ptr(blk), ptr(out))
                return (out.tobytes(),)
            yield (f"transform{kind}-{lbl}", run)


# ----------------------------------------------------------------------------------------------- quant
def _qp(qp=22, intra=1, signhide=0, cu_intra=1):
    return QuantParams(qp=qp, bitdepth=8, slice_is_intra=intra, signhide=signhide, scaling_list=0, cu_is_intra=cu_intra,
                       quant_coeff=None, dequant_coeff=None)


def cases_quant():
    rng = _rng(108)
    for w in (4, 8, 16, 32):
        for qp in (0, 10, 22, 27, 37, 51):
            for intra in (0, 1):
                for signhide in (0, 1):
                    for typ in (0, 2):
                        amp = [40, 600, 32767][int(rng.integers(0, 3))]
                        coef = A(rng.integers(-amp, amp + 1, w * w).astype(np.int16))
                        if amp == 32767:
                            coef[0], coef[1] = 32767, -32768
                        scan = int(rng.integers(0, 3))

                        def run(lib, w=w, qp=qp, intra=intra, signhide=signhide, typ=typ, coef=coef, scan=scan):
                            p = _qp(qp, intra, signhide)
                            q = A(np.zeros(w * w, np.int16))
                            lib.quant(C.byref(p), ptr(coef), ptr(q), w, w, typ, scan, 1)
                            d = A(np.zeros(w * w, np.int16))
                            lib.dequant(C.byref(p), ptr(q), ptr(d), w, w, typ if typ == 0 else 3, 1)
                            return (q.tobytes(), d.tobytes())
                        yield (f"quant{w}-qp{qp}-i{intra}-s{signhide}-t{typ}", run)


def _chroma_qp(qp):
    """kvz_get_scaled_qp (transform.c:141-155) for a chroma type at 8 bit, H.265 table 8-10"""
    q = min(max(qp, 0), 57)
    return q if q < 30 else (q - 6 if q >= 43 else (29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37)[q - 30])


class _Lists:
    """--scaling-list on for the calls inside: the oracle and the product take the per-coefficient tables through kvz_hip_quant_params (tests/scaling_lists.py
    builds them); the compiled reference reads its encoder control, which load_ref()'s set_scaling_list switches (and back to flat afterwards)"""

    def __init__(self, lib, name):
        self.lib, self.lists = lib, scaling_lists.get(name)

    def __enter__(self):
        if hasattr(self.lib, "set_scaling_list"):
            self.lib.set_scaling_list(self.lists)
        return self.lists

    def __exit__(self, *exc):
        if hasattr(self.lib, "set_scaling_list"):
            self.lib.set_scaling_list(None)


def _qp_lists(lists, w, qp, slice_intra, signhide, cu_intra, type_fwd, type_inv):
    """kvz_hip_quant_params as the kvazaar-side shim fills it with lists on (integration/kvazaar/strategies/hip/quant-hip.c fill_params): the forward table
    of list type_fwd, the inverse one of type_inv (quant-generic.c:59-60, 312-314); returns the arrays too (the struct only holds pointers)"""
    l2 = w.bit_length() - 1
    qf = qp if type_fwd == 0 else _chroma_qp(qp)
    qi = qp if type_inv == 0 else _chroma_qp(qp)
    qt = A(lists.tables(l2, scaling_lists.list_type(cu_intra, type_fwd), qf % 6)[0])
    dt = A(lists.tables(l2, scaling_lists.list_type(cu_intra, type_inv), qi % 6)[1])
    p = QuantParams(qp=qp, bitdepth=8, slice_is_intra=slice_intra, signhide=signhide, scaling_list=1, cu_is_intra=cu_intra,
                    quant_coeff=ptr(qt), dequant_coeff=ptr(dt))
    return p, (qt, dt)


def cases_quant_lists():
    """quant / dequant with scaling lists (quant-generic.c:59-60 forward table, :309-333 inverse: both branches of `shift > qp_scaled / 6` -- the second one needs
    4x4 blocks at QP >= 42, where shift = 6 + 4 - ... drops to the QP period)"""
    rng = _rng(208)
    for name in ("default", "custom"):
        for w in (4, 8, 16, 32):
            for qp in (0, 10, 22, 27, 37, 44, 51):
                for cu_intra in (1, 0):
                    for signhide in (0, 1):
                        for typ in (0, 2, 3):
                            if w == 32 and typ != 0:
                                continue  # no 32x32 chroma lists (scalinglist.c:42: two lists at that size)
                            amp = [40, 600, 32767][int(rng.integers(0, 3))]
                            coef = A(rng.integers(-amp, amp + 1, w * w).astype(np.int16))
                            if amp == 32767:
                                coef[0], coef[1] = 32767, -32768
                            scan = int(rng.integers(0, 3))

                            def run(lib, name=name, w=w, qp=qp, cu_intra=cu_intra, signhide=signhide, typ=typ, coef=coef, scan=scan):
                                with _Lists(lib, name) as lists:
                                    p, keep = _qp_lists(lists, w, qp, cu_intra, signhide, cu_intra, typ, typ)
                                    q = A(np.zeros(w * w, np.int16))
                                    lib.quant(C.byref(p), ptr(coef), ptr(q), w, w, typ, scan, 1 if cu_intra else 2)
                                    d = A(np.zeros(w * w, np.int16))
                                    lib.dequant(C.byref(p), ptr(q), ptr(d), w, w, typ, 1 if cu_intra else 2)
                                    del keep
                                    return (q.tobytes(), d.tobytes())
                            yield (f"quant-{name}{w}-qp{qp}-i{cu_intra}-s{signhide}-t{typ}", run)


def cases_quantize_residual_lists():
    """kvz_quantize_residual with scaling lists: forward table of type 0 / 2, inverse of type 0 / 2 / 3 (a V block quantises with U's list and dequantises with
    its own, quant-generic.c:241, 263)"""
    rng = _rng(209)
    for name in ("default", "custom"):
        for w in (4, 8, 16, 32):
            for color in (0, 1, 2):
                if w == 32 and color != 0:
                    continue
                for qp in (12, 22, 32, 45):
                    for rep in range(3):
                        stride = 64 if color == 0 else 32
                        n = stride * 32 + 64
                        ref = A(rng.integers(0, 256, n, dtype=np.uint8))
                        noise = [3, 25, 255][rep]
                        pred = A(np.clip(ref.astype(np.int32) + rng.integers(-noise, noise + 1, n), 0, 255).astype(np.uint8))
                        alias = rep == 1
                        cu_intra = 1 if rep != 2 else 0
                        signhide = int(rep == 2)
                        scan = int(rng.integers(0, 3)) if w <= 8 else 0
                        early = int(rep == 2 and w == 8)
                        trskip = int(w == 4 and rep == 1)

                        def run(lib, name=name, w=w, color=color, qp=qp, ref=ref, pred=pred, alias=alias, cu_intra=cu_intra,
                                signhide=signhide, scan=scan, early=early, trskip=trskip, stride=stride):
                            with _Lists(lib, name) as lists:
                                p, keep = _qp_lists(lists, w, qp, cu_intra, signhide, cu_intra, 0 if color == 0 else 2, (0, 2, 3)[color])
                                pr = A(pred.copy())
                                rec = A(pr if alias else np.full(len(pr), 7, np.uint8))
                                co = A(np.full(w * w, 99, np.int16))
                                has = lib.quantize_residual(C.byref(p), w, color, scan, trskip, stride, stride, ptr(ref), ptr(pr),
                                                            ptr(rec), ptr(co), early)
                                del keep
                                return (has, rec.tobytes(), co.tobytes())
                        yield (f"qres-{name}{w}-c{color}-qp{qp}-{rep}", run)


def cases_quantize_residual():
    rng = _rng(109)
    for w in (4, 8, 16, 32):
        for color in (0, 1, 2):
            if w == 32 and color != 0:
                continue  # 4:2:0 chroma TUs are at most 16x16; the reference has no 32x32 chroma scaling list
            for qp in (12, 22, 32, 45):
                for rep in range(3):
                    stride = 64 if color == 0 else 32
                    n = stride * 32 + 64
                    ref = A(rng.integers(0, 256, n, dtype=np.uint8))
                    noise = [3, 25, 255][rep]
                    pred = A(np.clip(ref.astype(np.int32) + rng.integers(-noise, noise + 1, n), 0, 255).astype(np.uint8))
                    alias = rep == 1
                    cu_intra = 1 if rep != 2 else 0
                    signhide = int(rep == 2)
                    scan = int(rng.integers(0, 3)) if w <= 8 else 0
                    early = int(rep == 2 and w == 8)
                    trskip = int(w == 4 and rep == 1)

                    def run(lib, w=w, color=color, qp=qp, ref=ref, pred=pred, alias=alias, cu_intra=cu_intra,
                            signhide=signhide, scan=scan, early=early, trskip=trskip, stride=stride):
                        p = _qp(qp, cu_intra, signhide, cu_intra)
                        pr = A(pred.copy())
                        rec = A(pr if alias else np.full(len(pr), 7, np.uint8))
                        co = A(np.full(w * w, 99, np.int16))
                        has = lib.quantize_residual(C.byref(p), w, color, scan, trskip, stride, stride, ptr(ref), ptr(pr),
                                                    ptr(rec), ptr(co), early)
                        return (has, rec.tobytes(), co.tobytes())
                    yield (f"qres{w}-c{color}-qp{qp}-{rep}", run)


def cases_coeff_misc():
    rng = _rng(110)
    # tests/coeff_sum_tests.c:39-62: arithmetic series from INT16_MIN, step 16 -> checked with the closed form too
    series = A(np.arange(-32768, 32768, 16).astype(np.int16))
    yield ("coeff_abs_sum-series", lambda lib: (lib.coeff_abs_sum(ptr(series), len(series)),))
    for ln in (16, 64, 256, 1024, 4096):
        c = A(rng.integers(-32768, 32768, ln).astype(np.int16))
        yield (f"coeff_abs_sum{ln}", lambda lib, c=c, ln=ln: (lib.coeff_abs_sum(ptr(c), ln),))
    for w in (4, 8, 16, 32):
        for wts in (0x0123045608900abc, 0x0300020001000000, 0xffffffffffffffff):
            c = A(rng.integers(-6, 7, w * w).astype(np.int16))
            yield (f"fast_coeff_cost{w}-{wts:x}", lambda lib, c=c, w=w, wts=wts: (lib.fast_coeff_cost(ptr(c), w, wts),))
    for t in (0, 2):
        for qp in range(-12, 70, 3):
            for off in (0, 12):
                yield (f"scaled_qp-{t}-{qp}-{off}", lambda lib, t=t, qp=qp, off=off: (lib.get_scaled_qp(t, qp, off),))


# ----------------------------------------------------------------------------------------------- intra
def cases_intra():
    rng = _rng(111)
    for l2 in (2, 3, 4, 5):
        w = 1 << l2
        refsets = []
        for rep in range(3):
            top = A(rng.integers(0, 256, 2 * w + 1 + 8, dtype=np.uint8))
            left = A(rng.integers(0, 256, 2 * w + 1 + 8, dtype=np.uint8))
            if rep == 1:
                top[:] = np.clip(100 + np.arange(len(top)) * 2, 0, 255)
                left[:] = np.clip(100 - np.arange(len(left)), 0, 255)
            if rep == 2:
                top[:] = 255
                left[:] = 0
            left[0] = top[0]
            refsets.append((top, left))
        for rep, (top, left) in enumerate(refsets):
            for mode in range(2, 35):
                def run(lib, l2=l2, w=w, mode=mode, top=top, left=left):
                    dst = A(np.zeros(w * w, np.uint8))
                    lib.angular_pred(l2, mode, ptr(top), ptr(left), ptr(dst))
                    return (dst.tobytes(),)
                yield (f"angular{w}-m{mode}-{rep}", run)

            def runp(lib, l2=l2, w=w, top=top, left=left):
                dst = A(np.zeros(w * w, np.uint8))
                lib.intra_pred_planar(l2, ptr(top), ptr(left), ptr(dst))
                dc = A(np.zeros(w * w, np.uint8))
                lib.intra_pred_filtered_dc(l2, ptr(top), ptr(left), ptr(dc))
                return (dst.tobytes(), dc.tobytes())
            yield (f"planar_dc{w}-{rep}", runp)


# ----------------------------------------------------------------------------------------------- ipol
def cases_ipol_sample():
    rng = _rng(112)
    S = 96
    frame = A(rng.integers(0, 256, S * S, dtype=np.uint8))
    frame2 = A(np.where(rng.integers(0, 2, S * S) > 0, 255, 0).astype(np.uint8))  # worst-case ringing
    for (w, h) in [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (4, 4), (24, 16), (12, 4)]:
        for (fx, fy) in [(0, 0), (1, 0), (0, 2), (3, 3), (2, 1), (1, 3)]:
            for fi, fr in enumerate((frame, frame2)):
                if w > 64 or h > 64:
                    continue
                org = 12 * S + 12

                def run(lib, w=w, h=h, fx=fx, fy=fy, fr=fr, org=org):
                    mv = A(np.array([fx, fy], np.int16))
                    d8 = A(np.zeros(64 * 64, np.uint8))
                    d16 = A(np.zeros(64 * 64, np.int16))
                    lib.sample_quarterpel_luma(ptr(fr, offset=org), S, w, h, ptr(d8), 64, 1, 1, ptr(mv))
                    lib.sample_quarterpel_luma_hi(ptr(fr, offset=org), S, w, h, ptr(d16), 64, 1, 1, ptr(mv))
                    return (d8.tobytes(), d16.tobytes())
                yield (f"qpel_luma{w}x{h}-{fx}{fy}-{fi}", run)
        if w <= 32 and h <= 32:
            for (fx, fy) in [(0, 0), (1, 7), (4, 4), (5, 2), (7, 3), (3, 6)]:
                org = 12 * S + 12

                def runc(lib, w=w, h=h, fx=fx, fy=fy, org=org):
                    mv = A(np.array([fx, fy], np.int16))
                    d8 = A(np.zeros(32 * 32, np.uint8))
                    d16 = A(np.zeros(32 * 32, np.int16))
                    lib.sample_octpel_chroma(ptr(frame, offset=org), S, w, h, ptr(d8), 32, 1, 1, ptr(mv))
                    lib.sample_octpel_chroma_hi(ptr(frame2, offset=org), S, w, h, ptr(d16), 32, 1, 1, ptr(mv))
                    return (d8.tobytes(), d16.tobytes())
                yield (f"opel_chroma{w}x{h}-{fx}{fy}", runc)


def cases_ipol_blocks():
    """The FME sequence of search_frac (search_inter.c:1079-1102): hpel hor_ver, hpel diag, then for each
    half-pel offset qpel hor_ver + qpel diag, sharing the intermediate buffers between calls."""
    rng = _rng(113)
    S = 96
    frames = A([rng.integers(0, 256, S * S, dtype=np.uint8), np.where(rng.integers(0, 2, S * S) > 0, 255, 0).astype(np.uint8)])
    for (w, h) in [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 64)]:
        for fi, fr in enumerate(frames):
            for (ox, oy) in [(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, 1), (-1, 1), (1, -1)]:
                org = 12 * S + 12

                def run(lib, w=w, h=h, fr=fr, ox=ox, oy=oy, org=org):
                    filt = A(np.zeros(4 * 64 * 64, np.uint8))
                    im = A(np.zeros(5 * IPOL_IM_PLANE, np.int16))
                    cols = A(np.zeros(5 * IPOL_COL_LEN, np.int16))
                    outs = []
                    src = ptr(fr, offset=org)
                    for fn in ("filter_hpel_blocks_hor_ver_luma", "filter_hpel_blocks_diag_luma",
                               "filter_qpel_blocks_hor_ver_luma", "filter_qpel_blocks_diag_luma"):
                        getattr(lib, fn)(src, S, w, h, ptr(filt), ptr(im), 4, ptr(cols), ox, oy)
                        planes = filt.reshape(4, 64, 64)[:, :h, :w]
                        outs.append(planes.tobytes())
                    return tuple(outs)
                yield (f"fme{w}x{h}-{fi}-o{ox}{oy}", run)


def cases_extended_block():
    rng = _rng(114)
    W, H = 64, 48
    frame = A(rng.integers(0, 256, W * H, dtype=np.uint8))
    for (bx, by, bw, bh) in [(-5, -5, 8, 8), (60, 44, 8, 8), (-20, 10, 16, 16), (30, -30, 16, 8), (70, 60, 8, 8),
                             (10, 10, 8, 8), (0, 0, 64, 48), (-3, 20, 32, 32), (50, 0, 16, 16)]:
        for (pl, pr, pt, pb, ps) in [(3, 4, 3, 4, 0), (4, 4, 4, 4, 1), (0, 0, 0, 0, 0)]:
            a = EpolParams(src_w=W, src_h=H, src_s=W, blk_x=bx, blk_y=by, blk_w=bw, blk_h=bh, pad_l=pl, pad_r=pr,
                           pad_t=pt, pad_b=pb, pad_b_simd=ps)

            def run(lib, a=a, bw=bw, bh=bh, pl=pl, pr=pr, pt=pt, pb=pb, ps=ps):
                ext_s = pl + bw + pr
                rows = pt + bh + pb + ps
                buf = A(np.full(ext_s * rows + 16, 0xAB, np.uint8))
                used = lib.get_extended_block(C.byref(a), ptr(frame), ptr(buf))
                return (used, buf.tobytes() if used else b"")
            yield (f"epol-{bx},{by}-{bw}x{bh}-{pl}{pr}{pt}{pb}{ps}", run)


# ----------------------------------------------------------------------------------------------- sao
def cases_sao():
    rng = _rng(115)
    for (bw, bh) in [(64, 64), (32, 32), (16, 16), (8, 8), (64, 17), (20, 64), (3, 3)]:
        orig = A(rng.integers(0, 256, bw * bh, dtype=np.uint8))
        rec = A(np.clip(orig.astype(np.int32) + rng.integers(-6, 7, bw * bh), 0, 255).astype(np.uint8))
        for eo in range(4):
            offs = np.array([0, int(rng.integers(0, 8)), int(rng.integers(0, 8)), -int(rng.integers(0, 8)), -int(rng.integers(0, 8))], np.int32)

            def run(lib, bw=bw, bh=bh, orig=orig, rec=rec, eo=eo, offs=offs):
                stats = A(np.arange(10, dtype=np.int32))  # accumulates into the caller's array
                lib.calc_sao_edge_dir(8, ptr(orig), ptr(rec), eo, bw, bh, ptr(stats))
                dd = lib.sao_edge_ddistortion(8, ptr(orig), ptr(rec), bw, bh, eo, ptr(offs))
                return (stats.tobytes(), dd)
            yield (f"sao_edge{bw}x{bh}-eo{eo}", run)
        for bp in (0, 5, 13, 28, 31):
            bands = A(rng.integers(-7, 8, 4).astype(np.int32))
            if bp == 5:
                bands[1] = 0

            def runb(lib, bw=bw, bh=bh, orig=orig, rec=rec, bp=bp, bands=bands):
                return (lib.sao_band_ddistortion(8, ptr(orig), ptr(rec), bw, bh, bp, ptr(bands)),)
            yield (f"sao_band{bw}x{bh}-bp{bp}", runb)
    # reconstruct: block inside a larger padded buffer (the edge classes read the 1-px ring, sao.c:321-348)
    S = 80
    big = A(rng.integers(0, 256, S * S, dtype=np.uint8))
    flat = A(np.full(S * S, 128, np.uint8))
    for (bw, bh) in [(64, 64), (32, 32), (63, 10), (9, 64), (1, 1)]:
        for color in (0, 1, 2):
            for typ in (1, 2):
                for eo in range(4 if typ == 2 else 1):
                    for bi, base in enumerate((big, flat)):
                        sp = SaoParams(type=typ, eo_class=eo, bitdepth=8)
                        sp.band_position[0], sp.band_position[1] = int(rng.integers(0, 29)), int(rng.integers(0, 29))
                        for i in range(10):
                            sp.offsets[i] = int(rng.integers(-7, 8))
                        sp.offsets[0] = sp.offsets[5] = 0

                        def run(lib, bw=bw, bh=bh, color=color, sp=sp, base=base):
                            out = A(np.full(72 * 72, 0x5A, np.uint8))
                            lib.sao_reconstruct_color(C.byref(sp), ptr(base, offset=4 * S + 4), ptr(out), S, 72, bw, bh, color)
                            return (out.tobytes(),)
                        yield (f"sao_rec{bw}x{bh}-c{color}-t{typ}-eo{eo}-{bi}", run)


def cases_find_last_scanpos(scan_table):
    """scan_table(scan_idx, log2) -> numpy uint32 array (taken from the oracle; the scan tables themselves are
    pinned against the reference in test_oracle_vs_ref.py)."""
    rng = _rng(116)
    for w in (4, 8, 16, 32):
        l2 = {4: 2, 8: 3, 16: 4, 32: 5}[w]
        for scan_idx in range(3):
            scan = scan_table(scan_idx, l2)
            for rep in range(4):
                coef = A(np.zeros(w * w, np.int16))
                if rep == 1:
                    coef[:] = rng.integers(-3, 4, w * w)
                elif rep == 2:
                    k = int(rng.integers(0, w * w))
                    coef[scan[k]] = int(rng.integers(200, 4000))
                elif rep == 3:
                    coef[:] = rng.integers(-32768, 32768, w * w)
                qc = A(np.full(w * w, 16384, np.int16))
                q_bits = 14 + 22 // 6 + (15 - 8 - l2)
                typ = int(rng.integers(0, 2)) * 2

                def run(lib, w=w, coef=coef, qc=qc, q_bits=q_bits, typ=typ, scan=scan, scan_idx=scan_idx):
                    # caller contract (rdo.c:694-735): ctx_set = 0, cg_last_scanpos = last_scanpos = -1 on entry; only the
                    # dest_coeff entries AFTER the last significant scan position are defined on return (the AVX2
                    # strategy clears more), and sig_coeff_inc only at the found position.
                    dest = A(np.full(w * w, 77, np.int16))
                    sig = A(np.full(32 * 32, -9, np.int32))
                    ctx_set = np.array([0], np.uint16)
                    cg_last = np.array([-1], np.int32)
                    last = np.array([-1], np.int32)
                    cg_pos = np.array([-5], np.int32)
                    c = A(coef.copy())
                    lib.find_last_scanpos(ptr(c), ptr(dest), typ, q_bits, ptr(qc), ptr(sig), 16, ptr(ctx_set), ptr(scan),
                                          ptr(cg_last), ptr(last), (w * w) // 16, ptr(cg_pos), w, scan_idx)
                    lp = int(last[0])
                    tail = dest[scan[lp + 1:]].tobytes()
                    sig_at = int(sig[scan[lp]]) if lp >= 0 else None
                    return (tail, sig_at, int(ctx_set[0]), int(cg_last[0]), lp, int(cg_pos[0]))
                yield (f"last_scanpos{w}-s{scan_idx}-{rep}", run)


def cases_plane_checksum():
    """nal-generic.c:57-82: planes wider / taller than 256 exercise the x >> 8 / y >> 8 terms of the mask, strides > width the row step"""
    rng = _rng(120)
    for (h, w, stride) in ((1, 1, 1), (8, 8, 8), (16, 24, 40), (120, 208, 208), (270, 300, 304), (64, 520, 520), (515, 64, 72),
                           (1080, 1920, 1920), (544, 960, 1984), (2160, 3840, 3840)):  # whole luma planes: 2 MB / 8 MB through the per-call path (nal.c:79-84)
        d = A(rng.integers(0, 256, h * stride, dtype=np.uint8))
        yield (f"plane_checksum{w}x{h}s{stride}", lambda lib, d=d, h=h, w=w, stride=stride: (lib.plane_checksum(ptr(d), h, w, stride),))
    full = A(np.full(64 * 64, 255, np.uint8))
    yield ("plane_checksum-ff", lambda lib: (lib.plane_checksum(ptr(full), 64, 64, 64),))
    # nal-generic.c:41-55 array_md5: message lengths around the 56 / 64-byte padding boundaries, a picture-sized plane (chunked staging)
    for (h, w) in ((1, 1), (1, 55), (1, 56), (7, 9), (8, 8), (1, 119), (3, 40), (120, 208), (1080, 1920), (1089, 961)):
        d = A(rng.integers(0, 256, h * w, dtype=np.uint8))

        def md5(lib, d=d, h=h, w=w):
            out = A(np.zeros(16, np.uint8))
            lib.plane_md5(ptr(d), h, w, w, ptr(out))
            return (out.tobytes(),)
        yield (f"plane_md5-{w}x{h}", md5)


ALL_GENERATORS = [cases_plane_checksum, cases_sad_satd_nxn, cases_dual, cases_reg_sad, cases_any_size, cases_ssd_versad_horsad_var,
                  cases_image_calc_sad, cases_bipred, cases_transform, cases_quant, cases_quantize_residual, cases_quant_lists, cases_quantize_residual_lists,
                  cases_coeff_misc, cases_intra, cases_ipol_sample, cases_ipol_blocks, cases_extended_block, cases_sao]


def all_cases():
    for g in ALL_GENERATORS:
        for c in g():
            yield c
