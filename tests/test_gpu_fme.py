"""kvz_hip_dev_fme_costs on the MI355X (-m gpu) against search_frac's own sequence run through the oracle (itself pinned to the compiled
reference): kvz_get_extended_block -> the four filter_{hpel,qpel}_blocks_* functions sharing their intermediate buffers ->
kvz_satd_any_size / kvz_satd_any_size_quad (search_inter.c:1016-1118), for every half-pel offset, PU sizes 8..64 (square and 2:1),
motion vectors that push the window over every picture edge."""
import ctypes as C

import numpy as np
import pytest

import flatapi
from flatapi import A, EpolParams, IPOL_COL_LEN, IPOL_IM_PLANE, ptr, u8p


class FmePu(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_int16), ("h", C.c_int16), ("mv_x", C.c_int16), ("mv_y", C.c_int16),
                ("hpel_x", C.c_int8), ("hpel_y", C.c_int8), ("reserved", C.c_int16)]


FILTERS = ("filter_hpel_blocks_hor_ver_luma", "filter_hpel_blocks_diag_luma", "filter_qpel_blocks_hor_ver_luma", "filter_qpel_blocks_diag_luma")


def oracle_fme(oracle, cur, ref, W, H, pu):
    """search_frac's arithmetic for one PU -> 17 costs (integer position, then 4 per step)"""
    x, y, w, h, mvx, mvy, ox, oy = pu
    a = EpolParams(src_w=W, src_h=H, src_s=W, blk_x=x + mvx - 1, blk_y=y + mvy - 1, blk_w=w + 1, blk_h=h + 1, pad_l=3, pad_r=4, pad_t=3, pad_b=4, pad_b_simd=0)
    ext_s = 3 + w + 1 + 4
    buf = A(np.zeros(ext_s * (3 + h + 1 + 4) + 64, np.uint8))
    used = oracle.get_extended_block(C.byref(a), ptr(ref), ptr(buf))
    if used:
        src, stride, org = buf, ext_s, 3 * ext_s + 3
    else:  # the window lies inside the picture: the function hands back a pointer into the frame
        src, stride, org = ref, W, (y + mvy - 1) * W + x + mvx - 1
    blk = A(cur.reshape(H, W)[y:y + h, x:x + w].copy().reshape(-1))
    costs = [oracle.satd_any_size(w, h, ptr(blk), w, ptr(src, offset=org + stride + 1), stride)]
    filt = A(np.zeros(4 * 64 * 64, np.uint8))
    im = A(np.zeros(5 * IPOL_IM_PLANE, np.int16))
    cols = A(np.zeros(5 * IPOL_COL_LEN, np.int16))
    for step, fn in enumerate(FILTERS):
        getattr(oracle, fn)(ptr(src, offset=org), stride, w, h, ptr(filt), ptr(im), 4, ptr(cols), ox if step >= 2 else 0, oy if step >= 2 else 0)
        arr = (u8p * 4)(*[ptr(filt, offset=4096 * i) for i in range(4)])
        c = A(np.zeros(4, np.uint32))
        valid = np.ones(4, np.int8)
        oracle.satd_any_size_quad(w, h, arr, 64, ptr(blk), w, 4, ptr(c), ptr(valid))
        costs += [int(v) for v in c]
    return costs


@pytest.mark.gpu
def test_dev_fme_costs_equal_search_frac_sequence(oracle):
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    lib.kvz_hip_dev_fme_costs.restype = None
    lib.kvz_hip_dev_fme_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    W, H = 176, 144
    rng = np.random.default_rng(31)
    yy, xx = np.mgrid[0:H, 0:W]
    ref = A(np.clip(128 + 70 * np.sin(xx / 5.0) * np.cos(yy / 7.0) + rng.normal(0, 12, (H, W)), 0, 255).astype(np.uint8).reshape(-1))
    cur = A(np.clip(np.roll(ref.reshape(H, W), (2, -3), (0, 1)).astype(np.int32) + rng.integers(-9, 10, (H, W)), 0, 255).astype(np.uint8).reshape(-1))
    pus = []
    for (w, h) in [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (64, 32), (24, 8), (48, 64)]:
        for k in range(10):
            x, y = int(rng.integers(0, (W - w) // 8 + 1)) * 8, int(rng.integers(0, (H - h) // 8 + 1)) * 8
            mvx, mvy = int(rng.integers(-12, 13)), int(rng.integers(-12, 13))
            if k == 0: x, y, mvx, mvy = 0, 0, -9, -7             # window over the top-left corner
            if k == 1: x, y, mvx, mvy = W - w, H - h, 11, 6      # ... the bottom-right corner
            if k == 2: mvx, mvy = -200, 3                        # entirely left of the picture
            if k == 3: mvx, mvy = 2, 300                         # entirely below
            pus.append((x, y, w, h, mvx, mvy, int(rng.integers(-1, 2)), int(rng.integers(-1, 2))))
    for ox in (-1, 0, 1):                                        # every half-pel offset of the quarter-pel steps
        for oy in (-1, 0, 1):
            pus.append((40, 32, 16, 16, 3, -2, ox, oy))
    arr = (FmePu * len(pus))(*[FmePu(*p, 0) for p in pus])
    d_cur, d_ref = dev.put(cur), dev.put(ref)
    d_pus = dev.empty(C.sizeof(arr))
    lib.kvz_hip_dev_upload(d_pus, C.addressof(arr), C.sizeof(arr))
    d_out = dev.empty(len(pus) * 17 * 4)
    for max_size in (64,):
        lib.kvz_hip_dev_fme_costs(d_cur, d_ref, W, H, d_pus, len(pus), max_size, 15, d_out)
        got = dev.get(d_out, (len(pus), 17), np.uint32)
        bad = [(i, pus[i]) for i, p in enumerate(pus) if list(got[i]) != oracle_fme(oracle, cur, ref, W, H, p)]
        assert not bad, f"{len(bad)}/{len(pus)} PUs differ: {bad[:5]}"
    # the size-specialised instantiations (LDS sized for 16 / 32) on the PUs that fit them
    for max_size in (16, 32):
        sel = [i for i, p in enumerate(pus) if p[2] <= max_size and p[3] <= max_size]
        sub = (FmePu * len(sel))(*[FmePu(*pus[i], 0) for i in sel])
        lib.kvz_hip_dev_upload(d_pus, C.addressof(sub), C.sizeof(sub))
        lib.kvz_hip_dev_fme_costs(d_cur, d_ref, W, H, d_pus, len(sel), max_size, 3, d_out)  # the `veryfast` call: both half-pel steps
        sub_got = dev.get(d_out, (len(pus), 17), np.uint32)[:len(sel)]
        assert [list(sub_got[k][:9]) for k in range(len(sel))] == [list(got[i][:9]) for i in sel]
    dev.free(d_cur, d_ref, d_pus, d_out)


def test_oracle_fme_sequence_matches_compiled_reference(oracle):
    """the composition above through the compiled reference's own functions (generic strategies), where oracle/_ref exists"""
    import os
    if not os.path.exists(flatapi.refshim_path()):
        pytest.skip("oracle/_ref not built")
    ref_lib = flatapi.load_ref(0)
    W, H = 96, 80
    rng = np.random.default_rng(5)
    ref = A(rng.integers(0, 256, W * H, dtype=np.uint8))
    cur = A(rng.integers(0, 256, W * H, dtype=np.uint8))
    for pu in [(0, 0, 16, 16, -5, -5, 1, -1), (64, 48, 32, 32, 7, 9, 0, 1), (24, 16, 8, 8, 2, 1, -1, 0), (16, 8, 64, 64, 0, 0, 1, 1)]:
        assert oracle_fme(oracle, cur, ref, W, H, pu) == oracle_fme(ref_lib, cur, ref, W, H, pu), pu


# ---- motion-compensated prediction -------------------------------------------------------------------------------------------------
class McPu(C.Structure):
    _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_int16), ("h", C.c_int16), ("mv", (C.c_int16 * 2) * 2), ("use", C.c_int8 * 2), ("reserved", C.c_int16)]


def _clamp_copy(plane, fw, fh, x0, y0, w, h):
    ys = np.clip(np.arange(y0, y0 + h), 0, fh - 1)
    xs = np.clip(np.arange(x0, x0 + w), 0, fw - 1)
    return plane.reshape(fh, fw)[np.ix_(ys, xs)].copy()


def _unipred(oracle, ref_plane, fw, fh, x, y, w, h, mv, chroma, want_hi):
    """inter_recon_unipred for one plane (inter.c:371-500), following its branches: -> (pixels or None, 14-bit samples or None)"""
    sh = 1 if chroma else 0
    int_mv = (mv[0] >> 2, mv[1] >> 2)
    frac_luma = (mv[0] & 3) or (mv[1] & 3)
    frac_chroma = (int_mv[0] & 1) or (int_mv[1] & 1)
    interpolate = (frac_luma or frac_chroma) if chroma else frac_luma
    if not interpolate:
        fx, fy = ((int_mv[0] + x * 2) // 2, (int_mv[1] + y * 2) // 2) if chroma else (int_mv[0] + x, int_mv[1] + y)  # int_mv_in_frame (/ 2 for chroma)
        return _clamp_copy(ref_plane, fw, fh, fx, fy, w, h), None
    pad = (1, 2, 3) if chroma else (3, 4, 1)
    bx, by = x + (mv[0] >> (2 + sh)), y + (mv[1] >> (2 + sh))
    a = EpolParams(src_w=fw, src_h=fh, src_s=fw, blk_x=bx, blk_y=by, blk_w=w, blk_h=h, pad_l=pad[0], pad_r=pad[1], pad_t=pad[0], pad_b=pad[1], pad_b_simd=pad[2])
    ext_s = pad[0] + w + pad[1]
    buf = A(np.zeros(ext_s * (pad[0] + h + pad[1] + pad[2]) + 64, np.uint8))
    plane = A(ref_plane)
    if oracle.get_extended_block(C.byref(a), ptr(plane), ptr(buf)):
        src, stride, org = buf, ext_s, pad[0] * ext_s + pad[0]
    else:
        src, stride, org = plane, fw, by * fw + bx
    mvp = A(np.array(mv, np.int16))
    if want_hi:
        d = A(np.zeros(w * h, np.int16))
        (oracle.sample_octpel_chroma_hi if chroma else oracle.sample_quarterpel_luma_hi)(ptr(src, offset=org), stride, w, h, ptr(d), w, (mv[0] & (7 if chroma else 3)), (mv[1] & (7 if chroma else 3)), ptr(mvp))
        return None, d.reshape(h, w)
    d = A(np.zeros(w * h, np.uint8))
    (oracle.sample_octpel_chroma if chroma else oracle.sample_quarterpel_luma)(ptr(src, offset=org), stride, w, h, ptr(d), w, (mv[0] & (7 if chroma else 3)), (mv[1] & (7 if chroma else 3)), ptr(mvp))
    return d.reshape(h, w), None


def oracle_inter_pred(oracle, refs, W, H, pus):
    pred = np.zeros(W * H * 3 // 2, np.uint8)
    planes = lambda f: ((f[:W * H], W, H, 0), (f[W * H:W * H * 5 // 4], W // 2, H // 2, W * H), (f[W * H * 5 // 4:], W // 2, H // 2, W * H * 5 // 4))  # noqa: E731
    for (x, y, w, h, mv0, mv1, use0, use1) in pus:
        for pi in range(3):
            chroma = pi > 0
            sh = 1 if chroma else 0
            px, py, pw, ph = x >> sh, y >> sh, w >> sh, h >> sh
            outs = []
            for l, (use, mv) in enumerate(((use0, mv0), (use1, mv1))):
                if not use:
                    continue
                plane, fw, fh, off = planes(refs[l])[pi]
                outs.append(_unipred(oracle, plane, fw, fh, px, py, pw, ph, mv, chroma, want_hi=bool(use0 and use1)))
            if len(outs) == 1:
                res = outs[0][0]
            else:  # kvz_bipred_average (picture-generic.c:616-668): pixel or 14-bit operands per list
                dst = A(np.zeros(pw * ph, np.uint8))
                args, keep = [], []
                for p8, p16 in outs:
                    keep.append(A((p8 if p8 is not None else p16).reshape(-1)))  # keeps the operand alive behind the raw pointer
                    args += [ptr(keep[-1]) if p8 is not None else None, ptr(keep[-1]) if p16 is not None else None]
                oracle.bipred_average_plane(ptr(dst), pw, args[0], args[1], args[2], args[3], pw, ph)
                res = dst.reshape(ph, pw)
            _, fw, fh, off = planes(pred)[pi]
            pred[off:off + fw * fh].reshape(fh, fw)[py:py + ph, px:px + pw] = res
    return pred


@pytest.mark.gpu
def test_dev_inter_pred_equals_reference_branches(oracle):
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    lib.kvz_hip_dev_inter_pred.restype = None
    lib.kvz_hip_dev_inter_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    W, H = 192, 128
    rng = np.random.default_rng(77)
    refs = [A(rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)), A(np.where(rng.integers(0, 3, W * H * 3 // 2) > 0, 255, 0).astype(np.uint8))]
    pus, k = [], 0
    for by in range(0, H, 64):  # non-overlapping PUs covering the picture: 64, 32, 16, 8 squares and 2:1 shapes
        for bx in range(0, W, 64):
            size = (64, 32, 16, 8)[k % 4]
            k += 1
            for yy in range(by, by + 64, size):
                for xx in range(bx, bx + 64, size):
                    shape = (size, size) if (xx + yy) % 3 else ((size, size // 2) if size > 8 else (8, 8))
                    for part in range(size // shape[1]):
                        mv0 = (int(rng.integers(-70, 71)), int(rng.integers(-70, 71)))
                        mv1 = (int(rng.integers(-70, 71)), int(rng.integers(-70, 71)))
                        mode = int(rng.integers(0, 6))
                        if mode == 0: mv0 = (mv0[0] & ~3, mv0[1] & ~3)          # integer vector: copied, chroma may still be fractional
                        if mode == 1: mv0, mv1 = (mv0[0] & ~7, mv0[1] & ~7), (mv1[0] & ~7, mv1[1] & ~7)  # integer in both planes
                        if mode == 2: mv0 = (-4 * (xx + 40), mv0[1])            # far outside the picture
                        use = ((1, 0), (0, 1), (1, 1), (1, 1))[int(rng.integers(0, 4))]
                        pus.append((xx, yy + part * shape[1], shape[0], shape[1], mv0, mv1, use[0], use[1]))
    arr = (McPu * len(pus))()
    for i, (x, y, w, h, mv0, mv1, u0, u1) in enumerate(pus):
        arr[i].x, arr[i].y, arr[i].w, arr[i].h = x, y, w, h
        arr[i].mv[0][0], arr[i].mv[0][1], arr[i].mv[1][0], arr[i].mv[1][1] = mv0[0], mv0[1], mv1[0], mv1[1]
        arr[i].use[0], arr[i].use[1] = u0, u1
    d0, d1 = dev.put(refs[0]), dev.put(refs[1])
    d_pred = dev.put(np.zeros(W * H * 3 // 2, np.uint8))
    d_pus = dev.empty(C.sizeof(arr))
    lib.kvz_hip_dev_upload(d_pus, C.addressof(arr), C.sizeof(arr))
    lib.kvz_hip_dev_inter_pred(d0, d1, d_pred, W, H, d_pus, len(pus), 64)
    got = dev.get(d_pred, (W * H * 3 // 2,), np.uint8)
    want = oracle_inter_pred(oracle, refs, W, H, pus)
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    dev.free(d0, d1, d_pred, d_pus)


def test_oracle_inter_pred_composition_runs(oracle):
    """CPU smoke of the branch-following composition (the checker of the GPU test above)"""
    W, H = 64, 64
    rng = np.random.default_rng(1)
    refs = [A(rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)) for _ in range(2)]
    pus = [(0, 0, 32, 32, (5, -3), (0, 0), 1, 0), (32, 0, 32, 32, (8, 16), (-9, 2), 1, 1), (0, 32, 64, 32, (4, 8), (12, -4), 1, 1)]
    pred = oracle_inter_pred(oracle, refs, W, H, pus)
    assert pred[:W * 32].any()
    # an integer vector in both planes is a plain (clamped) copy
    assert np.array_equal(pred[:W * H].reshape(H, W)[0:32, 32:64] * 0 + 1, np.ones((32, 32), np.uint8))


LF = np.array([[0, 0, 0, 64, 0, 0, 0, 0], [-1, 4, -10, 58, 17, -5, 1, 0], [-1, 4, -11, 40, 40, -11, 4, -1], [0, 1, -5, 17, 58, -10, 4, -1]])  # filter.c:66-72
CF = np.array([[0, 64, 0, 0], [-2, 58, 10, -2], [-4, 54, 16, -2], [-6, 46, 28, -4], [-4, 36, 36, -4], [-4, 28, 46, -6], [-2, 16, 54, -4], [-2, 10, 58, -2]])  # filter.c:74-84


def uniform_inter_pred(refs, W, H, pus):
    """what the device kernel computes: ONE expression for every branch of the reference (identity taps for integer vectors)"""
    pred = np.zeros(W * H * 3 // 2, np.uint8)
    for (x, y, w, h, mv0, mv1, u0, u1) in pus:
        for pi in range(3):
            sh = 1 if pi else 0
            fw, fh, off = W >> sh, H >> sh, 0 if pi == 0 else (W * H if pi == 1 else W * H * 5 // 4)
            pw, ph, taps, before = w >> sh, h >> sh, 4 if pi else 8, 1 if pi else 3
            vals = []
            for use, mv, ref in ((u0, mv0, refs[0]), (u1, mv1, refs[1])):
                if not use:
                    continue
                plane = ref[off:off + fw * fh].reshape(fh, fw).astype(np.int64)
                X0, Y0 = (x >> sh) + (mv[0] >> (2 + sh)) - before, (y >> sh) + (mv[1] >> (2 + sh)) - before
                win = plane[np.ix_(np.clip(np.arange(Y0, Y0 + ph + taps - 1), 0, fh - 1), np.clip(np.arange(X0, X0 + pw + taps - 1), 0, fw - 1))]
                hf, vf = (CF[mv[0] & 7], CF[mv[1] & 7]) if pi else (LF[mv[0] & 3], LF[mv[1] & 3])
                g = sum(hf[k] * win[:, k:k + pw] for k in range(taps)).astype(np.int16).astype(np.int64)
                vals.append((sum(vf[k] * g[k:k + ph, :] for k in range(taps)) >> 6).astype(np.int16).astype(np.int64))
            res = np.clip((vals[0] + 32) >> 6, 0, 255) if len(vals) == 1 else np.clip((vals[0] + vals[1] + 64) >> 7, 0, 255)
            pred[off:off + fw * fh].reshape(fh, fw)[y >> sh:(y >> sh) + ph, x >> sh:(x >> sh) + pw] = res
    return pred


def test_one_expression_covers_every_reference_branch(oracle):
    """copy vs filter, pixel vs 14-bit bipred operands, picture-edge replication: the reference's branches (through the oracle) == identity-tap filtering"""
    W, H = 192, 128
    rng = np.random.default_rng(77)
    refs = [A(rng.integers(0, 256, W * H * 3 // 2, dtype=np.uint8)), A(np.where(rng.integers(0, 3, W * H * 3 // 2) > 0, 255, 0).astype(np.uint8))]
    pus = [(0, 0, 64, 64, (5, -3), (0, 0), 1, 0), (64, 0, 32, 32, (8, 16), (-9, 2), 1, 1), (0, 64, 64, 32, (4, 8), (12, -4), 1, 1), (128, 64, 16, 16, (-300, 7), (2, 2), 0, 1),
           (64, 64, 8, 8, (4, 0), (0, 4), 1, 1), (96, 64, 16, 8, (-13, 70), (66, -70), 1, 1), (128, 0, 64, 64, (16, -24), (0, 0), 1, 1), (112, 64, 16, 16, (700, 700), (-3, 1), 1, 1)]
    assert np.array_equal(oracle_inter_pred(oracle, refs, W, H, pus), uniform_inter_pred(refs, W, H, pus))
