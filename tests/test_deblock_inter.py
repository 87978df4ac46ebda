"""Deblocking of pictures with inter CUs (filter.c:405-493 boundary strengths, :225-257 prediction-unit edges): the oracle against the compiled
reference's kvz_filter_deblock_lcu on random CU quadtrees with random partitionings, motion vectors, reference indices and coded-block flags
(P and B slice rules), and -- under -m gpu -- kvz_hip_dev_deblock_frames_inter against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import deblock_common as dc
import flatapi


class CuDbk(C.Structure):
    _fields_ = [("type", C.c_uint8), ("depth", C.c_uint8), ("tr_depth", C.c_uint8), ("part_size", C.c_uint8), ("cbf_y", C.c_uint8), ("mv_dir", C.c_uint8),
                ("mv_ref", C.c_int8 * 2), ("ref_id", C.c_int16 * 2), ("mv", (C.c_int16 * 2) * 2)]


PART_SIZES = {0: [(0, 0, 4, 4)], 1: [(0, 0, 4, 2), (0, 2, 4, 2)], 2: [(0, 0, 2, 4), (2, 0, 2, 4)], 3: [(0, 0, 2, 2), (2, 0, 2, 2), (0, 2, 2, 2), (2, 2, 2, 2)],
              4: [(0, 0, 4, 1), (0, 1, 4, 3)], 5: [(0, 0, 4, 3), (0, 3, 4, 1)], 6: [(0, 0, 1, 4), (1, 0, 3, 4)], 7: [(0, 0, 3, 4), (3, 0, 1, 4)]}  # cu.c:63-90, in CU quarters


def random_inter_picture(width, height, rng, slice_b, intra_share=0.15):
    """(records per 4x4 unit, ref_LX tables): CUs from a random quadtree; inter CUs get a partitioning (SMP / AMP where the CU is large enough), per-PU motion
    drawn from a small pool (so that equal / near-equal vectors across edges happen), references from consistent (list, index) -> picture tables"""
    w4, h4 = width // 4, height // 4
    info = (CuDbk * (w4 * h4))()
    ref_lx = [[int(v) for v in rng.permutation(6)[:3]], [int(v) for v in rng.permutation(6)[:3]]]
    if rng.random() < 0.5:
        ref_lx[1][0] = ref_lx[0][0]  # the same picture in both lists: the "same L0 & L1" rule of B slices
    depth = dc.random_depth_map(width, height, rng)
    pool = [(int(rng.integers(-40, 41)), int(rng.integers(-40, 41))) for _ in range(5)]

    def motion():
        base = pool[int(rng.integers(0, len(pool)))]
        return (base[0] + int(rng.integers(-4, 5)), base[1] + int(rng.integers(-4, 5)))

    for cy in range(0, height, 8):
        for cx in range(0, width, 8):
            d = int(depth[cy // 8, cx // 8])
            cw = 64 >> d
            if cx % cw or cy % cw:
                continue  # not a CU origin
            intra = rng.random() < intra_share
            ps = 0 if intra else int(rng.choice([0, 0, 1, 2] + ([4, 5, 6, 7] if cw >= 16 else []) + ([3] if cw == 8 else [])))
            tr_depth = min(d + int(rng.integers(0, 2)), 3) if (not intra or d > 0) else 1
            if d == 0:
                tr_depth = 1
            cbf = int(rng.random() < 0.5)
            for (ox, oy, pw, ph) in PART_SIZES[ps]:
                mv_dir = 1 if not slice_b else int(rng.choice([1, 2, 3]))
                mv = [motion(), motion()]
                mref = [int(rng.integers(0, 3)), int(rng.integers(0, 3))]
                for uy in range(cy + oy * cw // 4, min(height, cy + (oy + ph) * cw // 4), 4):
                    for ux in range(cx + ox * cw // 4, min(width, cx + (ox + pw) * cw // 4), 4):
                        r = info[(uy // 4) * w4 + ux // 4]
                        r.type, r.depth, r.tr_depth, r.part_size, r.cbf_y = (1 if intra else 2), d, tr_depth, ps, cbf
                        if not intra:
                            r.mv_dir = mv_dir
                            for l in range(2):
                                r.mv_ref[l], r.ref_id[l] = mref[l], ref_lx[l][mref[l]]
                                r.mv[l][0], r.mv[l][1] = mv[l]
    return info


def run(func, w, h, qp, b_off, t_off, frame, info, slice_b):
    out = frame.copy()
    n, c = w * h, w * h // 4
    func.restype = None
    func.argtypes = [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_int]
    func(w, h, qp, b_off, t_off, out.ctypes.data, out.ctypes.data + n, out.ctypes.data + n + c, C.addressof(info), slice_b)
    return out


CASES = [(64, 64, 1, 0), (192, 136, 2, 0), (192, 136, 3, 1), (416, 240, 4, 1), (416, 240, 5, 0)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}-{'B' if c[3] else 'P'}")
def test_oracle_inter_deblock_equals_compiled_reference(oracle, case):
    if not os.path.exists(flatapi.refshim_path()):
        pytest.skip("oracle/_ref not built")
    ref = flatapi.load_ref(0)
    w, h, seed, slice_b = case
    rng = np.random.default_rng(seed)
    for kind in ("smooth", "steps", "noise"):
        frame, _ = dc.test_picture(w, h, rng, kind)
        info = random_inter_picture(w, h, rng, slice_b)
        for qp, b_off, t_off in ((27, 0, 0), (37, 1, -2)):
            a = run(oracle.lib.kvz_oracle_deblock_frame_inter, w, h, qp, b_off, t_off, frame, info, slice_b)
            b = run(ref.lib.kvz_ref_deblock_frame_inter, w, h, qp, b_off, t_off, frame, info, slice_b)
            assert np.array_equal(a, b), (kind, qp, np.flatnonzero(a != b)[:8])
            if kind != "noise":
                assert not np.array_equal(a, frame)


def test_inter_rules_matter(oracle):
    """strength 0 / 1 / 2 edges all occur: the inter picture is filtered less than an all-intra one with the same quadtree, but more than not at all"""
    w, h = 192, 136
    rng = np.random.default_rng(9)
    frame, _ = dc.test_picture(w, h, rng, "steps")
    info = random_inter_picture(w, h, rng, 1, intra_share=0.0)
    inter = run(oracle.lib.kvz_oracle_deblock_frame_inter, w, h, 32, 0, 0, frame, info, 1)
    for r in info:
        r.type = 1
    intra = run(oracle.lib.kvz_oracle_deblock_frame_inter, w, h, 32, 0, 0, frame, info, 1)
    changed_inter, changed_intra = int((inter != frame).sum()), int((intra != frame).sum())
    assert 0 < changed_inter < changed_intra


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}-{'B' if c[3] else 'P'}")
def test_hip_inter_deblock_equals_oracle(oracle, case):
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    lib.kvz_hip_dev_deblock_frames_inter.restype = None
    lib.kvz_hip_dev_deblock_frames_inter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    w, h, seed, slice_b = case
    rng = np.random.default_rng(seed)
    frames, infos = [], []
    for kind in ("smooth", "steps", "noise"):
        frames.append(dc.test_picture(w, h, rng, kind)[0])
        infos.append(random_inter_picture(w, h, rng, slice_b))
    for qp, b_off, t_off in ((27, 0, 0), (37, 1, -2)):
        d_fr = dev.put(np.stack(frames))
        blob = b"".join(bytes(i) for i in infos)
        d_info = dev.empty(len(blob))
        lib.kvz_hip_dev_upload(d_info, blob, len(blob))
        lib.kvz_hip_dev_deblock_frames_inter(d_fr, w, h, len(frames), d_info, qp, b_off, t_off, slice_b)
        got = dev.get(d_fr, (len(frames), w * h * 3 // 2), np.uint8)
        for i in range(len(frames)):
            want = run(oracle.lib.kvz_oracle_deblock_frame_inter, w, h, qp, b_off, t_off, frames[i], infos[i], slice_b)
            assert np.array_equal(got[i], want), (i, qp, np.flatnonzero(got[i] != want)[:8])
        dev.free(d_fr, d_info)
