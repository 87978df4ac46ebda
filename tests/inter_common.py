"""Shared by the inter-oracle tests and tests/golden/make_golden.py --inter: synthetic clips with real motion, the low-delay sequence oracle
(oracle/kvz_oracle_inter.inc through ctypes) and the compiled reference encoder run on the same clip (oracle/_ref/kvazaar_ref with the
oracle/ref_cudump.c interposer recording the CU decisions).  TEST INFRASTRUCTURE -- never imported by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

import ctu_common as cc
import flatapi

CU_DTYPE = np.dtype([("type", "u1"), ("depth", "u1"), ("mode", "u1"), ("tr_depth", "u1"), ("cbf", "<u2"), ("skipped", "u1"), ("merged", "u1"), ("merge_idx", "u1"),
                     ("mv_dir", "u1"), ("mv_ref", "u1", (2,)), ("mv_cand", "u1", (2,)), ("mv", "<i2", (2, 2))], align=True)
assert CU_DTYPE.itemsize == 22


class LowdelayCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("qp", "gop_len", "gop_depth", "intra_period", "fme_level", "pu_depth_inter_max", "sao", "deblock", "mv_constraint", "no_wpp", "ra8_qp_model", "fast_residual_cost")]


PRESETS = {"faster": dict(fme_level=4, pu_depth_inter_max=3, sao=1, fast_residual_cost=0), "veryfast": dict(fme_level=2, pu_depth_inter_max=3, sao=1, fast_residual_cost=28),
           "superfast": dict(fme_level=2, pu_depth_inter_max=2, sao=1, fast_residual_cost=28), "ultrafast": dict(fme_level=0, pu_depth_inter_max=2, sao=0, fast_residual_cost=28)}


def clip(w, h, n, seed, noise=1.5, pan=(1.25, -0.5)):
    """n frames (Y|U|V bytes each) of a textured background under a global sub-pel pan, rectangles moving on their own, one object that appears in
    frame 2 (nothing to predict it from) and sensor noise: skipped, merged, AMVP and intra CUs all occur"""
    rng = np.random.default_rng(seed)
    H2, W2 = h + 64, w + 64  # the pan wraps around (np.roll): large pans bring in unrelated content, which is fine
    yy, xx = np.mgrid[0:H2, 0:W2].astype(np.float64)
    base = 128 + 45 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 30 * np.sin((xx + 2 * yy) / 31.0) + 12 * rng.standard_normal((H2, W2)).cumsum(axis=1) / 6
    cb = 128 + 40 * np.sin(xx[::2, ::2] / 27.0) + 20 * np.cos(yy[::2, ::2] / 19.0)
    cr = 128 + 40 * np.cos((xx[::2, ::2] - yy[::2, ::2]) / 33.0)
    rects = [(int(rng.integers(0, w - 40)), int(rng.integers(0, h - 40)), int(rng.integers(16, 40)), int(rng.integers(16, 40)), float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2)),
              int(rng.integers(40, 220))) for _ in range(4)]
    out = []

    def shift(p, dx, dy):  # bilinear sub-pel shift of a padded plane
        ix, iy = int(np.floor(dx)), int(np.floor(dy))
        fx, fy = dx - ix, dy - iy
        q = np.roll(p, (-iy, -ix), (0, 1))
        q = (1 - fx) * q + fx * np.roll(q, -1, 1)
        return (1 - fy) * q + fy * np.roll(q, -1, 0)

    for i in range(n):
        dx, dy = pan[0] * i, pan[1] * i
        Y = shift(base, dx, dy)[32:32 + h, 32:32 + w].copy()
        U = shift(cb, dx / 2, dy / 2)[16:16 + h // 2, 16:16 + w // 2].copy()
        V = shift(cr, dx / 2, dy / 2)[16:16 + h // 2, 16:16 + w // 2].copy()
        for (rx, ry, rw, rh, vx, vy, lum) in rects:
            x0, y0 = int(round(rx + vx * i)) % max(1, w - rw), int(round(ry + vy * i)) % max(1, h - rh)
            Y[y0:y0 + rh, x0:x0 + rw] = lum + 10 * np.sin(np.arange(rw) / 3.0)[None, :]
            U[y0 // 2:(y0 + rh) // 2, x0 // 2:(x0 + rw) // 2] = 90
        if i >= 2:
            Y[h // 3:h // 3 + 24, w // 2:w // 2 + 24] = 30 + 8 * ((np.arange(24)[:, None] + np.arange(24)[None, :]) % 5)
            V[h // 6:h // 6 + 12, w // 4:w // 4 + 12] = 200
        Y = Y + rng.normal(0, noise, Y.shape)
        out.append(np.concatenate([np.clip(np.rint(p), 0, 255).astype(np.uint8).reshape(-1) for p in (Y, U, V)]))
    return out


def oracle_encode(oracle, w, h, frames, qp, preset="veryfast", deblock=True, sao=None, mv_constraint=False, gop=(4, 3), no_wpp=False, overrides=None):
    """-> (rec_search [n, fs], rec_final [n, fs], cu [n, h/4, w/4] of CU_DTYPE, qps); overrides: search options that differ from the preset's (fme_level, fast_residual_cost ...)"""
    p = dict(PRESETS[preset])
    p.update(overrides or {})
    if sao is not None:
        p["sao"] = int(sao)
    cfg = LowdelayCfg(qp=qp, gop_len=gop[0], gop_depth=gop[1], intra_period=64, deblock=int(deblock), mv_constraint=int(mv_constraint), no_wpp=int(no_wpp), ra8_qp_model=1, **p)
    mc = cc.model_constants()
    fb = (C.c_float * 128)(*mc["entropy_fbits"])
    wts = (C.c_uint64 * 52)(*[int(mc["coeff_weights"][str(q)]) for q in range(52)])
    n, fs = len(frames), w * h * 3 // 2
    src = np.ascontiguousarray(np.concatenate(frames))
    rs, rf = np.zeros(n * fs, np.uint8), np.zeros(n * fs, np.uint8)
    cu = np.zeros(n * (w // 4) * (h // 4), CU_DTYPE)
    qps = np.zeros(n, np.int32)
    f = oracle.lib.kvz_oracle_lowdelay_encode
    f.restype = None
    f.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 5
    f(C.addressof(cfg), C.addressof(fb), C.addressof(wts), w, h, n, src.ctypes.data, rs.ctypes.data, rf.ctypes.data, cu.ctypes.data, qps.ctypes.data)
    return rs.reshape(n, fs), rf.reshape(n, fs), cu.reshape(n, h // 4, w // 4), qps


def reference_encode(w, h, frames, qp, workdir, preset="veryfast", deblock=True, sao=None, owf=0, threads=0, gop="lp-g4d3t1", cu=True, extra=()):
    """oracle/_ref/kvazaar_ref on the clip -> (--debug reconstruction [n, fs], cu [n, h/4, w/4] of CU_DTYPE or None).  The CU records are only taken with
    threads == 0 (LCUs then reach kvz_encode_coding_tree picture by picture, in raster order)"""
    exe = os.path.join(flatapi.ROOT, "oracle", "_ref", "kvazaar_ref")
    src, rec, dump = os.path.join(workdir, "in.yuv"), os.path.join(workdir, "rec.yuv"), os.path.join(workdir, "cu.txt")
    with open(src, "wb") as f:
        f.write(b"".join(fr.tobytes() for fr in frames))
    cmd = [exe, "-i", src, "--input-res", f"{w}x{h}", "--preset", preset, "--gop", gop, "-q", str(qp), "--debug", rec, "-o", os.path.join(workdir, "out.hevc"),
           "--threads", str(threads), "--owf", str(owf)] + list(extra)
    if not deblock:
        cmd.append("--no-deblock")
    if sao is not None:
        cmd += ["--sao", "full" if sao else "off"]
    env = dict(os.environ)
    if cu:
        assert threads == 0
        if os.path.exists(dump):
            os.remove(dump)
        env.update(LD_PRELOAD=os.path.join(flatapi.ROOT, "oracle", "_ref", "libkvz_cudump.so"), KVZ_CUDUMP=dump, KVZ_CUDUMP_INTER="1")
    subprocess.run(cmd, check=True, capture_output=True, env=env)
    n, fs = len(frames), w * h * 3 // 2
    recon = np.fromfile(rec, dtype=np.uint8).reshape(n, fs)
    if not cu:
        return recon, None
    rows = np.loadtxt(dump, dtype=np.int64).reshape(n, -1, 20)
    out = np.zeros((n, h // 4, w // 4), CU_DTYPE)
    for f in range(n):
        r = rows[f]
        yy, xx = r[:, 1] // 4, r[:, 0] // 4
        for k, name in ((2, "type"), (3, "depth"), (5, "tr_depth"), (6, "skipped"), (7, "merged"), (8, "merge_idx"), (9, "cbf"), (10, "mode"), (11, "mv_dir")):
            out[name][f, yy, xx] = r[:, k]
        out["mv"][f, yy, xx, 0, 0], out["mv"][f, yy, xx, 0, 1], out["mv"][f, yy, xx, 1, 0], out["mv"][f, yy, xx, 1, 1] = r[:, 12], r[:, 13], r[:, 14], r[:, 15]
        out["mv_ref"][f, yy, xx, 0], out["mv_ref"][f, yy, xx, 1] = r[:, 16], r[:, 17]
        out["mv_cand"][f, yy, xx, 0], out["mv_cand"][f, yy, xx, 1] = r[:, 18], r[:, 19]
    return recon, out


def first_difference(a, b, fields=("type", "depth", "skipped", "merged", "merge_idx", "mv_dir", "mv", "mv_ref", "mv_cand", "mode")):
    """the first CU (picture, then raster order of the 4x4 units inside LCUs in coding order) where two CU maps differ, or None"""
    n, h4, w4 = a.shape
    for f in range(n):
        for ly in range(0, h4, 16):
            for lx in range(0, w4, 16):
                for name in fields:
                    x, y = a[name][f, ly:ly + 16, lx:lx + 16], b[name][f, ly:ly + 16, lx:lx + 16]
                    if name == "mode":
                        m = (a["type"][f, ly:ly + 16, lx:lx + 16] == 1)
                        x, y = np.where(m, x, 0), np.where(m, y, 0)
                    if name in ("mv_dir", "mv", "mv_ref", "mv_cand", "skipped", "merged"):
                        m = (a["type"][f, ly:ly + 16, lx:lx + 16] == 2)
                        m = m.reshape(m.shape + (1,) * (x.ndim - 2))
                        x, y = np.where(m, x, 0), np.where(m, y, 0)
                    if name in ("merge_idx",):
                        m = (a["merged"][f, ly:ly + 16, lx:lx + 16] | a["skipped"][f, ly:ly + 16, lx:lx + 16]) > 0
                        x, y = np.where(m, x, 0), np.where(m, y, 0)
                    if not np.array_equal(x, y):
                        d = np.argwhere((x != y).reshape(x.shape[0], x.shape[1], -1).any(axis=2))
                        yy, xx = d[0]
                        return dict(frame=f, lcu=(lx // 16, ly // 16), field=name, x=4 * (lx + xx), y=4 * (ly + yy), ours=a[f, ly + yy, lx + xx], ref=b[f, ly + yy, lx + xx])
    return None


# (name, width, height, frames, qp, preset, deblock, sao, owf, clip) -- clip = ("motion", seed, noise, pan) of clip() above or ("synth", seed, kind) of kvazaar_amd/synth.py.
# tests/golden/inter_recon.json holds, per case, the REFERENCE encoder's digests (reconstruction per picture, CU decisions per picture, its bitstream's md5)
CASES = [
    ("pan", 200, 136, 4, 22, "veryfast", 1, 1, 0, ("motion", 5, 1.5, (1.25, -0.5))),
    ("noisy-qp27", 200, 136, 4, 27, "veryfast", 1, 1, 0, ("motion", 6, 3.0, (-2.0, 1.75))),
    ("cabac-coeff-cost-qp32", 264, 200, 4, 32, "veryfast", 1, 1, 0, ("motion", 7, 1.0, (0.5, 0.25))),
    ("fast-pan-owf-qp37", 264, 200, 4, 37, "veryfast", 1, 1, 2, ("motion", 8, 2.0, (4.0, 9.0))),
    ("ultrafast", 416, 240, 5, 22, "ultrafast", 1, 0, 0, ("motion", 9, 1.5, (1.25, -0.5))),
    ("ultrafast-fast-pan-owf-qp30", 416, 240, 5, 30, "ultrafast", 1, 0, 2, ("motion", 10, 1.5, (-6.0, 13.0))),
    ("vertical-pan-owf", 320, 320, 5, 22, "veryfast", 1, 1, 2, ("motion", 11, 0.5, (3.0, 21.0))),
    ("static-qp17", 320, 320, 5, 17, "veryfast", 1, 1, 0, ("motion", 12, 0.0, (0.0, 0.0))),
    ("two-gops", 128, 64, 9, 24, "veryfast", 1, 1, 0, ("motion", 13, 1.0, (-1.0, 0.5))),
    ("no-loop-filters", 416, 240, 4, 22, "veryfast", 0, 0, 0, ("motion", 3, 1.5, (1.25, -0.5))),
    ("deblock-only", 416, 240, 4, 22, "veryfast", 1, 0, 0, ("motion", 3, 1.5, (1.25, -0.5))),
    ("faster-pan", 200, 136, 4, 22, "faster", 1, 1, 0, ("motion", 5, 1.5, (1.25, -0.5))),            # `faster`: quarter-sample motion search, CABAC coefficient cost at every QP
    ("faster-qp32", 264, 200, 4, 32, "faster", 1, 1, 0, ("motion", 7, 1.0, (0.5, 0.25))),
    ("faster-owf-qp27", 416, 240, 5, 27, "faster", 1, 1, 2, ("motion", 9, 1.5, (1.25, -0.5))),
    ("ultrafast-8mod16", 200, 136, 4, 25, "ultrafast", 1, 0, 2, ("motion", 14, 1.5, (2.5, -1.25))),  # width and height 8 mod 16: 8x8 CUs at the edges may be inter although pu-depth-inter stops at 16x16 (search.c:702-713)
    ("superfast-8mod16-qp33", 136, 200, 3, 33, "superfast", 1, 1, 0, ("motion", 15, 1.0, (-1.5, 3.0))),
    ("survey-416x240", 416, 240, 8, 22, "veryfast", 1, 1, 2, ("synth", 1234, "small")),    # bitstream md5 1e7a8165...
    ("survey-1080p", 1920, 1080, 4, 22, "veryfast", 1, 1, 2, ("synth", 1, "large")),
    ("baseline-c4-2160p", 3840, 2160, 4, 22, "veryfast", 1, 1, 2, ("synth", 2, "large")),    # BASELINE config 4 at its own size
]


# the cases whose slice data is pinned (tests/golden/entropy_inter.json): picture QPs on both sides of fast-residual-cost 28, SAO on / off, the wavefront MV restriction, `faster`
ENTROPY_CASES = ["pan", "ultrafast", "vertical-pan-owf", "static-qp17", "two-gops", "no-loop-filters", "deblock-only", "survey-416x240", "noisy-qp27", "cabac-coeff-cost-qp32",
                 "faster-pan", "faster-qp32", "ultrafast-8mod16"]

# ... and BASELINE config 4 at its own size: fixture entry only (bench.py's leg and the GPU test check the device against it)
ENTROPY_BENCH_CASES = ["baseline-c4-2160p"]


def case_frames(case):
    name, w, h, n, qp, preset, dbk, sao, owf, src = case
    if src[0] == "motion":
        return clip(w, h, n, src[1], src[2], src[3])
    import kvazaar_amd.synth as synth
    return [np.concatenate([p.reshape(-1) for p in f]) for f in synth.frames(w, h, n, src[1], src[2])]


DBK_DTYPE = np.dtype([("type", "u1"), ("depth", "u1"), ("tr_depth", "u1"), ("part_size", "u1"), ("cbf_y", "u1"), ("mv_dir", "u1"), ("mv_ref", "i1", (2,)), ("ref_id", "<i2", (2,)),
                      ("mv", "<i2", (2, 2))], align=True)  # kvz_hip_cu_dbk (include/kvz_hip_dev.h)


def cu_dbk_records(cu):
    """what the deblocking filter reads of CU records (kvz_hip_dev_cu_dbk_from_info, kvz_dev.hpp dev_cu_dbk_kernel), on the host"""
    assert DBK_DTYPE.itemsize == 20
    flat = np.ascontiguousarray(cu).reshape(-1)
    out = np.zeros(flat.size, DBK_DTYPE)
    masks = np.array([0x1f, 0x0f, 0x07, 0x03, 0x1], np.uint16)
    out["type"], out["depth"], out["tr_depth"] = flat["type"], flat["depth"], flat["tr_depth"]
    out["cbf_y"] = (flat["cbf"] & masks[np.minimum(flat["tr_depth"], 4)]) != 0
    inter = flat["type"] == 2
    out["mv_dir"] = np.where(inter, flat["mv_dir"], 0)
    for l in range(2):
        used = inter & ((flat["mv_dir"] >> l) & 1 > 0)
        out["mv_ref"][:, l] = np.where(used, flat["mv_ref"][:, l].astype(np.int8), 0)
        out["mv"][:, l, 0] = np.where(inter, flat["mv"][:, l, 0], 0)
        out["mv"][:, l, 1] = np.where(inter, flat["mv"][:, l, 1], 0)
    return out


def cu_bytes(cu):
    """the decisions of one picture as bytes, motion and flags only where they mean something"""
    inter = cu["type"] == 2
    coded = inter & (cu["merged"] == 0) & (cu["skipped"] == 0)
    parts = [cu["type"], cu["depth"], np.where(cu["type"] == 1, cu["mode"], 0), np.where(inter, cu["skipped"], 0), np.where(inter, cu["merged"], 0),
             np.where(inter & ((cu["merged"] | cu["skipped"]) > 0), cu["merge_idx"], 0), np.where(inter, cu["mv_dir"], 0)]
    for l in range(2):
        used = inter & ((cu["mv_dir"] >> l) & 1 > 0)
        parts += [np.where(used, cu["mv"][..., l, 0], 0).astype("<i2"), np.where(used, cu["mv"][..., l, 1], 0).astype("<i2"), np.where(coded & used, cu["mv_cand"][..., l], 0)]
    return b"".join(np.ascontiguousarray(p).tobytes() for p in parts)


def digests(rec, cu):
    import hashlib
    return {"rec": [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in rec], "cu": [hashlib.sha256(cu_bytes(c)).hexdigest()[:24] for c in cu]}


# ---- the motion search of single PUs (kvz_hip_dev_pu_search's contract, include/kvz_hip_dev.h) ----
ME_PU = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "<i2"), ("h", "<i2"), ("mv_cand", "<i2", (2, 2)), ("start_mv", "<i2", (2,)), ("has_start", "u1"), ("num_merge", "u1"),
                  ("merge_dir", "u1", (5,)), ("reserved", "u1"), ("merge_mv", "<i2", (5, 2))], align=True)
ME_RESULT = np.dtype([("mv", "<i4", (2,)), ("mvp", "<i4"), ("valid", "<i4"), ("cost", "<f8"), ("bits", "<f8"), ("frac_mv", "<i4", (2,)), ("frac_mvp", "<i4"), ("frac_valid", "<i4"),
                      ("frac_cost", "<f8"), ("frac_bits", "<f8")], align=True)
assert ME_PU.itemsize == 48 and ME_RESULT.itemsize == 64


class MeParams(C.Structure):
    _fields_ = [("lambda_sqrt", C.c_double), ("mv_constraint", C.c_int32), ("sao", C.c_int32), ("deblock", C.c_int32), ("fme_level", C.c_int32)]


def lambda_sqrt(qp):
    return float(np.sqrt(0.57 * 2.0 ** ((qp - 12) / 3.0)))


def oracle_pu_search(oracle, cur, ref, w, h, pus, params):
    out = np.zeros(len(pus), ME_RESULT)
    f = oracle.lib.kvz_oracle_pu_motion_search
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    cur, ref, pus = np.ascontiguousarray(cur), np.ascontiguousarray(ref), np.ascontiguousarray(pus)
    f(cur.ctypes.data, ref.ctypes.data, w, h, pus.ctypes.data, len(pus), C.addressof(params), out.ctypes.data)
    return out


def traced_encode(oracle, case, capacity=400000):
    """the sequence oracle on a CASES entry with the motion-search recorder on -> (frames, rec_final, qps, pus, results, poc) of every search it ran"""
    name, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames = case_frames(case)
    pus, res, poc = np.zeros(capacity, ME_PU), np.zeros(capacity, ME_RESULT), np.zeros(capacity, np.int32)
    tr = oracle.lib.kvz_oracle_me_trace
    tr.restype = None
    tr.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    tr(pus.ctypes.data, res.ctypes.data, poc.ctypes.data, capacity)
    try:
        rs, rf, cu, qps = oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
        cnt = oracle.lib.kvz_oracle_me_trace_count()
    finally:
        tr(None, None, None, 0)
    assert cnt < capacity
    return frames, rf, qps, pus[:cnt].copy(), res[:cnt].copy(), poc[:cnt].copy()


def me_results_differ(a, b, fme_level):
    """indices where two ME_RESULT arrays differ in a field that is defined"""
    bad = (a["mv"] != b["mv"]).any(axis=1) | (a["valid"] != b["valid"]) | (a["mvp"] != b["mvp"]) | (a["cost"] != b["cost"]) | (a["bits"] != b["bits"])
    if fme_level:
        bad |= a["frac_valid"] != b["frac_valid"]
        both = (a["frac_valid"] != 0) & (b["frac_valid"] != 0)
        bad |= both & ((a["frac_mv"] != b["frac_mv"]).any(axis=1) | (a["frac_mvp"] != b["frac_mvp"]) | (a["frac_cost"] != b["frac_cost"]) | (a["frac_bits"] != b["frac_bits"]))
    return np.flatnonzero(bad)


def oracle_encode_bits(oracle, w, h, frames, qp, preset="veryfast", deblock=True, sao=None, mv_constraint=False, gop=(4, 3), no_wpp=False, overrides=None):
    """kvz_oracle_lowdelay_encode_bits -> [(slice data of the picture, [substream sizes])] per picture"""
    p = dict(PRESETS[preset])
    p.update(overrides or {})
    if sao is not None:
        p["sao"] = int(sao)
    cfg = LowdelayCfg(qp=qp, gop_len=gop[0], gop_depth=gop[1], intra_period=64, deblock=int(deblock), mv_constraint=int(mv_constraint), no_wpp=int(no_wpp), ra8_qp_model=1, **p)
    mc = cc.model_constants()
    fb = (C.c_float * 128)(*mc["entropy_fbits"])
    wts = (C.c_uint64 * 52)(*[int(mc["coeff_weights"][str(q)]) for q in range(52)])
    n, fs = len(frames), w * h * 3 // 2
    src = np.ascontiguousarray(np.concatenate(frames))
    rows = 1 if no_wpp else (h + 63) // 64
    cap = n * (w * h * 4 + 4096)
    data, sizes, offs = np.zeros(cap, np.uint8), np.zeros((n, rows), np.uint32), np.zeros(n + 1, np.uint64)
    f = oracle.lib.kvz_oracle_lowdelay_encode_bits
    f.restype = None
    f.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 4 + [C.c_size_t, C.c_void_p, C.c_void_p]
    f(C.addressof(cfg), C.addressof(fb), C.addressof(wts), w, h, n, src.ctypes.data, None, None, data.ctypes.data, cap, sizes.ctypes.data, offs.ctypes.data)
    return [(data[int(offs[k]):int(offs[k + 1])].tobytes(), [int(v) for v in sizes[k]]) for k in range(n)]


def b_slice_context_states(oracle, qp):
    """the 168 initial context states of a B slice at `qp` in the device coder's numbering (kvz_entropy.hpp: KVZ_HIP_CX_* then KVZ_EB_CX_*) = the oracle's CX numbering"""
    f = oracle.lib.kvz_oracle_b_slice_contexts
    f.restype = None
    f.argtypes = [C.c_int, C.c_void_p]
    out = np.zeros(172, np.uint8)
    f(int(qp), out.ctypes.data)
    return out[:168].copy()


def oracle_sequence_for_entropy(oracle, case):
    """everything the device's B-picture coder needs of a sequence, from the oracle: per picture the final CU records, the levels of every CTU, the SAO decisions
    (kvz_hip_sao_params arrays + merge) and the picture QPs"""
    name, w, h, n, qp, preset, dbk, sao, owf, src = case
    p = dict(PRESETS[preset])
    p["sao"] = int(sao)
    cfg = LowdelayCfg(qp=qp, gop_len=4, gop_depth=3, intra_period=64, deblock=int(dbk), mv_constraint=int(owf > 0), no_wpp=0, ra8_qp_model=1, **p)
    mc = cc.model_constants()
    fb = (C.c_float * 128)(*mc["entropy_fbits"])
    wts = (C.c_uint64 * 52)(*[int(mc["coeff_weights"][str(q)]) for q in range(52)])
    frames = case_frames(case)
    fs, cells, ctus = w * h * 3 // 2, (w // 4) * (h // 4), ((w + 63) // 64) * ((h + 63) // 64)
    src_all = np.ascontiguousarray(np.concatenate(frames))
    cu = np.zeros(n * cells, CU_DTYPE)
    coeff = np.zeros(n * ctus * 6144, np.int16)
    sao_l, sao_c, merge = np.zeros(n * ctus * 15, np.int32), np.zeros(n * ctus * 15, np.int32), np.zeros(n * ctus, np.uint8)
    qps = np.zeros(n, np.int32)
    f = oracle.lib.kvz_oracle_lowdelay_encode_parts
    f.restype = None
    f.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 7
    f(C.addressof(cfg), C.addressof(fb), C.addressof(wts), w, h, n, src_all.ctypes.data, cu.ctypes.data, coeff.ctypes.data, sao_l.ctypes.data, sao_c.ctypes.data,
      merge.ctypes.data, qps.ctypes.data)
    return dict(cu=cu.reshape(n, cells), coeff=coeff.reshape(n, ctus * 6144), sao_luma=sao_l.reshape(n, ctus, 15), sao_chroma=sao_c.reshape(n, ctus, 15),
                merge=merge.reshape(n, ctus), qps=qps, ctus=ctus)
