#!/usr/bin/env python3
"""Micro-benchmarks of the device-resident batch primitives (include/kvz_hip_dev.h) against their rooflines -- SURVEY.md 8(d).

Shapes follow the reference's tests/speed_tests.c (N x N cost functions and transforms on its gradient data), batched over
BATCH_CTUS CTUs' worth of blocks that are resident in HBM before the timed region.  For every kernel one JSON line:
  achieved GB/s = algorithmic bytes (2 N^2 per SAD/SATD block, 4 N^2 per transform block, (4N+2)+N^2 per angular
  block) / mean launch time measured with HIP events on the launch stream; frac = achieved / 8000 GB/s (MI355X HBM3E peak);
  matrix-core transforms additionally report utilisation = blocks/s * 4 N^3 / 5e15 (dense int8 peak) and what they actually issue
  (byte planes, zero padding: mfma_issued_frac).
bench.py remains the headline (CTUs/s); this file is the per-kernel view."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0
I8_DENSE_PEAK = 5.0e15  # int8 MFMA, dense: twice the bf16 / f16 rate (MI355X_MICROARCH.md)


def gradient_blocks(n, count, seed):
    """speed_tests.c init_gradient flavour: per-block offset + horizontal ramp, plus noise on the second operand"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 200, (count, 1), dtype=np.int32) + (np.arange(n * n, dtype=np.int32)[None, :] % n)
    a = np.clip(base, 0, 255).astype(np.uint8)
    b = np.clip(base + rng.integers(-8, 9, (count, n * n), dtype=np.int32), 0, 255).astype(np.uint8)
    return a, b


def time_call(dev, fn, reps, warmup):
    for _ in range(warmup):
        fn()
    dev.lib.kvz_hip_dev_sync()
    dev.lib.kvz_hip_dev_timer_start()
    for _ in range(reps):
        fn()
    return dev.lib.kvz_hip_dev_timer_stop() / reps  # ms per launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch-ctus", type=int, default=2040 * 64, help="CTUs' worth of blocks per launch (speed_tests shapes x this); the default puts >= 1 GB through every cost kernel so that the 256 MB Infinity Cache cannot hold the working set")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()

    import kvazaar_amd
    from kvazaar_amd import dev as devapi
    dev = devapi.Dev(kvazaar_amd.load_library())
    results = []

    def report(name, n, count, ms, bytes_per_block, extra=None):
        gbps = count * bytes_per_block / (ms * 1e-3) / 1e9
        r = {"kernel": name, "n": n, "blocks": count, "ms": round(ms, 4), "bytes_per_block": bytes_per_block,
             "achieved_GBps": round(gbps, 1), "peak_GBps": HBM_PEAK_GBPS, "frac": round(gbps / HBM_PEAK_GBPS, 4), "blocks_per_s": round(count / (ms * 1e-3))}
        if extra:
            r.update(extra)
        results.append(r)
        print(json.dumps(r), flush=True)

    for n in (8, 16, 32, 64):
        count = args.batch_ctus * (64 // n) ** 2
        a, b = gradient_blocks(n, min(count, 4096), n)
        reps_needed = (count + a.shape[0] - 1) // a.shape[0]
        a, b = np.tile(a, (reps_needed, 1))[:count], np.tile(b, (reps_needed, 1))[:count]
        da, db, do = dev.put(a), dev.put(b), dev.empty(4 * count)
        for cost in ("sad", "satd"):
            f = getattr(dev.lib, f"kvz_hip_dev_{cost}_nxn")
            ms = time_call(dev, lambda: f(n, da, db, count, do), args.reps, args.warmup)
            report(f"{cost}_{n}x{n}", n, count, ms, 2 * n * n + 4)
        dev.free(da, db, do)

    for name in ("dct4", "dct8", "dct16", "dct32", "idct4", "idct8", "idct16", "idct32"):
        kind = devapi.TRANSFORM_KINDS[name]
        n = devapi.TRANSFORM_SIZE[kind]
        count = args.batch_ctus * (64 // n) ** 2 // 2
        rng = np.random.default_rng(kind)
        x = np.tile(rng.integers(-255, 256, (min(count, 2048), n * n)).astype(np.int16), ((count + 2047) // 2048, 1))[:count]
        di, dt, do = dev.put(x), dev.empty(x.nbytes), dev.empty(x.nbytes)
        for mfma in (0, 1, 2):
            ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_transform(kind, di, dt, do, count, mfma), args.reps, args.warmup)
            if mfma == 0:
                suffix, extra = "_scalar", {"path": "one lane per coefficient, two passes through HBM"}
            elif (mfma == 1 and n < 16) or (mfma == 2 and n == 16):               # (the vector ALU form is the product's for 4- and 8-point blocks, the A/B form at 16)
                suffix, extra = "_rows", {"path": "vector ALU, one lane per block row: v_dot2_i32_i16 on packed pairs, transposes through LDS"}
            else:
                if mfma == 2 and n == 32:
                    continue                                                       # same kernel as mfma == 1
                suffix = "_mfma"
                per_issue = 32 if n == 32 else 16                                  # edge of the product the blocks ride on
                blocks_per_issue = 1 if n >= 16 else 16 // n
                k_issued = 32                                                      # K of both int8 instructions (16 x 16 x 32 runs half empty for 16-point rows)
                ops = count * 4 * n ** 3 / (ms * 1e-3)                             # 2 products x 2 n^3 multiply-adds' worth of operations
                issued = count / blocks_per_issue * 2 * 2 * 2 * per_issue * per_issue * k_issued / (ms * 1e-3)  # 2 products x 2 byte planes
                extra = {"path": "matrix cores, v_mfma_i32_*_i8 on byte planes" + ("" if n == 32 else ", four blocks per wavefront" if n == 16 else " (small blocks on a block diagonal; A/B path)"),
                         "mfma_algorithmic_frac": round(ops / I8_DENSE_PEAK, 5), "mfma_issued_frac": round(issued / I8_DENSE_PEAK, 5)}
            report(f"{name}{suffix}", n, count, ms, 4 * n * n, extra)
        dev.free(di, dt, do)

    for log2w in (2, 3, 4, 5):
        w = 1 << log2w
        count = args.batch_ctus * (64 // w) ** 2 // 2
        rng = np.random.default_rng(log2w)
        refs = rng.integers(0, 256, (2, count, 2 * w + 1), dtype=np.uint8)
        da, dl, do = dev.put(refs[0]), dev.put(refs[1]), dev.empty(count * w * w)
        ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_angular_pred(log2w, 7, da, dl, count, do), args.reps, args.warmup)
        report(f"angular_{w}x{w}", w, count, ms, (4 * w + 2) + w * w, {"mode": 7, "path": "horizontal (transposed on the way out), positive angle"})
        for mode, path in ((30, "vertical, positive angle"), (22, "vertical, negative angle (projected side reference)"), (14, "horizontal, negative angle")):
            ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_angular_pred(log2w, mode, da, dl, count, do), args.reps, args.warmup)
            report(f"angular_{w}x{w}_mode{mode}", w, count, ms, (4 * w + 2) + w * w, {"mode": mode, "path": path})
        dev.free(da, dl, do)

    # deblocking of whole 1080p frames (kvz_hip_dev_deblock_frames): every sample is read and (for the filtered ones) written once
    # per direction at most; algorithmic bytes = read + write of the picture = 2 * 1.5 * W * H
    import ctypes as C
    w, h = 1920, 1080
    nfr = max(8, args.batch_ctus // 510 // 2)
    rng = np.random.default_rng(7)
    yy, xx = np.mgrid[0:h, 0:w]
    luma = ((xx // 8 + yy // 8) % 2 * 6 + 100 + rng.integers(-1, 2, (h, w))).astype(np.uint8)
    frame = np.concatenate([luma.reshape(-1), np.full(w * h // 2, 128, np.uint8)])
    depth = rng.integers(1, 4, (h // 8, w // 8), dtype=np.uint8)
    dfr, ddp = dev.put(np.tile(frame, (nfr, 1))), dev.put(np.tile(depth.reshape(-1), (nfr, 1)))
    dev.lib.kvz_hip_dev_deblock_frames.restype = None
    dev.lib.kvz_hip_dev_deblock_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_deblock_frames(dfr, w, h, nfr, ddp, 32, 0, 0), args.reps, args.warmup)
    report("deblock_1080p_frame", 0, nfr, ms, 2 * (w * h * 3 // 2), {"path": "4 launches: luma / chroma x vertical / horizontal edges", "fps": round(nfr / (ms * 1e-3))})
    dev.free(dfr, ddp)

    # SAO of whole 1080p frames with random per-CTU parameters (kvz_hip_dev_sao_frames): read + write of the picture
    import sao_common as sc
    nctu = 30 * 17
    lum, chr_ = sc.random_params(rng, nctu, False), sc.random_params(rng, nctu, True)
    dl = dev.put(np.tile(np.frombuffer(bytes(lum), dtype=np.uint8), nfr))
    dch = dev.put(np.tile(np.frombuffer(bytes(chr_), dtype=np.uint8), nfr))
    din, dout2 = dev.put(np.tile(frame, (nfr, 1))), dev.empty(nfr * frame.nbytes)
    dev.lib.kvz_hip_dev_sao_frames.restype = None
    dev.lib.kvz_hip_dev_sao_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_sao_frames(din, dout2, w, h, nfr, dl, dch), args.reps, args.warmup)
    report("sao_1080p_frame", 0, nfr, ms, 2 * (w * h * 3 // 2), {"path": "one workgroup per CTU, 4 samples per lane in packed 16-bit halves; random per-CTU parameters (1/4 none, 1/4 band, 1/2 edge)", "fps": round(nfr / (ms * 1e-3))})
    dev.free(dl, dch)
    for name, typ, cls in (("none", 0, 0), ("band", 1, 0), ("edge_class0", 2, 0), ("edge_class1", 2, 1), ("edge_class2", 2, 2)):
        for arr in (lum, chr_):
            for q in arr:
                q.type, q.eo_class = typ, cls
        dl = dev.put(np.tile(np.frombuffer(bytes(lum), dtype=np.uint8), nfr))
        dch = dev.put(np.tile(np.frombuffer(bytes(chr_), dtype=np.uint8), nfr))
        ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_sao_frames(din, dout2, w, h, nfr, dl, dch), args.reps, args.warmup)
        report(f"sao_1080p_frame_{name}", 0, nfr, ms, 2 * (w * h * 3 // 2), {"path": f"every CTU {name}", "fps": round(nfr / (ms * 1e-3))})
        dev.free(dl, dch)
    # the read + write streaming reference: the runtime's own device-to-device copy of the same 128 pictures
    dev.lib.kvz_hip_dev_copy.restype = None
    dev.lib.kvz_hip_dev_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_copy(dout2, din, nfr * frame.nbytes), args.reps, args.warmup)
    report("copy_d2d_1080p_frame", 0, nfr, ms, 2 * (w * h * 3 // 2), {"path": "hipMemcpyAsync device to device: what a kernel that reads and writes every byte once can reach"})
    dev.free(din, dout2)

    # motion cost surface: every 16x16 block of a 1080p picture, +-16 full search (1089 candidates per block)
    bw, rng_ = 16, 16
    cur = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ref = np.roll(cur, (2, -3), (0, 1))
    xy = np.array([(x, y) for y in range(0, h - bw + 1, bw) for x in range(0, w - bw + 1, bw)], dtype=np.int16)
    side = 2 * rng_ + 1
    dc, dr, dxy, dout = dev.put(cur), dev.put(ref), dev.put(xy), dev.empty(4 * len(xy) * side * side)
    dev.lib.kvz_hip_dev_sad_surface.restype = None
    dev.lib.kvz_hip_dev_sad_surface.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_sad_surface(dc, dr, w, h, bw, rng_, dxy, len(xy), dout), args.reps, args.warmup)
    diffs = len(xy) * side * side * bw * bw / (ms * 1e-3)
    report("sad_surface_16x16_r16", bw, len(xy), ms, bw * bw + (bw + 2 * rng_) ** 2 + 4 * side * side,
           {"path": "LDS-staged window, v_qsad_pk_u16_u8: four candidates x four samples per instruction", "candidates_per_s": round(len(xy) * side * side / (ms * 1e-3)),
            "abs_diffs_per_s": round(diffs), "qsad_issue_frac": round(diffs / (1024 * 64 * 16 / 1.79e-9), 4),
            "note": "roof of the arithmetic: 16 differences per lane and v_qsad_pk_u16_u8, one VOP3 wave64 instruction per 1.79 ns and SIMD (profiles/r02_c_valu_issue.jsonl), 1024 SIMDs; the window is read from HBM once"})
    dev.free(dc, dr, dxy, dout)

    # fractional motion search: every 16x16 (and 32x32) PU of a 1080p picture, both half-pel steps (`veryfast`: fme_level 2) = 9 SATDs per PU;
    # algorithmic bytes: the (w + 8)(h + 8) window + the w h source block read, 17 costs written
    class FmePu(C.Structure):
        _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_int16), ("h", C.c_int16), ("mv_x", C.c_int16), ("mv_y", C.c_int16), ("hpel_x", C.c_int8), ("hpel_y", C.c_int8), ("r", C.c_int16)]

    class McPu(C.Structure):
        _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_int16), ("h", C.c_int16), ("mv", (C.c_int16 * 2) * 2), ("use", C.c_int8 * 2), ("r", C.c_int16)]
    dev.lib.kvz_hip_dev_fme_costs.restype = None
    dev.lib.kvz_hip_dev_fme_costs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev.lib.kvz_hip_dev_inter_pred.restype = None
    dev.lib.kvz_hip_dev_inter_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    nrep = 16  # PU lists of 16 pictures' worth per launch (the same picture pair: the window traffic is what it would be for distinct ones)
    dc, dr = dev.put(cur), dev.put(ref)
    for pw in (16, 32):
        coords = [(x, y) for y in range(0, h - pw + 1, pw) for x in range(0, w - pw + 1, pw)] * nrep
        arr = (FmePu * len(coords))(*[FmePu(x, y, pw, pw, int(rng.integers(-8, 9)), int(rng.integers(-8, 9)), 0, 0, 0) for (x, y) in coords])
        dp, dout = dev.empty(C.sizeof(arr)), dev.empty(len(coords) * 17 * 4)
        dev.lib.kvz_hip_dev_upload(dp, C.addressof(arr), C.sizeof(arr))
        for steps, label in ((3, "hpel"), (15, "hpel+qpel")):
            ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_fme_costs(dc, dr, w, h, dp, len(coords), pw, steps, dout), args.reps, args.warmup)
            ncand = 1 + 4 * bin(steps).count("1")
            report(f"fme_costs_{pw}x{pw}_{label}", pw, len(coords), ms, (pw + 8) ** 2 + pw * pw + 4 * 17,
                   {"path": "one workgroup per PU: window in LDS, shared 14-bit horizontal intermediates, planes scored in LDS", "candidates_per_s": round(len(coords) * ncand / (ms * 1e-3)),
                    "pictures_per_s": round(nrep / (ms * 1e-3), 1)})
        dev.free(dp, dout)
    # motion-compensated prediction: every 16x16 PU of a 1080p picture, quarter-pel vectors; uni- and bi-prediction.  Bytes per PU and list:
    # (w + 7)(h + 7) + 2 (w/2 + 3)(h/2 + 3) read, 1.5 w h written once
    yuv = np.concatenate([cur.reshape(-1), rng.integers(0, 256, w * h // 2, dtype=np.uint8)])
    d0, d1, dpred = dev.put(yuv), dev.put(np.roll(yuv, 7)), dev.empty(yuv.nbytes)
    for bi in (0, 1):
        coords = [(x, y) for y in range(0, h - 15, 16) for x in range(0, w - 15, 16)] * nrep
        arr = (McPu * len(coords))()
        for i, (x, y) in enumerate(coords):
            arr[i].x, arr[i].y, arr[i].w, arr[i].h = x, y, 16, 16
            for l in range(2):
                arr[i].mv[l][0], arr[i].mv[l][1] = int(rng.integers(-40, 41)), int(rng.integers(-40, 41))
            arr[i].use[0], arr[i].use[1] = 1, bi
        dp = dev.empty(C.sizeof(arr))
        dev.lib.kvz_hip_dev_upload(dp, C.addressof(arr), C.sizeof(arr))
        ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_inter_pred(d0, d1, dpred, w, h, dp, len(coords), 16), args.reps, args.warmup)
        report("inter_pred_16x16_" + ("bi" if bi else "uni"), 16, len(coords), ms, (1 + bi) * (23 * 23 + 2 * 11 * 11) + 384,
               {"path": "one wavefront per PU (four per workgroup): dword window in LDS, v_dot4 horizontal / v_dot2 vertical 8-tap, 4-tap chroma, bipred average", "pictures_per_s": round(nrep / (ms * 1e-3), 1)})
        dev.free(dp)
    dev.free(dc, dr, d0, d1, dpred)

    # the motion search of a PU, whole (kvz_hip_dev_pu_search: starting points, early termination, hexagon search, the two half-pel steps or the SATD re-pricing):
    # every 8x8 / 16x16 / 32x32 PU of a 1080p picture on smooth moving content (a search has a gradient to follow), AMVP predictors and merge candidates near
    # the true motion, the co-located motion one sample off.  Algorithmic bytes per PU (SURVEY.md 8d's accounting: what must be read once): the source block,
    # the (w + 8)^2 window around the result, 64 B result; the ~25-35 probes re-read reference samples that sit in L2
    class MePu(C.Structure):
        _fields_ = [("x", C.c_int16), ("y", C.c_int16), ("w", C.c_int16), ("h", C.c_int16), ("mv_cand", (C.c_int16 * 2) * 2), ("start_mv", C.c_int16 * 2), ("has_start", C.c_uint8),
                    ("num_merge", C.c_uint8), ("merge_dir", C.c_uint8 * 5), ("r", C.c_uint8), ("merge_mv", (C.c_int16 * 2) * 5)]

    class MeParams(C.Structure):
        _fields_ = [("lambda_sqrt", C.c_double), ("mv_constraint", C.c_int32), ("sao", C.c_int32), ("deblock", C.c_int32), ("fme_level", C.c_int32)]
    dev.lib.kvz_hip_dev_pu_search.restype = C.c_int
    dev.lib.kvz_hip_dev_pu_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    tex = 128 + 45 * np.sin(xx / 9.0) * np.cos(yy / 13.0) + 30 * np.sin((xx + 2 * yy) / 31.0)
    cur2 = np.clip(tex + rng.normal(0, 2, (h, w)), 0, 255).astype(np.uint8)
    ref2 = np.clip(np.roll(tex, (3, -5), (0, 1)) + rng.normal(0, 2, (h, w)), 0, 255).astype(np.uint8)   # true motion (-5, 3) integer samples
    dc, dr = dev.put(cur2), dev.put(ref2)
    for pw in (8, 16, 32):
        coords = [(x, y) for y in range(0, h - pw + 1, pw) for x in range(0, w - pw + 1, pw)] * (4 if pw == 8 else nrep)
        arr = (MePu * len(coords))()
        for i, (x, y) in enumerate(coords):
            a = arr[i]
            a.x, a.y, a.w, a.h = x, y, pw, pw
            for k in range(2):
                a.mv_cand[k][0], a.mv_cand[k][1] = -20 + int(rng.integers(-6, 7)), 12 + int(rng.integers(-6, 7))
            a.start_mv[0], a.start_mv[1], a.has_start = -16, 8, 1
            a.num_merge = 3
            for k in range(3):
                a.merge_dir[k] = (1, 3, 2)[k]
                a.merge_mv[k][0], a.merge_mv[k][1] = -20 + int(rng.integers(-9, 10)), 12 + int(rng.integers(-9, 10))
        dp, dres = dev.empty(C.sizeof(arr)), dev.empty(len(coords) * 64)
        dev.lib.kvz_hip_dev_upload(dp, C.addressof(arr), C.sizeof(arr))
        for fme_level, label in ((2, "hexbs+hpel"), (0, "hexbs+satd")):
            prm = MeParams(lambda_sqrt=float(np.sqrt(0.57 * 2.0 ** ((25 - 12) / 3.0))), mv_constraint=1, sao=1, deblock=1, fme_level=fme_level)
            ms = time_call(dev, lambda: dev.lib.kvz_hip_dev_pu_search(dc, dr, w, h, dp, len(coords), pw, C.addressof(prm), dres), args.reps, args.warmup)
            resv = dev.get(dres, (len(coords), 16), np.int32)
            found = float(np.mean((resv[:, 0] == -20) & (resv[:, 1] == 12)))
            report(f"pu_search_{pw}x{pw}_{label}", pw, len(coords), ms, pw * pw + (pw + 8) ** 2 + 64,
                   {"path": "one workgroup per PU: rounds of probes in parallel (SAD through L2, clamped), the reference's decisions replayed; fused half-pel pipeline in LDS",
                    "pictures_per_s": round(len(coords) / ((w // pw) * (h // pw)) / (ms * 1e-3), 1), "searches_ending_on_the_true_motion": round(found, 3)})
        dev.free(dp, dres)
    dev.free(dc, dr)

    print(json.dumps({"summary": "bench_kernels", "batch_ctus": args.batch_ctus, "reps": args.reps, "kernels": len(results)}))


if __name__ == "__main__":
    main()
