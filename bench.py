#!/usr/bin/env python3
"""bench.py -- CTUs/s of the batched all-intra hot path on MI355X (BASELINE.json metric, config[1]: 1920x1080 yuv420p
8-bit, --preset ultrafast all-intra).

A "step" is one pass of the hot path (kvz_hip_intra_frames: search + reconstruct every CTU) over one batch of synthetic
1080p frames that is already resident in HBM.  N GPUs = N processes (torch.distributed.run), each with its own batch
(frames are independent pictures with -p 1: weak scaling, no data-path collective).

Prints ONE JSON line on rank 0; see DESIGN.md "Measurement" for every field.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BYTES_PER_CTU = 24576        # 6144 source + 6144 reconstruction + 12288 coefficient bytes per CTU
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
COEFF_WEIGHTS_QP22 = 0x065403F0052C0004  # kvazaar's default fast-coeff-cost weights at QP 22 (fast_coeff_cost.h:48)


def synth_frames(w, h, n, seed):
    """SURVEY.md App. C generator (1080p / 2160p branch), distinct frames"""
    import synth
    return [np.concatenate([p.reshape(-1) for p in planes]) for planes in synth.frames(w, h, n, seed, "large")]


def pmc_traffic(args, launches):
    """HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE + WRITE_SIZE, collected with
    rocprofv3 in separate passes and committed under profiles/): only reported when the committed measurement was taken
    on this exact workload, otherwise null."""
    import glob
    best = None
    if args.tiles:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        w = d.get("workload", {})
        if (w.get("width"), w.get("height"), w.get("frames"), w.get("qp", 22)) == (args.width, args.height, args.frames, args.qp) and \
                (w.get("schedule") == "ticket") == (launches == 1):
            best = d
    return None if best is None else best["bytes_per_launch"]


def pmc_valu_issue(args, launches):
    """share of the SIMDs' VALU issue slots the dominant kernel used (committed SQ counters of this exact workload, else null): the
    bound of this kernel -- it is neither a streaming nor a matrix kernel"""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_sq.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        w = d.get("workload", {})
        if (w.get("width"), w.get("height"), w.get("frames"), w.get("qp", 22)) == (args.width, args.height, args.frames, args.qp) and launches == 1 and not args.tiles:
            best = d
    return None if best is None else {"frac": best["valu_issue_frac"], "valu_insts_per_launch": best["insts_valu"], "salu_insts_per_launch": best["insts_salu"],
                                      "source": "profiles/*pmc_sq.json: rocprofv3 --pmc SQ_INSTS_VALU / SQ_BUSY_CYCLES; a wave64 VALU instruction holds its SIMD for 4 cycles"}


def cpu_baseline(args, frames, model):
    """kvazaar's own AVX2 encoder (oracle/_ref, built from the reference sources) on all host cores over 64 of the benchmark's
    frames (kind "reference"), with the oracle's single-core restatement of exactly this pass nested as "port"; only the port when
    the prebuilt reference encoder is absent."""
    import ctu_common as cc
    import flatapi
    out = {}
    oracle = flatapi.load_oracle()
    n = max(1, min(len(frames), args.cpu_frames))
    t = time.time()
    for f in frames[:n]:
        cc.run_oracle(oracle, model, args.width, args.height, f)
    dt = time.time() - t
    ctus = n * ((args.width + 63) // 64) * ((args.height + 63) // 64)
    out.update(value=ctus / dt, unit="CTUs/s", cores=1, kind="port",
               sample=f"{n} of the benchmark's {args.width}x{args.height} frames through oracle/kvz_oracle_ctu.c (single thread, {dt:.1f} s)")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kvazaar_ref")
    if os.path.exists(ref_bin) and not args.no_ref_encoder:
        import tempfile
        nf = 64  # >= 60 frames for the reference timing; the benchmark's distinct frames, cycled
        with tempfile.NamedTemporaryFile(suffix=".yuv", dir="/tmp") as tmp:
            for i in range(nf):
                tmp.write(frames[i % len(frames)].tobytes())
            tmp.flush()
            threads = os.cpu_count() or 1
            cmd = [ref_bin, "-i", tmp.name, "--input-res", f"{args.width}x{args.height}", "--preset", "ultrafast", "-p", "1",
                   "-q", str(args.qp), "--threads", str(threads), "-o", "/dev/null"]
            times = []
            for _ in range(3):
                t = time.time()
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode == 0:
                    times.append(time.time() - t)
            best = sorted(times)[len(times) // 2] if times else None
            if best:
                # the reference's own CPU path is the headline baseline; the single-core port of exactly this pass rides along
                port = dict(out)
                out = {"value": nf * ((args.width + 63) // 64) * ((args.height + 63) // 64) / best, "unit": "CTUs/s", "cores": threads, "kind": "reference",
                       "sample": f"oracle/_ref/kvazaar_ref (kvazaar's AVX2 strategies, whole encoder incl. CABAC + deblocking) --preset ultrafast -p 1 -q {args.qp} --threads {threads}, "
                                 f"{nf} frames of the benchmark's clip, median of 3, wall incl. file read",
                       "port": port}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=1536, help="frames per GPU and step (the batch resident in HBM)")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames generated on the host; the batch cycles through them")
    ap.add_argument("--qp", type=int, default=22)
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-encoder", action="store_true")
    ap.add_argument("--wpp", action="store_true", help="with --tiles: keep WPP on (kvazaar --tiles CxR --wpp); by default tiles imply --no-wpp as in kvazaar (cfg.c:925-978): "
                                                       "one coder per tile in raster order, i.e. one serial CTU chain per tile")
    ap.add_argument("--no-wpp", action="store_true", help="kvazaar --no-wpp: one serial CTU chain per picture (contexts run from the end of a row into the next)")
    ap.add_argument("--frozen-contexts", action="store_true", help="A/B only: freeze the CABAC contexts at slice start (not kvazaar's behaviour)")
    ap.add_argument("--tiles", default="", help="COLSxROWS: strong-scaling variant (BASELINE config 5): --frames pictures in total, cut into kvazaar's "
                                                "uniform tiles, the tiles dealt to the ranks; every tile is an independent sub-picture (SURVEY.md 8e)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("KVZ_HIP_DEVICE", str(local_rank))

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl")
    else:
        torch.cuda.set_device(local_rank)

    import ctu_common as cc
    import kvazaar_amd
    lib = kvazaar_amd.load_library()  # raises when libkvz_hip.so is missing: no fallback
    model = cc.hip_cost_model(lib, args.qp, cc.coeff_weights(args.qp))
    if args.frozen_contexts:
        model.adaptive = 0
    if args.no_wpp or (args.tiles and not args.wpp):
        model.no_wpp = 1

    from kvazaar_amd import sharding
    batches = []  # (HipBatch, CTUs per picture of that batch, pictures)
    if args.tiles:
        cols, rows = (int(v) for v in args.tiles.lower().split("x"))
        tiles = sharding.tile_grid(args.width, args.height, cols, rows)
        lo, hi = sharding.frames_for_rank(len(tiles), rank, world)
        distinct = synth_frames(args.width, args.height, max(1, min(args.distinct, args.frames)), 1)  # every rank cuts the same clip
        by_geometry = {}
        for t in tiles[lo:hi]:
            by_geometry.setdefault((t[2], t[3]), []).append(t)
        for (tw, th), ts in sorted(by_geometry.items()):
            b = cc.HipBatch(lib, tw, th, args.frames * len(ts))
            subs = [[sharding.crop_tile(f, args.width, args.height, t) for f in distinct] for t in ts]
            for i in range(args.frames):
                for j in range(len(ts)):
                    b.upload(i * len(ts) + j, subs[j][i % len(distinct)])
            batches.append((b, lib.kvz_hip_batch_ctus_per_frame(C.c_void_p(b.handle)), args.frames * len(ts)))
        ctus_per_frame = sum(((t[2] + 63) // 64) * ((t[3] + 63) // 64) for t in tiles)
        job_ctus_per_step = args.frames * ctus_per_frame  # whole job, all ranks
    else:
        distinct = synth_frames(args.width, args.height, max(1, min(args.distinct, args.frames)), 1 + rank)
        batch = cc.HipBatch(lib, args.width, args.height, args.frames)
        for i in range(args.frames):
            batch.upload(i, distinct[i % len(distinct)])
        ctus_per_frame = lib.kvz_hip_batch_ctus_per_frame(C.c_void_p(batch.handle))
        batches.append((batch, ctus_per_frame, args.frames))
        job_ctus_per_step = args.frames * ctus_per_frame * world

    kernel_ms = []
    state = {"launches": 0}

    def step():
        n = 0
        for b, _, _ in batches:  # asynchronous: batches of different geometry overlap on their own streams
            n += lib.kvz_hip_intra_frames(b.handle, C.byref(model))
        for b, _, _ in batches:
            lib.kvz_hip_batch_sync(b.handle)
        state["launches"] = n
        kernel_ms.append(sum(b.kernel_ms() for b, _, _ in batches))

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()

    # EXACTLY `steps` steps between (synchronize + barrier) pairs; MAX over ranks
    dt = sharding.timed_steps(step, args.steps, dist, torch.cuda.synchronize, "cuda")
    launches = state["launches"]

    if rank == 0:
        total_ctus = job_ctus_per_step * args.steps
        value = total_ctus / dt
        # dominant kernel = the CTU kernel: all launches of a step are that kernel; HIP events on the batch's own stream
        k_ms = float(np.mean(kernel_ms))  # rank 0's launches
        per_launch_s = k_ms / 1e3 / launches
        bytes_per_launch = sum(c * n for _, c, n in batches) * BYTES_PER_CTU / launches
        achieved = bytes_per_launch / per_launch_s / 1e9
        result = {
            "metric": "CTUs/s (all-intra ultrafast hot path)", "value": value, "unit": "CTUs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.tiles else "weak", "vs_baseline": None,
            "dtype": "u8/i16 (f64 RD costs)", "data": "synthetic",
            "fps": value / ctus_per_frame,
            "config": {"workload": f"{args.width}x{args.height} yuv420p 8-bit all-intra ultrafast CTU pass (kvz_hip_intra_frames), QP {args.qp}",
                       "frames_per_gpu_per_step": None if args.tiles else args.frames, "frames_per_step": args.frames if args.tiles else args.frames * world,
                       "ctus_per_frame": ctus_per_frame, "distinct_frames": len(distinct),
                       "wpp": not bool(model.no_wpp),
                       "parallelism": (f"--tiles {args.tiles}: tiles sharded over {world} GPU(s), no data-path collective" if args.tiles
                                       else f"frames sharded over {world} GPU(s), no data-path collective")},
            "roofline": {"bound": "hbm", "kernel": "intra_ctu_ticket_kernel" if launches == 1 else "intra_ctu_wave_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args, launches), "valu_issue": pmc_valu_issue(args, launches), "traffic_source": "profiles/*pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, raw counters x 1024) on this workload, else null",
                         "launches_per_step": launches,
                         "avg_launch_us": per_launch_s * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "instruction-issue-bound CTU search (valu_issue.frac of the SIMDs' VALU slots), not a streaming kernel (DESIGN.md 5); per-kernel HBM and MFMA fractions of the streaming primitives: bench_kernels.py"},
        }
        # auxiliary: the per-picture chain an encoder needs from the device before entropy coding -- CTU pass, deblocking, picture
        # hash -- timed the same way on rank 0's batches (not the headline: BASELINE's metric is the CTU pass)
        lib.kvz_hip_batch_deblock.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.kvz_hip_batch_checksums.argtypes = [C.c_void_p, C.c_void_p]
        sums = [np.zeros((n, 3), np.uint32) for _, _, n in batches]
        t0 = time.perf_counter()
        for b, _, _ in batches:
            lib.kvz_hip_intra_frames(b.handle, C.byref(model))
            lib.kvz_hip_batch_deblock(b.handle, args.qp, 0, 0)
        for (b, _, _), o in zip(batches, sums):
            lib.kvz_hip_batch_checksums(b.handle, o.ctypes.data)
        chain_s = time.perf_counter() - t0
        result["chain"] = {"stages": "CTU pass + deblocking + picture-hash checksums (rank 0)", "value": sum(c * n for _, c, n in batches) / chain_s,
                           "unit": "CTUs/s", "ms": chain_s * 1e3}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, distinct, model)
        print(json.dumps(result))
    for b, _, _ in batches:
        b.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
