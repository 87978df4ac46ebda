#!/usr/bin/env python3
"""bench.py -- CTUs/s of the batched all-intra hot path on MI355X (BASELINE.json metric, config[1]: 1920x1080 yuv420p
8-bit, --preset ultrafast all-intra).

A "step" is one pass of the hot path (kvz_hip_intra_frames: search + reconstruct every CTU) over one batch of synthetic
1080p frames that is already resident in HBM.  N GPUs = N processes (torch.distributed.run), each with its own batch
(frames are independent pictures with -p 1: weak scaling, no data-path collective).

After the timed region the batch is CHECKED: frames of the batch that hold the picture of tests/golden/encoder_recon.json
(the reference encoder's own reconstruction of the first frame of the clip) are downloaded and hashed, and the device
checksums of all frames of the batch must agree between copies of the same picture -> "verified" (false => exit code 1).

Prints ONE JSON line on rank 0; see DESIGN.md "Measurement" for every field.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_CTU = 24576        # 6144 source + 6144 reconstruction + 12288 coefficient bytes per CTU
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
GOLDEN = os.path.join(ROOT, "tests", "golden", "encoder_recon.json")


def clip_seed(w, h):
    """the 1080p clip is seed 1, the 2160p clip seed 2 (what the golden encoder digests were taken on)"""
    return 2 if (w, h) == (3840, 2160) else 1


def synth_frames(w, h, n, seed):
    """SURVEY.md App. C generator (1080p / 2160p branch), distinct frames"""
    from kvazaar_amd import synth
    return [np.concatenate([p.reshape(-1) for p in planes]) for planes in synth.frames(w, h, n, seed, "large")]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


# preset -> (coeff_cabac or None = by QP, search_32x32, rdoq, suffix of the golden key, search_nxn)
PRESETS = {"ultrafast": (None, 0, 0, "", 0), "faster": (1, 0, 0, None, 0), "fast": (1, 1, 0, "/fast", 0), "medium-pu13": (1, 1, 1, "/medium-pu13", 0),
           "medium": (1, 1, 1, "/medium", 1)}
# ... and what the reference CLI is given for the CPU baseline of that search (loop filters off where the preset has SAO: the pass timed here is the search)
PRESET_CLI = {"ultrafast": ["--preset", "ultrafast"], "faster": ["--preset", "faster", "--sao", "off"], "fast": ["--preset", "fast", "--sao", "off"],
              "medium-pu13": ["--preset", "medium", "--pu-depth-intra", "1-3", "--sao", "off"], "medium": ["--preset", "medium", "--sao", "off"]}


def golden_digest(w, h, seed, qp, deblock, tiles=None, wpp=False, no_wpp=False, suffix="", picture=0):
    """digest(s) of the reference encoder's reconstruction of frame `picture` of the clip (tests/golden/make_golden.py clip_key), or None.  Picture 0 has
    one-frame fixtures for every configuration; the plain ultrafast pass also has all 8 (1080p) / 4 (4K) pictures of the bench clip (ENCODER_CLIPS_BENCH)."""
    try:
        g = json.load(open(GOLDEN))
    except (OSError, ValueError):
        return None
    stage = 'deblock' if deblock else 'nodeblock'
    if not tiles and not no_wpp and suffix == "":
        for n in (8, 4):
            v = g.get(f"{w}x{h}/n{n}/seed{seed}/large/qp{qp}/{stage}")
            if v and picture < len(v):
                return v[picture]
    if picture != 0:
        return None
    key = (f"{w}x{h}/n1/seed{seed}/large/qp{qp}/{stage}" + ("/nowpp" if no_wpp else "")
           + (f"/tiles{tiles}" + ("-wpp" if wpp else "") if tiles else ""))
    if tiles:
        v = g.get(key + "/per-tile")
        return v[0] if v else None
    if suffix is None:
        return None
    v = g.get(key + suffix)
    return v[0] if v else None


def verify_batches(batches, distinct_n, golden_for):
    """batches: [(HipBatch, slots)] with slots[i] = (distinct picture index, tile index or None) of batch frame i.
    1. every frame's device checksum equals that of the first frame holding the same (picture, tile)  [whole batch]
    2. frames are downloaded and hashed against the reference encoder's reconstruction digest where the fixture has one: three copies of picture 0
       (first / middle / last of the batch) and one copy of every other distinct picture  [golden; golden_for(tile, picture) -> digest or None]"""
    consistent, checked, golden_ok, golden_n, pictures_hashed = True, 0, True, 0, set()
    for b, slots in batches:
        sums = b.checksums()
        first = {}
        for i, s in enumerate(slots):
            if s in first:
                consistent &= bool((sums[i] == sums[first[s]]).all())
            else:
                first[s] = i
            checked += 1
        for tile in sorted({s[1] for s in slots}, key=lambda t: -1 if t is None else t):
            for pic in sorted({s[0] for s in slots}):
                want = golden_for(tile, pic)
                if want is None:
                    continue
                idx = [i for i, s in enumerate(slots) if s == (pic, tile)]
                for i in sorted({idx[0], idx[len(idx) // 2], idx[-1]} if pic == 0 else {idx[0]}):
                    golden_ok &= sha(b.download(i)["rec"]) == want
                    golden_n += 1
                pictures_hashed.add(pic)
    return {"frames_checksummed": checked, "copies_consistent": consistent, "golden_frames_hashed": golden_n, "distinct_pictures_hashed": len(pictures_hashed),
            "golden_ok": (golden_ok if golden_n else None)}


def pmc_file(args, launches, pattern):
    """the committed PMC measurement of this exact workload (profiles/<round>_*<pattern>), newest round first, else None"""
    import glob
    best = None
    if args.tiles:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        w = d.get("workload", {})
        if (w.get("width"), w.get("height"), w.get("frames"), w.get("qp", 22)) == (args.width, args.height, args.frames, args.qp) and \
                (w.get("schedule", "ticket") == "ticket") == (launches == 1):
            best = (path, d)
    return best


def limiter(args, launches):
    """what actually bounds the CTU kernel (it is neither a streaming nor a matrix kernel): the share of the SIMDs' VALU issue
    cycles it uses, from the committed SQ counters of this workload and the MEASURED cycles per wave64 VALU instruction
    (tools/valu_issue_bench.hip -> profiles/*valu_issue.jsonl)"""
    got = pmc_file(args, launches, "*pmc_sq.json")
    if got is None:
        return None
    path, d = got
    out = {"kind": "valu_issue", "source": os.path.relpath(path, ROOT)}
    for k in ("valu_issue_frac", "valu_issue_frac_range", "simd_ns_per_valu_inst", "insts_valu", "insts_salu", "insts_lds", "lane_utilisation", "wave_issue_frac", "wave_wait_frac",
              "wave_issue_stall_frac", "waves_per_simd", "note"):
        if k in d:
            out[k] = d[k]
    return out


INTER_BYTES_PER_CTU = 42824  # the all-intra 24 576 + the reference window of one list, (64 + 2 * 32 + 7)^2 = 18 225 luma samples, + 23 B rounding: "~42.8 KB"


def pmc_leg(leg):
    """The committed counters of an auxiliary leg's kernels (profiles/<round>_pmc_leg_<leg>.json, written by tools/pmc_leg.sh from rocprofv3 --pmc passes of
    `bench.py --only <leg>`; newest round wins): per UNIT of the leg (CTU or picture), so that they scale to the launch measured here."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_pmc_leg_{leg}.json")))
    if not paths:
        return None
    try:
        return paths[-1], json.load(open(paths[-1]))
    except (OSError, ValueError):
        return None


def leg_roofline(leg, kernel, bytes_per_unit, units, kernel_s, unit="CTU"):
    """`roofline` of an auxiliary leg: achieved = algorithmic bytes of the launch / the kernel's time (HIP events on its stream), against the HBM peak; traffic and the SQ
    issue / wait shares from the committed counters of the same kernel (per unit x this launch's units)."""
    achieved = bytes_per_unit * units / kernel_s / 1e9
    out = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "algorithmic_bytes_per_launch": bytes_per_unit * units, f"algorithmic_bytes_per_{unit}": bytes_per_unit, "avg_launch_us": kernel_s * 1e6, "launches_per_step": 1}
    got = pmc_leg(leg)
    if got is not None:
        path, d = got
        pu = d.get("per_unit", {})
        if "FETCH_SIZE" in pu and "WRITE_SIZE" in pu:
            out["traffic"] = (pu["FETCH_SIZE"] + pu["WRITE_SIZE"]) * 1024.0 * units
            out["traffic_over_algorithmic"] = out["traffic"] / out["algorithmic_bytes_per_launch"]
            out["traffic_source"] = os.path.relpath(path, ROOT) + f": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KB, raw) of `bench.py --only {leg}`, per {unit} x this launch's {unit}s"
        if "SQ_WAVE_CYCLES" in pu:
            wc = pu["SQ_WAVE_CYCLES"]
            lim = {"kind": "wave time: issuing / waiting / stalled", "source": os.path.relpath(path, ROOT),
                   "wave_issue_frac": pu.get("SQ_ACTIVE_INST_ANY", 0) / wc, "wave_wait_frac": pu.get("SQ_WAIT_ANY", 0) / wc, "wave_issue_stall_frac": pu.get("SQ_WAIT_INST_ANY", 0) / wc,
                   f"insts_valu_per_{unit}": pu.get("SQ_INSTS_VALU"), f"insts_salu_per_{unit}": pu.get("SQ_INSTS_SALU"), f"insts_lds_per_{unit}": pu.get("SQ_INSTS_LDS"),
                   f"insts_vmem_rd_per_{unit}": pu.get("SQ_INSTS_VMEM_RD"), f"insts_vmem_wr_per_{unit}": pu.get("SQ_INSTS_VMEM_WR"),
                   "note": d.get("note", "shares of SQ_WAVE_CYCLES a resident wavefront spends issuing (SQ_ACTIVE_INST_ANY), parked on s_waitcnt / a barrier (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY)")}
            out["limiter"] = lim
    return out


def cpu_reference(w, h, pictures, cli, n_each, note=""):
    """kvazaar's own encoder (oracle/_ref/kvazaar_ref, AVX2 strategies, whole encoder) on this box's schedulable CPUs for an auxiliary leg, bounded: one single-thread encoder,
    then as many concurrent single-thread encoders as CPUs are granted (`independent`: what the host's cores can do at best, no thread queue or output order between them)."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kvazaar_ref")
    if not os.path.exists(ref_bin):
        return None
    import tempfile
    cpus = host_cpu_facts()["schedulable_cpus"]
    ctus = ((w + 63) // 64) * ((h + 63) // 64)
    with tempfile.NamedTemporaryFile(suffix=".yuv", dir="/tmp") as tmp:
        for i in range(max(n_each, len(pictures))):
            tmp.write(pictures[i % len(pictures)].tobytes())
        tmp.flush()
        base = [ref_bin, "-i", tmp.name, "--input-res", f"{w}x{h}"] + cli + ["-n", str(n_each), "--threads", "0", "--owf", "0", "-o", "/dev/null"]

        def timed(n_proc):
            t0 = time.time()
            ps = [subprocess.Popen(base, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(n_proc)]
            ok = all(p.wait() == 0 for p in ps)
            return time.time() - t0, ok
        s1, ok1 = timed(1)
        sn, okn = timed(cpus)
    return {"unit": "CTUs/s", "kind": "reference", "cpus": cpus, "cli": " ".join(cli),
            "one_thread": {"value": n_each * ctus / s1 if ok1 else None, "sample": f"{n_each} pictures, --threads 0 --owf 0 ({s1:.1f} s)"},
            "independent_one_thread_encoders": {"value": cpus * n_each * ctus / sn if okn else None, "sample": f"{cpus} concurrent encoders x {n_each} pictures, --threads 0 --owf 0 ({sn:.1f} s)"},
            "value": (cpus * n_each * ctus / sn if okn else None), "note": note}


def run_encoder(ref_bin, yuv, w, h, qp, extra, n_frames, preset="ultrafast"):
    """one reference encoder process -> (seconds, ok)"""
    cmd = [ref_bin, "-i", yuv, "--input-res", f"{w}x{h}"] + PRESET_CLI[preset] + ["-p", "1", "-q", str(qp), "-n", str(n_frames), "-o", "/dev/null"] + extra
    t = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True)
    return time.time() - t, r.returncode == 0


def cpu_baseline(args, frames, model):
    """kvazaar's own encoder (oracle/_ref, built from the reference sources; whole encoder incl. CABAC + deblocking) on the GPU box's
    host cores, SURVEY.md 8(d): (a) AVX2, one process with --threads = all host threads; (b) AVX2, saturated: nproc/16 concurrent
    16-thread encoders on the same clip; (c) AVX2, 1 thread; (d) generic (--no-cpuid), 1 thread.  `value` = the best multi-core
    figure.  The oracle's single-core restatement of exactly the CTU pass rides along as "port"."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctu_common as cc   # test infrastructure: the oracle runner, used for this leg only
    import flatapi
    out = {}
    w, h = args.width, args.height
    ctus_pf = ((w + 63) // 64) * ((h + 63) // 64)
    oracle = flatapi.load_oracle()
    n = max(1, min(len(frames), args.cpu_frames))
    t = time.time()
    for f in frames[:n]:
        cc.run_oracle(oracle, model, w, h, f)
    dt = time.time() - t
    port = dict(value=n * ctus_pf / dt, unit="CTUs/s", cores=1, kind="port",
                sample=f"{n} of the benchmark's {w}x{h} frames through oracle/kvz_oracle_ctu.c (single thread, {dt:.1f} s)")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kvazaar_ref")
    if not os.path.exists(ref_bin) or args.no_ref_encoder:
        return port
    import tempfile
    host = host_cpu_facts()
    threads = host["schedulable_cpus"]
    with tempfile.NamedTemporaryFile(suffix=".yuv", dir="/tmp") as tmp:
        nf = 128  # frames in the file: the benchmark's distinct frames, cycled (the long legs re-read it with --loop-input)
        for i in range(nf):
            tmp.write(frames[i % len(frames)].tobytes())
        tmp.flush()

        def median_of(extra, n_frames, reps):
            ts = []
            for _ in range(reps):
                s, ok = run_encoder(ref_bin, tmp.name, w, h, args.qp, extra, n_frames, args.preset)
                if ok:
                    ts.append(s)
            return sorted(ts)[len(ts) // 2] if ts else None

        legs = {}
        n1 = 8 if w * h <= 1920 * 1080 else 2
        s1 = median_of(["--threads", "0", "--owf", "0"], n1, 1)
        if s1:
            legs["avx2_1_thread"] = {"value": n1 * ctus_pf / s1, "cores": 1, "sample": f"{n1} frames, --threads 0 --owf 0 ({s1:.2f} s)"}
        s = median_of(["--no-cpuid", "--threads", "0", "--owf", "0"], n1, 1)
        if s:
            legs["generic_1_thread"] = {"value": n1 * ctus_pf / s, "cores": 1, "sample": f"{n1} frames, --no-cpuid --threads 0 --owf 0 ({s:.2f} s)"}
        # one process on every schedulable CPU: long enough (>= 600 frames at 1080p) that start-up and the first file read are a few per cent
        n_long = max(nf, min(640, int(640 * (1920 * 1080) / (w * h))))
        s = median_of(["--threads", str(threads), "--loop-input"], n_long, 1 if n_long > 256 else 3)
        if s:
            legs["avx2_one_process_all_threads"] = {"value": n_long * ctus_pf / s, "cores": threads,
                                                    "sample": f"{n_long} frames (--loop-input over the {nf}-frame file), --threads {threads}, wall incl. start-up and file read ({s:.2f} s)"}
        # N independent single-thread encoders, N = schedulable CPUs: no thread queue, no output-order coupling -- what the host's cores can do at best
        # (frames per encoder bounded so that the leg takes ~15 s even if the box turns out to schedule far fewer CPUs than it lists)
        fps_known = max([legs[k]["value"] / ctus_pf for k in legs] or [1.0])
        n_ind = max(4, min(32, int(15.0 * 1.5 * fps_known / threads)))
        t = time.time()
        ps = [subprocess.Popen([ref_bin, "-i", tmp.name, "--input-res", f"{w}x{h}"] + PRESET_CLI[args.preset] + ["-p", "1", "-q", str(args.qp), "-n", str(n_ind),
                                "--threads", "0", "--owf", "0", "-o", "/dev/null"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(threads)]
        ok = all(p.wait() == 0 for p in ps)
        s = time.time() - t
        if ok:
            legs["avx2_independent_1thread_xN"] = {"value": threads * n_ind * ctus_pf / s, "cores": threads,
                                                   "sample": f"{threads} concurrent encoders x --threads 0 --owf 0 x {n_ind} frames each, wall of the slowest {s:.2f} s"}
        if threads >= 32:
            procs = threads // 16
            nsat = nf // 2
            t = time.time()
            ps = [subprocess.Popen([ref_bin, "-i", tmp.name, "--input-res", f"{w}x{h}"] + PRESET_CLI[args.preset] + ["-p", "1", "-q", str(args.qp), "-n", str(nsat),
                                    "--threads", "16", "-o", "/dev/null"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(procs)]
            ok = all(p.wait() == 0 for p in ps)
            s = time.time() - t
            if ok:
                legs["avx2_saturated"] = {"value": procs * nsat * ctus_pf / s, "cores": threads,
                                          "sample": f"{procs} concurrent encoders x --threads 16 x {nsat} frames each, wall {s:.2f} s"}
    multi = [k for k in ("avx2_independent_1thread_xN", "avx2_saturated", "avx2_one_process_all_threads") if k in legs]
    if not multi:
        return port
    best = max(multi, key=lambda k: legs[k]["value"])
    eff = None
    if "avx2_1_thread" in legs:
        eff = legs[best]["value"] / legs["avx2_1_thread"]["value"]  # how many single-thread encoders' worth the best multi-core leg reaches
    return {"value": legs[best]["value"], "unit": "CTUs/s", "cores": legs[best]["cores"], "kind": "reference",
            "sample": f"oracle/_ref/kvazaar_ref (kvazaar's AVX2 strategies, whole encoder incl. CABAC + deblocking) {' '.join(PRESET_CLI[args.preset])} -p 1 -q {args.qp}; best of the multi-core legs = "
                      f"{best}: {legs[best]['sample']}", "legs": legs, "port": port, "host": host, "single_thread_equivalents": eff}


def host_cpu_facts():
    """what this process may actually run on: the affinity mask, the cgroup CPU quota, the load when the baseline starts -- `cores` of the
    baseline legs is the schedulable count, not os.cpu_count()"""
    facts = {"os_cpu_count": os.cpu_count(), "sched_getaffinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None}
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
        except OSError:
            continue
        facts["cgroup_" + os.path.basename(path)] = " ".join(txt)
        if path.endswith("cpu.max") and txt and txt[0] != "max":
            quota = float(txt[0]) / float(txt[1])
        elif path.endswith("cfs_quota_us") and txt and int(txt[0]) > 0:
            try:
                quota = int(txt[0]) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            except (OSError, ValueError):
                pass
    facts["cgroup_cpu_quota"] = quota
    try:
        facts["loadavg_1min"] = os.getloadavg()[0]
    except OSError:
        pass
    n = facts["sched_getaffinity"] or facts["os_cpu_count"] or 1
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    facts["schedulable_cpus"] = n
    return facts


def exchange_leg(dist, torch, rank, world, lib, reps=50):
    """The sharded INTER configuration's one collective, measured on its own: per picture every rank contributes the final
    reconstruction of its tiles of a 3840x2160 --tiles 4x2 picture and receives everybody else's (kvazaar_amd/sharding.py ReferenceExchange:
    persistent buffers, all_gather_into_tensor over RCCL, ONE paste launch into the full reference frame).  Tile contents are synthetic
    device buffers written into the send slots (the inter CTU pass that would produce them is not part of this round); checked by an all-reduced byte sum."""
    from kvazaar_amd import sharding
    w, h = 3840, 2160
    plan = sharding.exchange_plan(w, h, 4, 2, world)
    ex = sharding.ReferenceExchange(dist, plan, rank, world, w, h, torch.device("cuda"), lib)
    g = torch.Generator(device="cuda").manual_seed(1000 + rank)
    own_sum = 0
    for k in range(len(ex.mine)):
        slot = ex.send_slot(k)
        slot.copy_(torch.randint(0, 256, (slot.numel(),), dtype=torch.uint8, device="cuda", generator=g))
        own_sum += int(slot.sum(dtype=torch.int64))
    frame = ex.exchange()  # warm-up (RCCL channel setup)
    own = torch.tensor([own_sum], dtype=torch.int64, device="cuda")
    if dist is not None:
        dist.all_reduce(own)
    ok = int(frame.sum(dtype=torch.int64)) == int(own.item())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t = time.perf_counter()
    for _ in range(reps):
        ex.exchange()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    s = (time.perf_counter() - t) / reps
    recv = plan["recv_bytes_per_rank"]
    return {"workload": "3840x2160 --tiles 4x2 reference-frame exchange: all_gather_into_tensor of the ranks' reconstructed tiles + one paste launch (one per inter picture)",
            "ms_per_picture": s * 1e3, "recv_bytes_per_rank": recv, "recv_GBps_per_rank": recv / s / 1e9, "frame_bytes": plan["frame_bytes"], "assembled_ok": ok,
            "xgmi_expected_us": recv / 153e9 * 1e6 if world > 1 else 0.0,
            "note": "expected = received bytes / 153 GB/s (one xGMI link, ring all-gather is per-link bound); measured = the collective + the paste kernel (12.4 MB written "
                    "per picture) + two launches; at world = 1 nothing is received: the figure is the paste alone"}


def build_batches(args, lib, rank, world, width, height, frames, tiles_arg, HipBatch):
    """-> (batches [(HipBatch, slots)], distinct frames, CTUs per (whole) picture, job CTUs per step over all ranks)"""
    from kvazaar_amd import sharding
    seed = clip_seed(width, height)
    batches = []
    if tiles_arg:
        cols, rows = (int(v) for v in tiles_arg.lower().split("x"))
        tiles = sharding.tile_grid(width, height, cols, rows)
        lo, hi = sharding.frames_for_rank(len(tiles), rank, world)
        distinct = synth_frames(width, height, max(1, min(args.distinct, frames)), seed)  # every rank cuts the same clip
        by_geometry = {}
        for ti in range(lo, hi):
            by_geometry.setdefault((tiles[ti][2], tiles[ti][3]), []).append(ti)
        for (tw, th), tis in sorted(by_geometry.items()):
            b = HipBatch(lib, tw, th, frames * len(tis))
            subs = [[sharding.crop_tile(f, width, height, tiles[ti]) for f in distinct] for ti in tis]
            slots = []
            for i in range(frames):
                for j, ti in enumerate(tis):
                    b.upload(i * len(tis) + j, subs[j][i % len(distinct)])
                    slots.append((i % len(distinct), ti))
            batches.append((b, slots))
        # kvazaar's uniform grid gives tiles of up to two heights and two widths: the batches of this rank run side by side, each on its share of the workgroup slots
        # (the persistent pass of the first one launched would otherwise fill the device, and the passes would run one after the other)
        for b, _ in batches:
            b.set_device_share(1, len(batches))
        ctus_per_frame = sum(((t[2] + 63) // 64) * ((t[3] + 63) // 64) for t in tiles)
        return batches, distinct, ctus_per_frame, frames * ctus_per_frame
    distinct = synth_frames(width, height, max(1, min(args.distinct, frames)), seed + rank)
    b = HipBatch(lib, width, height, frames)
    for i in range(frames):
        b.upload(i, distinct[i % len(distinct)])
    batches.append((b, [(i % len(distinct), None) for i in range(frames)]))
    return batches, distinct, b.ctus_per_frame, frames * b.ctus_per_frame * world


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-executes this command line under torch.distributed.run with N ranks on 127.0.0.1 (a free port) and
    returns its exit code.  Rank 0 of the children prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("KVZ_HIP_DEVICE", None)  # every rank takes the device of its LOCAL_RANK
    return subprocess.call(cmd, env=env)


def stub_bench(args, rank, world):
    """The multi-rank skeleton of main() with the device work stubbed out (KVZ_BENCH_STUB=1; gloo): process group, the timed region of sharding.timed_steps
    with its barrier and MAX over ranks, one JSON line from rank 0.  What tests/test_dist_cpu.py runs through `bench.py --gpus 2` on a machine without a GPU."""
    import torch.distributed as dist
    from kvazaar_amd import sharding
    dist.init_process_group(backend="gloo")
    assert dist.get_world_size() == world == args.gpus and dist.get_rank() == rank
    for _ in range(args.warmup):
        time.sleep(0.001)
    dt = sharding.timed_steps(lambda: time.sleep(0.002 * (rank + 1)), args.steps, dist, lambda: None, "cpu")
    units = args.frames * world
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": units * args.steps / dt, "unit": "units/s", "n_gpus": dist.get_world_size(), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "scaling": "weak", "data": "stub"}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU and step (the batch resident in HBM); default 1536, with --tiles: pictures of the whole job -- "
                    "3072 (every GPU of an 8-GPU node then still holds 3072 serial tile chains, what saturates it; 231 GB at one GPU) or 384 with --wpp")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic frames generated on the host; the batch cycles through them")
    ap.add_argument("--qp", type=int, default=22)
    ap.add_argument("--preset", default="ultrafast", choices=sorted(PRESETS) + ["veryfast-inter"], help="veryfast-inter (with --tiles CxR, 3840x2160): BASELINE config 4 sharded by tile "
                    "-- `--preset veryfast --gop lp-g4d3t1`, every rank searches its tiles of --frames independent sequences, the reference frames are exchanged after every picture.  "
                    "Otherwise which of kvazaar's all-intra searches the pass runs (the headline metric is ultrafast): "
                    "faster = CABAC coefficient cost at every QP; fast = + 32x32 CUs searched; medium-pu13 = + RDOQ (`--preset medium --pu-depth-intra 1-3`); "
                    "medium = + 8x8 CUs tried as four 4x4 PUs: BASELINE config 3's preset as it is")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-encoder", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the auxiliary legs (chain, chain_d2h, configs_extra); they only run at --gpus 1")
    ap.add_argument("--only", default="", choices=["", "inter", "medium", "intra4k", "tiles4k", "entropy"], help="developer / profiling: run ONE auxiliary leg at a reduced size and print its "
                    "entry (tools/pmc_leg.sh collects the leg's counters this way); the headline is not measured")
    ap.add_argument("--entropy-pictures", type=int, default=384, help="developer: pictures of `--only entropy`")
    ap.add_argument("--medium-frames", type=int, default=48, help="developer: pictures of `--only medium` (the default run's C3 leg takes 96)")
    ap.add_argument("--inter-sequences", type=int, default=0, help="developer: sequences per launch of the inter leg (default 384; 96 with --only inter)")
    ap.add_argument("--wpp", action="store_true", help="with --tiles: keep WPP on (kvazaar --tiles CxR --wpp); by default tiles imply --no-wpp as in kvazaar (cfg.c:925-978): "
                                                       "one coder per tile in raster order, i.e. one serial CTU chain per tile")
    ap.add_argument("--no-wpp", action="store_true", help="kvazaar --no-wpp: one serial CTU chain per picture (contexts run from the end of a row into the next)")
    ap.add_argument("--frozen-contexts", action="store_true", help="A/B only: freeze the CABAC contexts at slice start (not kvazaar's behaviour)")
    ap.add_argument("--tiles", default="", help="COLSxROWS: strong-scaling variant (BASELINE config 5): --frames pictures in total, cut into kvazaar's "
                                                "uniform tiles, the tiles dealt to the ranks; every tile is an independent sub-picture (SURVEY.md 8e)")
    args = ap.parse_args()
    args.frames_given = args.frames is not None
    if args.frames is None:
        args.frames = 1536 if not args.tiles else (384 if args.wpp else 3072)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.only:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU, the driver's own launch line) and hand over
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench: --gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU (python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}), "
                 "or run `python bench.py --gpus N` without a launcher")
    if world > 1:
        os.environ.setdefault("KVZ_HIP_DEVICE", str(local_rank))
    if world > 1 and os.environ["KVZ_HIP_DEVICE"] != str(local_rank):  # (a single-GPU run picks its device with KVZ_HIP_DEVICE as include/kvz_hip.h says)
        sys.exit(f"bench: rank {rank} is bound to device {os.environ['KVZ_HIP_DEVICE']} but LOCAL_RANK is {local_rank}: one GPU per rank")
    if os.environ.get("KVZ_BENCH_STUB"):  # tests/test_dist_cpu.py: the launch path without a GPU (gloo, a stubbed step)
        stub_bench(args, rank, world)
        return

    import torch
    dist = None
    torch.cuda.set_device(int(os.environ.get("KVZ_HIP_DEVICE", local_rank)) % max(1, torch.cuda.device_count()))  # torch's tensors and the library's kernels on one device
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl")

    import kvazaar_amd
    from kvazaar_amd import sharding
    from kvazaar_amd.batch import HipBatch, PinnedResults, cost_model
    lib = kvazaar_amd.load_library()  # raises when libkvz_hip.so is missing: no fallback

    def model_for(qp, tiles_arg=""):
        m = cost_model(lib, qp)
        if args.frozen_contexts:
            m.adaptive = 0
        if args.no_wpp or (tiles_arg and not args.wpp):
            m.no_wpp = 1
        cab, s32, rdoq, _, nxn = PRESETS[args.preset]
        if cab is not None:
            m.coeff_cabac = cab
        m.search_32x32, m.rdoq, m.search_nxn = s32, rdoq, nxn
        return m

    if args.only:
        only_leg(args, lib, model_for, HipBatch)
        return
    if args.preset == "veryfast-inter":
        tiled_inter_bench(args, lib, dist, torch, rank, world, HipBatch, cost_model)
        if dist is not None:
            dist.destroy_process_group()
        return
    model = model_for(args.qp, args.tiles)
    # HBM the batch needs on this rank: source + reconstruction (1.5 B / pixel each) + two coefficient blocks (3 B / pixel each) + borders and CU maps
    per_picture = args.width * args.height * 9.3
    pictures_here = args.frames if not args.tiles else -(-args.frames // world)
    free_b, _total_b = torch.cuda.mem_get_info()
    frames_asked = args.frames
    if pictures_here * per_picture > 0.9 * free_b:
        fit = int(0.9 * free_b / per_picture)
        args.frames = fit if not args.tiles else fit * world
        if rank == 0:
            print(f"bench: {frames_asked} pictures need {pictures_here * per_picture / 1e9:.0f} GB per GPU, {free_b / 1e9:.0f} GB free: running {args.frames}", file=sys.stderr)
    batches, distinct, ctus_per_frame, job_ctus_per_step = build_batches(args, lib, rank, world, args.width, args.height, args.frames, args.tiles, HipBatch)

    kernel_ms = []
    state = {"launches": 0}

    def step():
        n = 0
        for b, _ in batches:  # asynchronous: batches of different geometry overlap on their own streams
            n += b.launch(model)
        for b, _ in batches:
            b.sync()
        state["launches"] = n
        kernel_ms.append(sum(b.kernel_ms() for b, _ in batches))

    for _ in range(args.warmup):
        step()
    kernel_ms.clear()

    # EXACTLY `steps` steps between (synchronize + barrier) pairs; MAX over ranks
    dt = sharding.timed_steps(step, args.steps, dist, torch.cuda.synchronize, "cuda")
    launches = state["launches"]

    # ---- what was timed is checked (every rank; the verdicts are AND-ed) ----
    seed0 = clip_seed(args.width, args.height)
    golden_applies = rank == 0 or bool(args.tiles)  # frame-sharded ranks > 0 hold other clips (seed + rank): consistency check only
    def golden_for(tile, picture=0):
        if not golden_applies or args.frozen_contexts:
            return None
        if tile is not None:  # per-tile digests of the tiled encode
            if picture != 0:
                return None
            per_tile = golden_digest(args.width, args.height, seed0, args.qp, 0, args.tiles, args.wpp)
            return per_tile[tile] if per_tile else None
        return golden_digest(args.width, args.height, seed0, args.qp, 0, None, False, bool(model.no_wpp), PRESETS[args.preset][3], picture)

    verify = verify_batches(batches, len(distinct), golden_for)
    ok_local = verify["copies_consistent"] and verify["golden_ok"] is not False
    if dist is not None:
        t = torch.tensor([1 if ok_local else 0, verify["golden_frames_hashed"]], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok_all = bool(t[0].item())
    else:
        ok_all = ok_local

    exchange = None
    if not args.no_extra:
        try:
            exchange = exchange_leg(dist, torch, rank, world, lib)
        except Exception as e:  # auxiliary: never take the headline down
            exchange = {"error": repr(e)}

    if rank == 0:
        total_ctus = job_ctus_per_step * args.steps
        value = total_ctus / dt
        # dominant kernel = the CTU kernel: all launches of a step are that kernel; HIP events on the batch's own stream
        k_ms = float(np.mean(kernel_ms))  # rank 0's launches
        per_launch_s = k_ms / 1e3 / launches
        rank_ctus = sum(b.ctus_per_frame * b.n for b, _ in batches)
        bytes_per_launch = rank_ctus * BYTES_PER_CTU / launches
        achieved = bytes_per_launch / per_launch_s / 1e9
        traffic = pmc_file(args, launches, "*pmc_traffic.json")
        result = {
            "metric": "CTUs/s (all-intra ultrafast hot path)", "value": value, "unit": "CTUs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.tiles else "weak", "vs_baseline": None,
            "dtype": "u8/i16 (f64 RD costs)", "data": "synthetic",
            "fps": value / ctus_per_frame,
            # golden_ok None = the fixture has no encoder digest for this workload (only the copy-consistency check ran)
            "verified": bool(ok_all and (verify["golden_ok"] is True or verify["golden_ok"] is None)), "verify": verify,
            "config": {"workload": f"{args.width}x{args.height} yuv420p 8-bit all-intra {args.preset} CTU pass (kvz_hip_intra_frames), QP {args.qp}",
                       "frames_per_gpu_per_step": None if args.tiles else args.frames, "frames_per_step": args.frames if args.tiles else args.frames * world,
                       "ctus_per_frame": ctus_per_frame, "distinct_frames": len(distinct),
                       "wpp": not bool(model.no_wpp),
                       "parallelism": (f"--tiles {args.tiles}: tiles sharded over {world} GPU(s), no data-path collective" if args.tiles
                                       else f"frames sharded over {world} GPU(s), no data-path collective")},
            # "bound" names the roof the fraction is taken against (the task's enum: hbm | mfma); what limits this kernel is "limiter"
            "roofline": {"bound": "hbm", "kernel": "intra_ctu_ticket_kernel" if launches == 1 else "intra_ctu_wave_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None if traffic is None else traffic[1]["bytes_per_launch"],
                         "traffic_source": None if traffic is None else os.path.relpath(traffic[0], ROOT) + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on this workload",
                         "limiter": limiter(args, launches),
                         "launches_per_step": launches,
                         "avg_launch_us": per_launch_s * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "the CTU search is a latency x concurrency machine limited by VALU issue (limiter), not a streaming kernel (DESIGN.md 5); per-kernel HBM and MFMA fractions of the streaming primitives: bench_kernels.py"},
        }
        if exchange is not None:
            result["exchange"] = exchange
        if world == 1 and not args.no_extra and args.preset == "ultrafast":  # the auxiliary legs verify against the ultrafast digests
            extra_legs(args, lib, result, batches, model, model_for, HipBatch, PinnedResults)
        elif not args.no_extra:
            result["legs_skipped"] = ("the auxiliary legs (chain, chain_full, entropy, configs_extra) are single-GPU measurements of the ultrafast preset: run `python bench.py` "
                                      f"(--gpus 1) for them; this run: {world} rank(s), preset {args.preset}")
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args, distinct, model)
            if result.get("chain_full", {}).get("value") and result["cpu_baseline"].get("value"):  # like for like: both sides search, filter and code
                result["chain_full"]["vs_cpu_baseline"] = result["chain_full"]["value"] / result["cpu_baseline"]["value"]
        print(json.dumps(result))
    for b, _ in batches:
        b.close()
    if dist is not None:
        dist.destroy_process_group()
    if not ok_all:
        sys.exit(1)


def tiled_inter_bench(args, lib, dist, torch, rank, world, HipBatch, cost_model):
    """BASELINE config 4 sharded by tile (`--preset veryfast-inter --tiles 4x2 [--gpus N]`): 3840x2160 `--preset veryfast --gop lp-g4d3t1 -q 22 --tiles CxR`, every
    rank holding its tiles of --frames independent sequences.  A step = ONE B picture of every sequence: the inter CTU pass of the rank's tiles (tile geometry: the reference
    is the whole frame), their loop filters, then the exchange -- all-gather of the filtered tiles and of their CU records (RCCL over xGMI), pasted into every rank's
    reference frames.  value = CTUs of B pictures per second, whole job (strong scaling: the tiles are dealt to the ranks).  Verified: the assembled pictures and CU decisions
    of the first three B pictures against the reference encoder's `--tiles` run (tests/golden/inter_tiles.json)."""
    from kvazaar_amd import inter, sharding
    w, h = 3840, 2160
    cols, rows = (int(v) for v in (args.tiles or "4x2").split("x"))
    n = args.frames if args.frames_given else 256  # x 8 tiles = 2 048 chains: without WPP a tile offers one CTU at a time
    tiles = sharding.tile_grid(w, h, cols, rows)
    pictures = synth_frames(w, h, 4, clip_seed(w, h))
    qps = [inter.lowdelay_picture_qp(args.qp, k) for k in range(4)]
    seq = inter.TiledInterSequences(lib, w, h, cols, rows, n, rank, world, dist)
    # the I picture: every tile through the intra pass + its loop filters (tiles: --no-wpp), once -- the sequences are copies of one clip
    rec0 = np.zeros(w * h * 3 // 2, np.uint8)
    ys, cs = w * h, (w // 2) * (h // 2)
    for (x, y, tw, th) in tiles:
        mi = cost_model(lib, qps[0])
        mi.no_wpp = 1
        bi = HipBatch(lib, tw, th, 1)
        bi.upload(0, sharding.crop_tile(pictures[0], w, h, (x, y, tw, th)))
        bi.launch(mi)
        bi.loop_filters(mi, deblock=True, sao=True)
        t = bi.download(0)["rec"]
        bi.close()
        c = (tw // 2) * (th // 2)
        rec0[:ys].reshape(h, w)[y:y + th, x:x + tw] = t[:tw * th].reshape(th, tw)
        rec0[ys:ys + cs].reshape(h // 2, w // 2)[y // 2:(y + th) // 2, x // 2:(x + tw) // 2] = t[tw * th:tw * th + c].reshape(th // 2, tw // 2)
        rec0[ys + cs:].reshape(h // 2, w // 2)[y // 2:(y + th) // 2, x // 2:(x + tw) // 2] = t[tw * th + c:].reshape(th // 2, tw // 2)
    cu0 = inter.intra_picture_cu_info(w, h)
    try:
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "inter_tiles.json"))).get(f"baseline-c4-2160p-tiles{cols}x{rows}") if args.qp == 22 else None
    except (OSError, ValueError):
        gold = None
    i_ok = None if gold is None else hashlib.sha256(rec0.tobytes()).hexdigest()[:24] == gold["rec"][0]

    def start():
        seq.set_reference(np.broadcast_to(rec0, (n, rec0.size)), np.broadcast_to(cu0, (n,) + cu0.shape))

    def picture(k):
        seq.upload_sources(lambda i: pictures[k % len(pictures)])
        seq.run_picture(inter.veryfast_params(qps[min(k, 3)], k, mv_constraint=False))

    # verification chain: B pictures 1..3 from the I picture, sequence 0 and the last one against the reference encoder / each other
    start()
    checks = []
    for k in range(1, 4):
        picture(k)
        f0, c0 = seq.download_reference(0)
        f1, c1 = seq.download_reference(n - 1)
        ok = bool(np.array_equal(f0, f1) and np.array_equal(c0, c1))
        if gold is not None:
            ok = ok and hashlib.sha256(f0.tobytes()).hexdigest()[:24] == gold["rec"][k] and inter.cu_digest(c0) == gold["cu"][k]
        checks.append(ok)
    # timing: every step is the FIRST B picture again, from the I picture's reconstruction -- the reference frames and CU records are put back from a device copy inside
    # the step (a few ms of device-to-device copies, counted), because run_picture leaves the coded picture as the reference and the same source coded against its own
    # reconstruction is a picture of skipped CUs (which the first version of this line timed for all steps but the first)
    start()
    seq.save_reference()
    for _ in range(args.warmup):
        picture(1)
        seq.restore_reference()
    seq.upload_sources(lambda i: pictures[1])
    prm = inter.veryfast_params(qps[1], 1, mv_constraint=False)
    pass_ms = []

    def step():
        seq.restore_reference()
        seq.run_picture(prm)
        pass_ms.append(seq.pass_ms)
    dt = sharding.timed_steps(step, args.steps, dist, torch.cuda.synchronize, "cuda")
    job_ctus = n * sum(((t[2] + 63) // 64) * ((t[3] + 63) // 64) for t in tiles)
    ok_all = all(checks) and i_ok is not False
    verified = ok_all if gold is not None else None  # without the reference encoder's digests (QP != 22, file missing) the checks only say "the sequences agree": not a verdict
    if dist is not None:
        t = torch.tensor([1 if ok_all else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok_all = bool(t[0].item())
    if rank == 0:
        value = job_ctus * args.steps / dt
        k_s = float(np.mean(pass_ms)) / 1e3
        print(json.dumps({
            "metric": "CTUs/s (inter CTU pass + loop filters + reference exchange, BASELINE config 4 sharded by tile)", "value": value, "unit": "CTUs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/i16 (f64 RD costs)", "data": "synthetic", "fps": value / (((w + 63) // 64) * ((h + 63) // 64)),
            "verified": (bool(ok_all) if verified is not None else None),
            "verify": {"i_picture_equals_reference_encoder": i_ok, "b_pictures_equal_reference_encoder_and_between_sequences": checks,
                       "golden": "tests/golden/inter_tiles.json" if gold is not None else None,
                       "reason": None if gold is not None else "no reference-encoder digests for this QP / tile grid: only the agreement between sequences was checked"},
            "config": {"workload": f"{w}x{h} --preset veryfast --gop lp-g4d3t1 -q {args.qp} --tiles {cols}x{rows}: one B picture (QP {qps[1]}) of {n} sequences per step", "sequences": n,
                       "tiles": len(tiles), "tiles_per_rank": seq.per_rank, "parallelism": f"--tiles {cols}x{rows} dealt to {world} GPU(s); one all-gather of pictures + one of CU records per picture"},
            "exchange": {"recv_bytes_per_rank_per_step": (world - 1) * seq.per_rank * n * (seq.slot_px + seq.slot_cu), "frame_bytes": seq.fs, "sequences": n},
            "roofline": leg_roofline("inter", "inter_ctu_ticket_kernel_fast", INTER_BYTES_PER_CTU, n * seq.ctus, k_s),
        }))
    if not ok_all:
        sys.exit(1)


def only_leg(args, lib, model_for, HipBatch):
    """one auxiliary leg at a reduced size (the kernels' counters per CTU / picture do not depend on the batch)"""
    if args.only == "inter":
        out = inter_leg(args, lib, model_for, HipBatch, synth_frames(3840, 2160, 4, clip_seed(3840, 2160)), sequences=args.inter_sequences or 96)
        out.pop("chain", None)
    elif args.only == "medium":
        out = leg_medium(args, lib, model_for, HipBatch, n_med=args.medium_frames)
    elif args.only == "intra4k":
        out = leg_intra4k(args, lib, model_for, HipBatch, n4k=192, steps=1)
    elif args.only == "tiles4k":
        out = leg_tiles4k(args, lib, model_for, HipBatch, n4k=192, steps=1)
    else:
        n = args.entropy_pictures
        frames = synth_frames(args.width, args.height, args.distinct, clip_seed(args.width, args.height))
        b0 = HipBatch(lib, args.width, args.height, n)
        for i in range(n):
            b0.upload(i, frames[i % len(frames)])
        model = model_for(args.qp)
        b0.launch(model)
        b0.sync()
        b0.entropy_code(model)
        t0 = time.perf_counter()
        data, sizes = b0.entropy_code(model)
        ent_s = time.perf_counter() - t0
        out = {"workload": f"entropy coder of {n} {args.width}x{args.height} pictures", "value": n / ent_s, "unit": "pictures/s", "ms": ent_s * 1e3, "units_per_launch": n}
        b0.close()
    print(json.dumps(out))


def inter_leg(args, lib, model_for, HipBatch, pictures, sequences=384):
    """BASELINE config 4 (3840x2160 `--preset veryfast --gop lp-g4d3t1 -q 22`) on the device, as far as the inter CTU pass goes: the I picture through the batched
    intra pass + deblocking + SAO (picture QP 21: intra_qp_offset -1), then the first B picture (picture QP 25: GOP layer 3) of `sequences` independent sequences in
    one launch of kvz_hip_dev_inter_ctu_pass -- every sequence the same clip, so that one result can be checked against the reference encoder's CU decisions
    (tests/golden/inter_recon.json) and the others against it.  value = CTUs of B pictures searched and reconstructed per second (loop filters not included)."""
    from kvazaar_amd import inter
    w, h = 3840, 2160
    if args.qp != 22:
        return {"workload": "3840x2160 --preset veryfast --gop lp-g4d3t1 (BASELINE config 4)", "skipped": "the fixture holds --qp 22"}
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "inter_recon.json")))["baseline-c4-2160p"]
    mi = model_for(args.qp - 1)
    bi = HipBatch(lib, w, h, 1)
    bi.upload(0, pictures[0])
    bi.launch(mi)
    bi.loop_filters(mi, deblock=True, sao=True)
    rec0 = bi.download(0)["rec"]
    bi.close()
    i_ok = hashlib.sha256(np.ascontiguousarray(rec0).tobytes()).hexdigest()[:24] == gold["rec"][0]
    ip = inter.InterPictures(lib, w, h, sequences, with_levels=True)
    cu0 = inter.intra_picture_cu_info(w, h)
    try:
        gold_bits = json.load(open(os.path.join(ROOT, "tests", "golden", "entropy_inter.json"))).get("baseline-c4-2160p")
    except (OSError, ValueError):
        gold_bits = None
    n_b = min(3, len(pictures) - 1)
    for i in range(sequences):
        ip.upload(i, pictures[1], rec0, cu0)
    for k in range(2, n_b + 1):  # the later pictures' sources, resident before the clock starts
        ip.new_source_set()
        ip.use_source_set(k - 1)
        for i in range(sequences):
            ip.upload_source(i, pictures[k])
    ip.use_source_set(0)
    qps = [inter.lowdelay_picture_qp(args.qp, k) for k in range(n_b + 1)]
    prm = inter.veryfast_params(qps[1], 1)
    ip.run(prm)           # warm-up: allocates the work-tree slabs and the loop filters' scratch pictures; the timed run rewrites d_rec
    ip.loop_filters(prm)
    ip.sync()
    t = time.perf_counter()
    ip.run(prm)
    s = time.perf_counter() - t
    lib.kvz_hip_dev_inter_kernel_ms.restype = C.c_float
    kernel_s = float(lib.kvz_hip_dev_inter_kernel_ms()) / 1e3
    _, cu_first = ip.download(0)
    _, cu_last = ip.download(sequences - 1)
    cu_ok = inter.cu_digest(cu_first) == gold["cu"][1]
    # ... and the sequence carried on: loop filters of picture 1, then pictures 2 and 3 from the device's own previous pictures (pass -> CU records for the filters ->
    # deblocking + SAO decision + SAO -> reference of the next picture), everything resident
    pass_s, filt_s, ent_s, chain_ok, rec_ok, bits_ok, slice_bytes = [s], [], [], [bool(cu_ok)], [], [], []
    for k in range(1, n_b + 1):
        prm = inter.veryfast_params(qps[k], k)
        if k > 1:
            ip.advance()
            ip.use_source_set(k - 1)
            t = time.perf_counter()
            ip.run(prm)
            pass_s.append(time.perf_counter() - t)
            chain_ok.append(inter.cu_digest(ip.download(0)[1]) == gold["cu"][k] and np.array_equal(ip.download(0)[1], ip.download(sequences - 1)[1]))
        t = time.perf_counter()
        ip.loop_filters(prm)
        ip.sync()
        filt_s.append(time.perf_counter() - t)
        rec_ok.append(hashlib.sha256(ip.download(0)[0].tobytes()).hexdigest()[:24] == gold["rec"][k] and np.array_equal(ip.download(0)[0], ip.download(sequences - 1)[0]))
        # the pictures' slice data, coded on the device from the CU records, the levels and the SAO decisions just made
        if k == 1:
            ip.entropy_code(prm)  # first use: scratch allocations
        t = time.perf_counter()
        data, sizes = ip.entropy_code(prm)
        ent_s.append(time.perf_counter() - t)
        first = int(sizes[0].sum())
        slice_bytes.append(first)
        bits_ok.append(bool(gold_bits and [int(v) for v in sizes[0]] == gold_bits[k]["sizes"] and hashlib.sha256(bytes(data[:first])).hexdigest()[:24] == gold_bits[k]["sha"]
                            and np.array_equal(sizes[0], sizes[-1]) and bytes(data[:first]) == bytes(data[len(data) - first:])))
    chain_total = sum(pass_s) + sum(filt_s) + sum(ent_s)
    chain = {"stages": f"{n_b} B pictures of {sequences} sequences, each: CTU pass -> CU records for the filters -> deblocking + SAO decision + SAO -> slice data (entropy coder, "
                       "downloaded) -> the next picture's reference; pictures, CU records and levels resident",
             "value": n_b * sequences * ip.ctus / chain_total, "unit": "CTUs/s", "fps": n_b * sequences / chain_total, "picture_qps": qps,
             "pass_ms": [round(x * 1e3, 1) for x in pass_s], "loop_filters_ms": [round(x * 1e3, 1) for x in filt_s], "entropy_ms": [round(x * 1e3, 1) for x in ent_s],
             "slice_data_bytes_per_picture": slice_bytes,
             "verified": bool(all(chain_ok) and all(rec_ok) and all(bits_ok)),
             "verify": {"cu_decisions_equal_reference_encoder_per_picture": [bool(v) for v in chain_ok], "final_pictures_equal_reference_encoder": [bool(v) for v in rec_ok],
                        "slice_data_equals_reference_encoder_bitstream": [bool(v) for v in bits_ok]}}
    ip.close()
    # the reference encoder on the same clip and settings, on this box's host cores: (a) one thread, (b) its default threading, (c) as many independent one-thread encoders
    # as the box grants CPUs.  Whole encoder (I picture, entropy coding, loop filters included): a reported baseline, bounded to a few seconds each
    cpu = None
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kvazaar_ref")
    if os.path.exists(ref_bin) and not args.no_ref_encoder and not args.no_cpu_baseline and not getattr(args, "only", ""):
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".yuv") as tmp:
            nfr = 8
            for i in range(nfr):
                tmp.write(pictures[i % len(pictures)].tobytes())
            tmp.flush()
            base = [ref_bin, "-i", tmp.name, "--input-res", f"{w}x{h}", "--preset", "veryfast", "--gop", "lp-g4d3t1", "-q", str(args.qp), "-o", "/dev/null"]
            def timed(cmds):
                t0 = time.time()
                ps = [subprocess.Popen(c, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for c in cmds]
                ok = all(p.wait() == 0 for p in ps)
                return time.time() - t0, ok
            cpus = len(os.sched_getaffinity(0))
            try:
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
                if quota != "max":
                    cpus = max(1, min(cpus, int(int(quota) / int(period))))
            except (OSError, ValueError):
                pass
            s1, ok1 = timed([base + ["-n", "4", "--threads", "0", "--owf", "0"]])
            sd, okd = timed([base + ["-n", str(nfr)]])
            sn, okn = timed([base + ["-n", "4", "--threads", "0", "--owf", "0"] for _ in range(cpus)])
            cpu = {"unit": "CTUs/s", "kind": "reference", "cpus": cpus,
                   "one_thread": {"value": 4 * ip.ctus / s1 if ok1 else None, "sample": f"4 pictures, --threads 0 --owf 0 ({s1:.1f} s)"},
                   "default_threading": {"value": nfr * ip.ctus / sd if okd else None, "sample": f"{nfr} pictures, threads / owf auto ({sd:.1f} s)"},
                   "independent_one_thread_encoders": {"value": cpus * 4 * ip.ctus / sn if okn else None, "sample": f"{cpus} concurrent encoders x 4 pictures, --threads 0 ({sn:.1f} s)"}}
    return {"cpu_reference": cpu, "workload": f"{w}x{h} --preset veryfast --gop lp-g4d3t1 -q {args.qp} (BASELINE config 4): CTU pass of the first B picture (QP {args.qp + 3}; merge / AMVP / temporal "
                        f"candidates, hexbs + half-pel search, early skip, uni- and bi-prediction, the intra alternative, zero-coefficient RDO, CABAC-context life cycle) of "
                        f"{sequences} independent sequences in one launch, from the I picture's reconstruction (intra pass + deblocking + SAO at QP {args.qp - 1}, on the device)",
            "value": sequences * ip.ctus / s, "unit": "CTUs/s", "fps": sequences / s, "ms": s * 1e3, "kernel_ms": kernel_s * 1e3, "units_per_launch": sequences * ip.ctus,
            "roofline": leg_roofline("inter", "inter_ctu_ticket_kernel_fast", INTER_BYTES_PER_CTU, sequences * ip.ctus, kernel_s),
            "verified": bool(i_ok and cu_ok and np.array_equal(cu_first, cu_last)),
            "verify": {"i_picture_reconstruction_equals_reference_encoder": bool(i_ok), "b_picture_cu_decisions_equal_reference_encoder": bool(cu_ok),
                       "copies_consistent": bool(np.array_equal(cu_first, cu_last))},
            "chain": chain,
            "note": f"one wavefront per CTU, candidates / program state / tables in LDS, twelve CTUs per CU (DESIGN.md 3.8); {sequences} sequences per launch: a 4K picture alone is a WPP chain of 127 CTU steps (~0.94 s); `chain` carries the sequence on through the loop filters and the next pictures; the traffic above the algorithmic bytes is scratch: the 85 recursion calls of a CTU each save the callee-saved registers their body clobbers (~1.1 MB written, ~0.5 MB read per CTU; profiles/experiments/README.md r04_v for the forms without calls, all slower)"}


def extra_legs(args, lib, result, batches, model, model_for, HipBatch, PinnedResults):
    """auxiliary measurements at --gpus 1 (not the headline):
    chain      CTU pass + deblocking + picture-hash checksums of the whole batch
    chain_d2h  the same + download of everything the host entropy coder needs (coefficients, CU depth / mode, reconstruction of every
               frame) into pinned host memory, two half-size batches alternating on their own streams so that one batch's D2H
               overlaps the other's kernel: "frames an encoder can actually consume per second"
    configs_extra  a short 3840x2160 run (the north-star size), verified against the reference encoder's 4K digest"""
    rank_ctus = sum(b.ctus_per_frame * b.n for b, _ in batches)
    t0 = time.perf_counter()
    for b, _ in batches:
        b.launch(model)
        b.deblock(args.qp, wait=False)
    for b, _ in batches:
        b.checksums()
    chain_s = time.perf_counter() - t0
    result["chain"] = {"stages": "CTU pass + deblocking + picture-hash checksums", "value": rank_ctus / chain_s, "unit": "CTUs/s", "ms": chain_s * 1e3}
    if args.tiles:
        return
    # ... and with `--sao full` on top (deblocking, SAO parameter decision per LCU, SAO reconstruction): what presets from veryfast up run
    b0 = batches[0][0]
    b0.launch(model)
    b0.loop_filters(model, deblock=True, sao=True)  # first use allocates the SAO buffers
    t0 = time.perf_counter()
    b0.launch(model)
    b0.loop_filters(model, deblock=True, sao=True, wait=False)
    b0.checksums()
    sao_s = time.perf_counter() - t0
    try:
        want = json.load(open(GOLDEN)).get(f"{args.width}x{args.height}/n1/seed{clip_seed(args.width, args.height)}/large/qp{args.qp}/deblock/sao")
    except (OSError, ValueError):
        want = None
    result["chain_sao"] = {"stages": "CTU pass + deblocking + SAO decision (kvz_sao_search_lcu of every LCU) + SAO reconstruction + picture-hash checksums",
                           "value": b0.ctus_per_frame * b0.n / sao_s, "unit": "CTUs/s", "ms": sao_s * 1e3,
                           "verified": (sha(b0.download(0)["rec"]) == want[0]) if want else None}
    # ---- the entropy coder in its real mode on the device (kvz_hip_batch_entropy_code): slice data instead of levels over PCIe ----
    try:
        b0.launch(model)
        b0.sync()
        b0.entropy_code(model)  # first use: scratch allocations
        t0 = time.perf_counter()
        data, sizes = b0.entropy_code(model)
        ent_s = time.perf_counter() - t0
        ent_ok = None
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "entropy.json"))).get("bench-1080p")
        except (OSError, ValueError):
            gold = None
        if gold and (args.width, args.height, args.qp, args.preset) == (1920, 1080, 22, "ultrafast") and not args.no_wpp and not args.frozen_contexts:
            ent_ok, at = True, 0
            for i in range(min(b0.n, len(gold), args.distinct)):  # picture i of the batch is frame i of the clip
                total = int(sizes[i].sum())
                ent_ok = ent_ok and [int(v) for v in sizes[i]] == gold[i]["sizes"] and hashlib.sha256(bytes(data[at:at + total])).hexdigest()[:24] == gold[i]["sha"]
                at += total
            ent_ok = bool(ent_ok and all(np.array_equal(sizes[i], sizes[i % args.distinct]) for i in range(b0.n)))
        result["entropy"] = {"stages": "kvz_encode_coding_tree + kvz_encode_coeff_nxn + CABAC of every picture's slice data on the device from the resident results of the CTU pass "
                                       "(bins per CTU, row-start contexts, one arithmetic coder per WPP substream moving its code value out 32 bits at a time, emulation prevention by position), substreams and entry points downloaded",
                             "value": b0.n / ent_s, "unit": "pictures/s", "ctus_per_s": b0.ctus_per_frame * b0.n / ent_s, "ms": ent_s * 1e3,
                             "slice_data_bytes_per_picture": len(data) / b0.n, "levels_bytes_per_picture": b0.ctus_per_frame * 12288, "units_per_launch": b0.n,
                             "roofline": dict(leg_roofline("entropy", "dev_entropy_bins_phased_kernel + dev_entropy_row_ctx_kernel + dev_entropy_code_wide_kernel + dev_entropy_escape_count_kernel + dev_entropy_compact_kernel",
                                                           b0.ctus_per_frame * 12288 + 2 * (args.width * args.height // 64) + len(data) / b0.n, b0.n, ent_s, unit="picture"),
                                              note="achieved = (levels + CU depth / mode maps read, slice data written) per picture / the call's wall time, which includes the download of the slice data"),
                             "verified": ent_ok, "verify": "slice data and entry points of the clip's pictures equal the reference encoder's bitstream (tests/golden/entropy.json)"}
    except Exception as e:  # auxiliary: never take the headline down
        result["entropy"] = {"error": repr(e)}
    # ---- chain + D2H, double-buffered ----
    main_batch = batches[0][0]
    half = max(1, min(args.frames // 2, 384))
    distinct = synth_frames(args.width, args.height, max(1, min(args.distinct, half)), clip_seed(args.width, args.height))
    pair = []
    for _ in range(2):
        b = HipBatch(lib, args.width, args.height, half)
        for i in range(half):
            b.upload(i, distinct[i % len(distinct)])
        pair.append((b, PinnedResults(b)))
    reps = 3

    def run(with_d2h):
        for b, _ in pair:
            b.sync()
        t = time.perf_counter()
        for _ in range(reps):
            for k, (b, pinned) in enumerate(pair):
                b.sync()          # the host is done with this batch's previous results
                b.launch(model)
                b.deblock(args.qp, wait=False)
                pair[1 - k][0].order_after(b)  # the other batch's next pass starts behind this batch's deblocking, not inside it ...
                if with_d2h:
                    pinned.download_async()    # ... so that this copy runs on the copy engines during that pass
        for b, _ in pair:
            b.sync()
        return time.perf_counter() - t

    run(True)  # warm-up (first touch of the pinned pages)
    s_plain, s_d2h = run(False), run(True)
    ctus = 2 * reps * half * main_batch.ctus_per_frame
    payload = pair[0][1].bytes / half
    # the downloaded bytes are the device's: frame 0 of the pinned buffer == the golden picture (deblocked) where the fixture has it
    want = golden_digest(args.width, args.height, clip_seed(args.width, args.height), args.qp, 1)
    got = sha(pair[0][1].array("rec")[:args.width * args.height * 3 // 2])
    result["chain_d2h"] = {"stages": "CTU pass + deblocking + D2H of coefficients, CU depth/mode and reconstruction of every frame into pinned host memory; two batches of "
                                     f"{half} frames alternating on two streams", "value": ctus / s_d2h, "unit": "CTUs/s", "fps": ctus / s_d2h / main_batch.ctus_per_frame,
                           "without_d2h": ctus / s_plain, "payload_bytes_per_frame": payload, "pcie_GBps": 2 * reps * half * payload / s_d2h / 1e9,
                           "host_copy_verified": (got == want) if want else None}
    # ---- the whole all-intra chain: pass -> deblocking -> entropy coder -> slice data on the host (what the CPU baseline's encoder does) ----
    try:
        try:
            gold = json.load(open(os.path.join(ROOT, "tests", "golden", "entropy.json"))).get("bench-1080p")
        except (OSError, ValueError):
            gold = None
        applies = (args.width, args.height, args.qp, args.preset) == (1920, 1080, 22, "ultrafast") and not args.no_wpp and not args.frozen_contexts
        for b, pinned in pair:  # the D2H leg's batches make room
            b.close()
            pinned.close()
        pair = []
        half_full = args.frames  # batches of the headline's own size: the coder's third stage is one lane per WPP substream, bound by the chain (768 pictures: 37 ms, 1 536: 40 ms)
        full = []
        for _ in range(2):
            b = HipBatch(lib, args.width, args.height, half_full)
            for i in range(half_full):
                b.upload(i, distinct[i % len(distinct)])
            full.append(b)
        reps_full = 3  # six turns: the pipeline's first pass and last coder have nothing beside them
        (s_res, pictures_res, _, ok_res), tries_res = chain_full_best(2, full, model, args.qp, reps_full, gold if applies else None, len(distinct))
        # ... and with the source pictures arriving over PCIe inside the timed region, as the reference encoder reads its input: every batch but a batch's first gets its
        # pictures from a pinned host buffer on the batch's upload queue (kvz_hip_batch_upload_all_async), beside the other batch's pass
        from kvazaar_amd.batch import pinned_bytes, pinned_free
        frame_bytes = args.width * args.height * 3 // 2
        src_ptr, src_view = pinned_bytes(lib, half_full * frame_bytes)
        for i in range(half_full):
            src_view[i * frame_bytes:(i + 1) * frame_bytes] = distinct[i % len(distinct)]
        (s_full, pictures, per_pic, ok, n_up), tries_full = chain_full_best(2, full, model, args.qp, reps_full, gold if applies else None, len(distinct), src_ptr=src_ptr)
        for b in full:
            b.close()
        pinned_free(lib, src_ptr)
        result["chain_full"] = {"stages": "source pictures uploaded from pinned host memory -> CTU pass -> deblocking -> entropy coder on the device -> slice data and entry points downloaded into "
                                          f"pinned host memory; two batches of {half_full} pictures in turn, a batch's pass started when the other batch's coder has queued its chain-bound third "
                                          "stage (kvz_hip_batch_entropy_code_then), its next pictures uploaded beside the other batch's pass (kvz_hip_batch_upload_all_async)",
                                "value": pictures * main_batch.ctus_per_frame / s_full, "unit": "CTUs/s", "fps": pictures / s_full, "ms_per_batch": s_full / (2 * reps_full) * 1e3, "batches_timed": 2 * reps_full,
                                "attempts_ms_per_batch": [t and t / (2 * reps_full) * 1e3 for t in tries_full],  # the faster of two timed regions is reported (chain_full_best says why)
                                "h2d": {"batches_uploaded_in_timed_region": n_up, "bytes_per_batch": half_full * frame_bytes, "h2d_GBps": n_up * half_full * frame_bytes / s_full / 1e9,
                                        "note": "averaged over the timed region; the first pass of each of the two batches runs on pictures already resident"},
                                "source_resident": {"value": pictures_res * main_batch.ctus_per_frame / s_res, "fps": pictures_res / s_res, "ms_per_batch": s_res / (2 * reps_full) * 1e3, "attempts_ms_per_batch": [t and t / (2 * reps_full) * 1e3 for t in tries_res], "verified": ok_res,
                                                    "note": "the same chain with the pictures resident in HBM across all batches (what round 5 reported)"},
                                "slice_data_bytes_per_picture": per_pic, "verified": ok,
                                "verify": "slice data and entry points of the clip's pictures equal the reference encoder's bitstream (tests/golden/entropy.json bench-1080p)",
                                "note": "the like-for-like line against cpu_baseline (kvazaar's whole encoder: search, deblocking, CABAC, bitstream); parameter sets, slice "
                                        "headers and NAL framing (a few dozen bytes per picture) stay on the host"}
    except Exception as e:  # auxiliary: never take the headline down
        result["chain_full"] = {"error": repr(e)}
    for b, pinned in pair:
        b.close()
        pinned.close()
    # ---- the headline on 64 distinct pictures, checked against the reference encoder run here ----
    if args.preset == "ultrafast" and not args.no_wpp and not args.frozen_contexts and not args.tiles:
        try:
            result["distinct64"] = leg_distinct(args, lib, model, HipBatch)
            if result["distinct64"].get("value"):
                result["distinct64"]["vs_headline"] = result["distinct64"]["value"] / result["value"]
        except Exception as e:  # auxiliary: never take the headline down
            result["distinct64"] = {"error": repr(e)}
    # ---- the north-star size ----
    if (args.width, args.height) != (3840, 2160):
        result["configs_extra"] = [leg_intra4k(args, lib, model_for, HipBatch), leg_tiles4k(args, lib, model_for, HipBatch)]
        # ---- BASELINE config 4: `--preset veryfast --gop lp-g4d3t1` at 3840x2160: the first B picture of many independent sequences ----
        try:
            result["configs_extra"].append(inter_leg(args, lib, model_for, HipBatch, synth_frames(3840, 2160, 4, clip_seed(3840, 2160)), sequences=args.inter_sequences or 384))
        except Exception as e:  # auxiliary: never take the headline down
            result["configs_extra"].append({"workload": "3840x2160 --preset veryfast --gop lp-g4d3t1 (BASELINE config 4)", "error": repr(e)})
        result["configs_extra"].append(leg_medium(args, lib, model_for, HipBatch))


def chain_full_best(attempts, pair, *a, **kw):
    """chain_full's timed region `attempts` times: the fastest attempt's result + every attempt's seconds (None: the attempt failed).  The region moves gigabytes over
    PCIe in both directions and its rate follows the load of the box's host side (a shared node: the same code measured 1.5 and 7.9 GB/s of upload minutes apart); and
    about one attempt in seventy ends in a CTU hand-off that never arrives (the pass's bounded wait reports it, the batch's results are void: DESIGN.md section 8,
    tools/chain_stress.py) -- such an attempt is reset and does not count.  The kernel-only legs, the headline included, are single measurements."""
    runs, secs, failed = [], [], []
    for _ in range(attempts):
        try:
            r = chain_full(pair, *a, **kw)
            runs.append(r)
            secs.append(r[0])
        except Exception as e:
            failed.append(repr(e))
            secs.append(None)
            for b in pair:
                b.reset()
    if not runs:
        raise RuntimeError("every attempt failed: " + "; ".join(failed))
    return min(runs, key=lambda r: r[0]), secs


def chain_full(pair, model, qp, reps, gold, n_distinct, src_ptr=None):
    """CTU pass -> deblocking -> entropy coder on the device -> slice data + entry points downloaded, for two resident batches in turn.  The coder's first stage wants
    the whole device and so does the pass (one persistent launch that takes every workgroup slot it finds); the coder's third stage is a few hundred wavefronts that
    each follow one substream's chain, and then there is the download.  So the other batch's pass is started when this batch's coder has queued its third stage
    (kvz_hip_batch_entropy_code_then) and runs beside the rest of it (tools/chain_probe.py: 1 536-picture batches, one after the other 427 ms per batch, this way 402 ms;
    a pass started any earlier keeps the coder's first stage waiting: no gain).  What the host gets per picture is what kvazaar's
    encoder_state_worker_encode_lcu_bitstream wrote (encoderstate.c:636-745) -- the like-for-like counterpart of the reference encoder timed on the CPU, which also
    searches, filters and codes.  Returns (seconds, pictures, slice-data bytes per picture, verified)."""
    for b in pair:
        b.sync()
    for b in pair:                 # first use: the coder's scratch allocations
        b.launch(model)
        b.deblock(qp, wait=False)
        b.entropy_code(model)
    for b in pair:  # the slice data of a batch comes down beside the next batch's deblocking and coder (13 ms per 1 536 1080p pictures that sat between two passes)
        b.entropy_defer_download(True)
    last = {}
    uploads = [0]
    t = time.perf_counter()
    turns = reps * len(pair)
    cur = pair[0]
    cur.launch(model)
    cur.deblock(qp, wait=False)
    for i in range(turns):
        nxt = pair[(i + 1) % len(pair)]
        more = i + 1 < turns
        if src_ptr is not None and i + len(pair) < turns:
            # the pictures of cur's NEXT pass, which is due one turn from now, 65 ms after nxt's pass has ended: the copy (84 ms on the copy engine) waits on the device
            # for the end of cur's pass in flight -- the only reader of the source pictures -- and runs beside cur's coder and the start of nxt's pass.  Nothing on the
            # chain may queue a host-to-device copy of its own behind it: the coder computes its offset lists on the device, the pass keeps the model's tables, the
            # hand-off check reads a pinned word (tools/chain_h2d_trace.sh shows what each of those cost)
            cur.upload_all_async(src_ptr)
            uploads[0] += 1
        last[id(cur)] = cur.entropy_code(model, then=(nxt, model) if more else None)  # waits for cur's pass + deblocking, codes, starts nxt's pass, downloads
        if more:
            nxt.deblock(qp, wait=False)  # queued behind the pass the call above started
        cur = nxt
    for b in pair:  # the last downloads
        b.sync()
    s_full = time.perf_counter() - t
    for b in pair:
        b.entropy_defer_download(False)
    pictures = reps * sum(b.n for b in pair)
    ok, nbytes = None, 0
    for b in pair:
        data, sizes = last[id(b)]
        nbytes += len(data)
        if gold:
            good, at = True, 0
            for i in range(min(b.n, len(gold), n_distinct)):  # picture i of a batch is frame i of the clip
                total = int(sizes[i].sum())
                good = good and [int(v) for v in sizes[i]] == gold[i]["sizes"] and hashlib.sha256(bytes(data[at:at + total])).hexdigest()[:24] == gold[i]["sha"]
                at += total
            good = bool(good and all(np.array_equal(sizes[i], sizes[i % n_distinct]) for i in range(b.n)))
            ok = good if ok is None else (ok and good)
    if src_ptr is not None:
        return s_full, pictures, nbytes / sum(b.n for b in pair), ok, uploads[0]
    return s_full, pictures, nbytes / sum(b.n for b in pair), ok


def nal_payloads(stream):
    """the payloads of the VCL NAL units (types 0..21) of an Annex B byte stream as they stand in it (emulation prevention included), in order"""
    starts, i, n = [], 0, len(stream)
    while True:
        i = stream.find(b"\x00\x00\x01", i)
        if i < 0:
            break
        starts.append(i + 3)
        i += 3
    out = []
    for k, st in enumerate(starts):
        e = starts[k + 1] - 3 if k + 1 < len(starts) else n
        while e > st and stream[e - 1] == 0:
            e -= 1
        if ((stream[st] >> 1) & 0x3F) <= 21:
            out.append(stream[st + 2:e])
    return out


def leg_distinct(args, lib, model, HipBatch, n_distinct=64, steps=3):
    """The headline pass on a batch that cycles through 64 DISTINCT pictures (the default batch repeats 8), every one of them checked against the reference encoder run
    inside this bench on the same pictures -- not against fixtures: kvazaar_ref (oracle/_ref, AVX2, all granted CPUs) codes the 64-picture clip once with --debug;
    the device's deblocked reconstruction of each distinct picture must hash like the encoder's, and the device coder's slice data must be the tail of the encoder's slice
    NAL unit for that picture."""
    import subprocess
    import tempfile
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kvazaar_ref")
    w, h, n = args.width, args.height, args.frames
    if not os.path.exists(ref_bin):
        return {"skipped": "oracle/_ref/kvazaar_ref not built"}
    frames = synth_frames(w, h, n_distinct, clip_seed(w, h))
    fb = w * h * 3 // 2
    cpus = host_cpu_facts()["schedulable_cpus"]
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        yuv, rec, out = os.path.join(d, "in.yuv"), os.path.join(d, "rec.yuv"), os.path.join(d, "out.hevc")
        with open(yuv, "wb") as f:
            for fr in frames:
                f.write(fr.tobytes())
        t0 = time.perf_counter()
        r = subprocess.run([ref_bin, "-i", yuv, "--input-res", f"{w}x{h}", "--preset", "ultrafast", "-p", "1", "-q", str(args.qp), "--threads", str(cpus), "--debug", rec, "-o", out],
                           capture_output=True, text=True)
        ref_s = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": "kvazaar_ref: " + r.stderr[-300:]}
        recon = np.fromfile(rec, np.uint8)
        want_rec = [sha(recon[i * fb:(i + 1) * fb]) for i in range(n_distinct)]
        payloads = nal_payloads(open(out, "rb").read())
    if len(payloads) != n_distinct:
        return {"error": f"{len(payloads)} slice NAL units for {n_distinct} pictures"}
    b = HipBatch(lib, w, h, n)
    for i in range(n):
        b.upload(i, frames[i % n_distinct])
    b.run(model)
    t0 = time.perf_counter()
    kms = []
    for _ in range(steps):
        b.run(model)
        kms.append(b.kernel_ms())
    s = time.perf_counter() - t0
    b.deblock(args.qp, wait=False)
    data, sizes = b.entropy_code(model)
    sums = b.checksums()
    consistent = all(bool((sums[i] == sums[i % n_distinct]).all()) for i in range(n))
    rec_ok = all(sha(b.download(i)["rec"]) == want_rec[i] for i in range(n_distinct))
    ent_ok, at = True, 0
    for i in range(n_distinct):
        total = int(sizes[i].sum())
        ent_ok = ent_ok and payloads[i].endswith(bytes(data[at:at + total])) and total > 0
        at += total
    ent_ok = bool(ent_ok and all(np.array_equal(sizes[i], sizes[i % n_distinct]) for i in range(n)))
    units = n * b.ctus_per_frame
    res = {"workload": f"{w}x{h} all-intra ultrafast CTU pass, QP {args.qp}, {n} frames resident cycling through {n_distinct} distinct pictures, {steps} steps",
           "value": steps * units / s, "unit": "CTUs/s", "fps": steps * n / s, "ms_per_step": s / steps * 1e3, "kernel_ms": float(np.mean(kms)),
           "verified": bool(consistent and rec_ok and ent_ok),
           "verify": {"reference": f"oracle/_ref/kvazaar_ref --preset ultrafast -q {args.qp} --threads {cpus} --debug on the same {n_distinct} pictures, run inside this bench ({ref_s:.1f} s)",
                      "distinct_pictures_reconstruction_hashed": n_distinct, "reconstruction_ok": bool(rec_ok), "distinct_pictures_slice_data_compared": n_distinct, "slice_data_ok": bool(ent_ok),
                      "copies_consistent": bool(consistent), "frames_checksummed": n}}
    b.close()
    return res


def leg_intra4k(args, lib, model_for, HipBatch, n4k=384, steps=3):
    """the headline pass at the north star's size, 3840x2160 all-intra ultrafast, verified against the reference encoder's 4K digests; with kvazaar's own encoder on the
    host's CPUs beside it (the north star's >= 10x is quoted on this workload)"""
    w, h = 3840, 2160
    m4 = model_for(args.qp)
    d4 = synth_frames(w, h, 4, clip_seed(w, h))
    b4 = HipBatch(lib, w, h, n4k)
    for i in range(n4k):
        b4.upload(i, d4[i % len(d4)])
    b4.run(m4)
    t = time.perf_counter()
    kms = []
    for _ in range(steps):
        b4.run(m4)
        kms.append(b4.kernel_ms())
    s = time.perf_counter() - t
    v = verify_batches([(b4, [(i % len(d4), None) for i in range(n4k)])], len(d4), lambda tile, picture=0: golden_digest(w, h, clip_seed(w, h), args.qp, 0, picture=picture))
    units = n4k * b4.ctus_per_frame
    value = steps * units / s
    out = {"workload": f"{w}x{h} yuv420p 8-bit all-intra ultrafast CTU pass, QP {args.qp}, {n4k} frames resident, {steps} steps",
           "value": value, "unit": "CTUs/s", "fps": steps * n4k / s, "kernel_ms": float(np.mean(kms)), "units_per_launch": units,
           "roofline": leg_roofline("intra4k", "intra_ctu_ticket_kernel", BYTES_PER_CTU, units, float(np.mean(kms)) / 1e3),
           "verified": bool(v["copies_consistent"] and v["golden_ok"] is not False), "verify": v}
    b4.close()
    # the whole chain at this size (chain_full): two batches of n4k / 2 pictures
    if not getattr(args, "only", ""):
        try:
            try:
                gold = json.load(open(os.path.join(ROOT, "tests", "golden", "entropy.json"))).get("bench-2160p")
            except (OSError, ValueError):
                gold = None
            half = n4k  # 2 x 384 pictures of 3840x2160: the coder's third stage wants big batches (one lane per WPP substream)
            pair = []
            for _ in range(2):
                b = HipBatch(lib, w, h, half)
                for i in range(half):
                    b.upload(i, d4[i % len(d4)])
                pair.append(b)
            g4 = gold if (args.qp == 22 and not args.no_wpp and not args.frozen_contexts) else None
            (s_full, pictures, per_pic, ok), tries4 = chain_full_best(2, pair, m4, args.qp, 3, g4, len(d4))
            out["chain_full"] = {"stages": f"CTU pass -> deblocking -> entropy coder on the device -> slice data downloaded; two batches of {half} pictures in turn, a batch's pass started when the other batch's coder has queued its third stage",
                                 "value": pictures * pair[0].ctus_per_frame / s_full, "unit": "CTUs/s", "fps": pictures / s_full, "attempts_ms_per_batch": [t and t / 6 * 1e3 for t in tries4], "slice_data_bytes_per_picture": per_pic, "verified": ok,
                                 "verify": "slice data and entry points of the clip's pictures equal the reference encoder's bitstream (tests/golden/entropy.json bench-2160p)"}
            for b in pair:
                b.close()
        except Exception as e:  # auxiliary
            out["chain_full"] = {"error": repr(e)}
    if not args.no_cpu_baseline and not args.no_ref_encoder and not getattr(args, "only", ""):
        out["cpu_reference"] = cpu_reference(w, h, d4, ["--preset", "ultrafast", "-p", "1", "-q", str(args.qp)], 4)
        if out["cpu_reference"] and out["cpu_reference"].get("value"):
            out["vs_cpu_reference"] = value / out["cpu_reference"]["value"]
            if out.get("chain_full", {}).get("value"):  # like for like: both sides search, filter and code
                out["chain_full"]["vs_cpu_reference"] = out["chain_full"]["value"] / out["cpu_reference"]["value"]
    return out


def leg_tiles4k(args, lib, model_for, HipBatch, n4k=384, steps=3):
    """BASELINE config 5 at one GPU: the 4K pictures cut into kvazaar's --tiles 4x2 (8 tiles of 15x17 CTUs, no WPP inside a tile: 3 072 serial chains)"""
    import types
    w, h = 3840, 2160
    mt = model_for(args.qp, "4x2")
    targs = types.SimpleNamespace(distinct=4)
    tb, _, ctus_pf, _ = build_batches(targs, lib, 0, 1, w, h, n4k, "4x2", HipBatch)

    def tile_step():
        for b, _ in tb:
            b.launch(mt)
        for b, _ in tb:
            b.sync()
    tile_step()
    t = time.perf_counter()
    for _ in range(steps):
        tile_step()
    s = time.perf_counter() - t
    per_tile = golden_digest(w, h, clip_seed(w, h), args.qp, 0, "4x2", False)
    v = verify_batches(tb, 4, lambda tile, picture=0: (per_tile[tile] if per_tile and picture == 0 else None))
    k_s = sum(b.kernel_ms() for b, _ in tb) / 1e3
    out = {"workload": f"{w}x{h} --tiles 4x2 (BASELINE config 5 on ONE GPU; tiles imply --no-wpp as in kvazaar) all-intra ultrafast CTU pass, QP {args.qp}, "
                       f"{n4k} pictures = {8 * n4k} tile chains resident, {steps} steps", "value": steps * n4k * ctus_pf / s, "unit": "CTUs/s",
           "fps": steps * n4k / s, "kernel_ms": k_s * 1e3,
           "units_per_launch": n4k * ctus_pf, "dispatches_per_step": len(tb),
           "roofline": leg_roofline("tiles4k" if pmc_leg("tiles4k") else "intra4k", "intra_ctu_ticket_kernel", BYTES_PER_CTU, n4k * ctus_pf, k_s),
           "verified": bool(v["copies_consistent"] and v["golden_ok"] is not False), "verify": v}
    for b, _ in tb:
        b.close()
    return out


def leg_medium(args, lib, model_for, HipBatch, n_med=192):
    """BASELINE config 3: `--preset medium` (32x32 search, RDOQ, NxN partitions) at 3840x2160.  192 pictures resident: a CTU of this pass takes ~20 ms and a 4K picture
    offers ~17 of them at a time, so the 96 pictures of rounds 4-5 left a quarter of the 1 792 workgroup slots idle (96: 93.7 k CTUs/s, 144: 105 k, 192: 110 k, 384: 115 k)"""
    w, h = 3840, 2160
    d4 = synth_frames(w, h, 4, clip_seed(w, h))
    mm = model_for(args.qp)
    mm.coeff_cabac, mm.search_32x32, mm.rdoq, mm.search_nxn = 1, 1, 1, 1
    bm = HipBatch(lib, w, h, n_med)
    for i in range(n_med):
        bm.upload(i, d4[i % len(d4)])
    bm.run(mm)
    t = time.perf_counter()
    bm.run(mm)
    s = time.perf_counter() - t
    vm = verify_batches([(bm, [(i % len(d4), None) for i in range(n_med)])], len(d4),
                        lambda tile, picture=0: golden_digest(w, h, clip_seed(w, h), args.qp, 0, suffix="/medium") if picture == 0 else None)
    units = n_med * bm.ctus_per_frame
    value = units / s
    out = {"workload": f"{w}x{h} yuv420p 8-bit all-intra `--preset medium` CTU pass (32x32 CUs searched, kvz_rdoq in every quantisation, 8x8 CUs also "
                       f"as four 4x4 PUs), QP {args.qp}, {n_med} frames resident, 1 step", "value": value, "unit": "CTUs/s",
           "fps": n_med / s, "kernel_ms": bm.kernel_ms(), "units_per_launch": units,
           "roofline": leg_roofline("medium", "intra_ctu_ticket_kernel<true, true, true>", BYTES_PER_CTU, units, bm.kernel_ms() / 1e3),
           "verified": bool(vm["copies_consistent"] and vm["golden_ok"] is not False), "verify": vm,
           "matrix_cores": "every 16- and 32-point transform of this pass runs on v_mfma_i32_*_i8 (kvz_mfma.hpp); their share of the pass and the MFMA "
                           "rate of the transform kernels alone: DESIGN.md 5, bench_kernels.py"}
    bm.close()
    if not args.no_cpu_baseline and not args.no_ref_encoder and not getattr(args, "only", ""):
        out["cpu_reference"] = cpu_reference(w, h, d4, ["--preset", "medium", "-p", "1", "-q", str(args.qp)], 1, note="one picture per encoder: `medium` all-intra at 3840x2160 runs at a fraction of a picture per second and thread")
        if out["cpu_reference"] and out["cpu_reference"].get("value"):
            out["vs_cpu_reference"] = value / out["cpu_reference"]["value"]
    return out


if __name__ == "__main__":
    main()
