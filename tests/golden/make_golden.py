#!/usr/bin/env python3
"""Generates the golden fixtures of this directory from the REFERENCE ITSELF (oracle/_ref, compiled from /root/reference by
oracle/Makefile) -- run in the build container, where /root/reference exists:

    make -C oracle ref && python tests/golden/make_golden.py

Fixtures (small, committed; the GPU box and any machine without the reference tree check against them):
  strategy_cases.json   sha256 of the reference's GENERIC strategy output for every seeded case of tests/cases.py
                        (label -> digest of repr(outputs)), incl. tests/cases.py cases_find_last_scanpos
  deblock.json          sha256 of kvz_filter_deblock_lcu's result (filter.c:783, all LCUs) for seeded pictures / CU quadtrees
  sao_frame.json        sha256 of kvz_sao_reconstruct's result (sao.c:302-361, every CTU and plane) for seeded pictures / parameters
  model_constants.json  kvz_f_entropy_bits (rdo.c:83) and kvz_fast_coeff_get_weights per QP (fast_coeff_cost.c:84-88) as the reference build
                        returns them: the cost-model inputs of tests that run without the reference tree
  encoder_recon.json    the reference ENCODER end to end: sha256 of the reconstruction `kvazaar --preset ultrafast -p 1 -q QP` (the CLI
                        built from /root/reference, all-intra, deblocking on / off) writes with --debug for seeded clips -- what the
                        batched CTU pass (+ deblocking) must reproduce picture for picture; ".../cu" entries: digests of the CU depth
                        and intra mode maps of the encoder's cu_array behind it (recorded with the oracle/ref_cudump.c interposer)
  entropy_inter.json    (--entropy-inter) the same for low-delay sequences (I and B pictures) of tests/inter_common.py ENTROPY_CASES
  entropy.json          (--entropy) the slice data of the reference encoder's bitstreams for tests/entropy_common.py CASES: digest and substream sizes per picture
(the bitstream md5s of the reference encoder are asserted by tests/test_e2e_dropin.py)
tests/test_oracle_golden.py holds the known answers of the reference's own unit tests; this file adds outputs of the
compiled reference for the functions those tests do not pin."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import cases  # noqa: E402
import deblock_common as dc  # noqa: E402
import flatapi  # noqa: E402


def digest(outputs):
    return hashlib.sha256(repr(outputs).encode()).hexdigest()[:24]


def strategy_digests(lib, scan_table):
    out = {}
    for label, run in cases.all_cases():
        r = run(lib)
        if label.startswith("pixel_var"):
            r = tuple(float(np.float64(v)).hex() for v in r)  # the one floating-point output: exact bits of the generic result
        assert label not in out, label
        out[label] = digest(r)
    for label, run in cases.cases_find_last_scanpos(scan_table):
        out["find_last_scanpos/" + label] = digest(run(lib))
    return out


DEBLOCK_CASES = [(64, 64, 1), (192, 136, 2), (416, 240, 3)]


def deblock_digests(func):
    out = {}
    for (w, h, seed) in DEBLOCK_CASES:
        rng = np.random.default_rng(seed)
        for kind in ("smooth", "steps", "noise"):
            frame, _ = dc.test_picture(w, h, rng, kind)
            depth = dc.random_depth_map(w, h, rng)
            for qp, b_off, t_off in ((22, 0, 0), (37, 1, -2)):
                res = dc.run_cpu(func, w, h, qp, b_off, t_off, frame, depth)
                out[f"{w}x{h}/{kind}/qp{qp}/b{b_off}/t{t_off}"] = hashlib.sha256(res.tobytes()).hexdigest()[:24]
    return out


def sao_digests(func):
    import sao_common as sc
    out = {}
    for (w, h, seed) in ((64, 64, 11), (136, 72, 12), (416, 240, 13)):
        rng = np.random.default_rng(seed)
        n = ((w + 63) // 64) * ((h + 63) // 64)
        for trial in range(3):
            frame = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
            luma, chroma = sc.random_params(rng, n, False), sc.random_params(rng, n, True)
            out[f"{w}x{h}/{trial}"] = hashlib.sha256(sc.run_cpu(func, w, h, frame, luma, chroma).tobytes()).hexdigest()[:24]
    return out


# (width, height, frames, seed, kind, qp): sizes with partial CTUs; QPs on both sides of fast_residual_cost_limit (28 in `ultrafast`,
# cfg.c:485-512): below it coefficients are priced with the fast estimate, from it on with the CABAC model (rdo.c:311-340)
ENCODER_CLIPS = [(64, 64, 2, 9, "small", 22), (72, 88, 2, 1, "small", 12), (200, 136, 2, 3, "small", 27), (64, 64, 2, 9, "small", 30),
                 (200, 136, 2, 3, "small", 37), (416, 240, 3, 1234, "small", 22), (416, 240, 2, 99, "small", 17), (416, 240, 2, 7, "small", 32),
                 (192, 136, 2, 5, "small", 45), (832, 480, 1, 5, "large", 22), (832, 480, 1, 5, "large", 28), (1920, 1080, 1, 1, "large", 22),
                 (1920, 1080, 1, 1, "large", 32),
                 # BASELINE configs 3-5 geometry (3840x2160 = 60x34 CTUs, partial bottom row), the size the north-star target is stated on
                 (3840, 2160, 1, 2, "large", 22), (3840, 2160, 1, 2, "large", 32)]


# ... and without wavefront parallel processing (--no-wpp; also what --tiles implies, cfg.c:925-978): one coder through the picture
ENCODER_CLIPS_NO_WPP = [(416, 240, 2, 1234, "small", 22), (200, 136, 2, 3, "small", 32), (64, 136, 1, 3, "small", 22),
                        (1920, 1080, 1, 3, "large", 22), (3840, 2160, 1, 3, "large", 22)]  # the longest serial CTU chains the schedule sees

# (width, height, frames, seed, kind, qp, tiles, wpp): kvazaar's uniform tile grid (encoder.c:383-404); tiles are independent sub-pictures
# with their own coder (encoderstate.c:944-979), WPP off unless asked for (cfg.c:925-978).  The last two are BASELINE config 5's real
# geometry: 8 tiles of 15x17 CTUs (960x1088 / 960x1072), chains of 255 CTUs without WPP.
ENCODER_CLIPS_TILES = [(416, 240, 2, 1234, "small", 22, "2x2", False), (416, 240, 1, 1234, "small", 32, "2x2", True), (832, 480, 1, 5, "large", 22, "3x2", False),
                       (3840, 2160, 1, 2, "large", 22, "4x2", False), (3840, 2160, 1, 2, "large", 22, "4x2", True)]


# (width, height, frames, seed, kind, qp, no_wpp): `--sao full` on top of the all-intra ultrafast configuration: the SAO parameter decision
# (sao.c:671) runs per LCU right after that LCU's deblocking; the digest is the final (post-SAO) reconstruction
ENCODER_CLIPS_SAO = [(64, 64, 2, 9, "small", 22, False), (200, 136, 2, 3, "small", 27, False), (416, 240, 2, 1234, "small", 22, False), (416, 240, 2, 7, "small", 32, False),
                     (192, 136, 2, 5, "small", 45, False), (416, 240, 1, 1234, "small", 22, True), (832, 480, 1, 5, "large", 22, False), (1920, 1080, 1, 1, "large", 22, False),
                     (1920, 1080, 1, 1, "large", 37, False),
                     (192, 136, 4, 0, "adversarial", 32, False), (192, 136, 4, 0, "adversarial", 40, False)]  # band SAO is only chosen on these


# (width, height, frames, seed, kind, qp): `--preset fast -p 1` = --pu-depth-intra 1-3 (32x32 CUs searched), fast-residual-cost 0 (CABAC coefficient cost at every
# QP), --sao full; digests of the reconstruction before the loop filters (--no-deblock --sao off on the command line), after deblocking, and the final picture
ENCODER_CLIPS_FAST = [(64, 64, 2, 9, "small", 22), (72, 88, 2, 1, "small", 12), (200, 136, 2, 3, "small", 27), (416, 240, 2, 1234, "small", 22), (416, 240, 1, 7, "small", 32),
                      (192, 136, 4, 0, "adversarial", 32), (832, 480, 1, 5, "large", 22), (1920, 1080, 1, 1, "large", 22), (1920, 1080, 1, 1, "large", 37)]


# (width, height, frames, seed, kind, qp): `--preset medium --pu-depth-intra 1-3 -p 1` = preset `fast` + --rdoq (kvz_rdoq in every quantisation; `medium` itself also
# searches the 4x4 NxN partitions, pu-depth-intra 1-4, which the pass does not have yet); the same three stages as ENCODER_CLIPS_FAST
ENCODER_CLIPS_RDOQ = [(64, 64, 2, 9, "small", 22), (72, 88, 2, 1, "small", 12), (200, 136, 2, 3, "small", 27), (416, 240, 2, 1234, "small", 22), (416, 240, 1, 7, "small", 32),
                      (192, 136, 4, 0, "adversarial", 32), (832, 480, 1, 5, "large", 22), (1920, 1080, 1, 1, "large", 27)]


# (width, height, frames, seed, kind, qp): `--preset medium -p 1` as it is -- pu-depth-intra 1-4: 8x8 CUs are also tried as four 4x4 PUs (part_size NxN) -- on clips
# where that partition is taken (noise at QP 12: every CU; the synthetic scenes at QP 12-27: a few per cent of the 8x8 CUs); the same three stages
ENCODER_CLIPS_MEDIUM = [(64, 64, 2, 9, "small", 22), (72, 88, 2, 1, "small", 12), (200, 136, 2, 3, "small", 27), (416, 240, 2, 1234, "small", 22),
                        (192, 136, 4, 0, "adversarial", 12), (192, 136, 4, 0, "adversarial", 32), (832, 480, 1, 5, "large", 17), (1920, 1080, 1, 1, "large", 27),
                        (3840, 2160, 1, 2, "large", 22)]  # BASELINE config 3 at its own size (checked on the GPU; the oracle needs minutes for it)


# (width, height, frames, seed, kind, qp): the pictures bench.py keeps resident -- the first 8 (1080p) / 4 (4K) frames of SURVEY.md App. C's clips, ALL of them: the bench
# hashes one copy of every distinct picture of its batch against these (reconstruction before the loop filters, and deblocked)
ENCODER_CLIPS_BENCH = [(1920, 1080, 8, 1, "large", 22), (3840, 2160, 4, 2, "large", 22)]


def clip_key(w, h, n, seed, kind, qp, deblock, no_wpp=False, tiles=None, wpp=False):
    return (f"{w}x{h}/n{n}/seed{seed}/{kind}/qp{qp}/{'deblock' if deblock else 'nodeblock'}" + ("/nowpp" if no_wpp else "")
            + (f"/tiles{tiles}" + ("-wpp" if wpp else "") if tiles else ""))


def reference_encoder_recon(w, h, frames, qp, deblock, workdir, cu_maps=None, no_wpp=False, tiles=None, wpp=False, sao=False, preset="ultrafast", extra=()):
    """runs the reference CLI (oracle/_ref/kvazaar_ref) on the clip; returns its --debug reconstruction, one array per frame.
    cu_maps: a list that receives, per frame, the (depth, intra mode) maps per 8x8 cell the encoder's search left in its cu_array
    (recorded through the oracle/ref_cudump.c interposer; single-threaded so that LCUs arrive frame by frame)"""
    exe = os.path.join(flatapi.ROOT, "oracle", "_ref", "kvazaar_ref")
    src, rec = os.path.join(workdir, "in.yuv"), os.path.join(workdir, "rec.yuv")
    with open(src, "wb") as f:
        f.write(b"".join(fr.tobytes() for fr in frames))
    cmd = [exe, "-i", src, "--input-res", f"{w}x{h}", "--preset", preset, "-p", "1", "-q", str(qp), "--debug", rec, "-o", os.path.join(workdir, "out.hevc")] + list(extra)
    if not deblock:
        cmd.append("--no-deblock")
    if no_wpp:
        cmd.append("--no-wpp")
    if tiles:
        cmd += ["--tiles", tiles] + (["--wpp"] if wpp else [])
    if sao:
        cmd += ["--sao", "full"]
    elif preset != "ultrafast":
        cmd += ["--sao", "off"]
    env = dict(os.environ)
    dump = os.path.join(workdir, "cu.txt")
    if cu_maps is not None:
        if os.path.exists(dump):
            os.remove(dump)
        cmd += ["--threads", "0", "--owf", "0"]
        env.update(LD_PRELOAD=os.path.join(flatapi.ROOT, "oracle", "_ref", "libkvz_cudump.so"), KVZ_CUDUMP=dump)
    subprocess.run(cmd, check=True, capture_output=True, env=env)
    if cu_maps is not None:
        cells = (w // 8) * (h // 8)
        rows = np.loadtxt(dump, dtype=np.int64).reshape(len(frames), cells, 5)
        for fr in rows:
            depth, mode = np.zeros((h // 8, w // 8), np.uint8), np.zeros((h // 8, w // 8), np.uint8)
            depth[fr[:, 1] // 8, fr[:, 0] // 8] = fr[:, 2]
            mode[fr[:, 1] // 8, fr[:, 0] // 8] = fr[:, 3]
            cu_maps.append((depth, mode))
    return list(np.fromfile(rec, dtype=np.uint8).reshape(len(frames), -1))


def encoder_digests(workdir):
    import ctu_common as cc
    out = {}
    for (w, h, n, seed, kind, qp) in ENCODER_CLIPS:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        for deblock in (0, 1):
            maps = [] if deblock == 0 else None
            recs = reference_encoder_recon(w, h, frames, qp, deblock, workdir, maps)
            out[clip_key(w, h, n, seed, kind, qp, deblock)] = [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in recs]
            if maps is not None:  # the CU quadtree and the intra modes behind that reconstruction
                out[clip_key(w, h, n, seed, kind, qp, deblock) + "/cu"] = [cu_digest(d, m) for d, m in maps]
    for (w, h, n, seed, kind, qp) in ENCODER_CLIPS_NO_WPP:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        recs = reference_encoder_recon(w, h, frames, qp, 0, workdir, None, True)
        out[clip_key(w, h, n, seed, kind, qp, 0, True)] = [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in recs]
    for (w, h, n, seed, kind, qp, no_wpp) in ENCODER_CLIPS_SAO:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        recs = reference_encoder_recon(w, h, frames, qp, 1, workdir, None, no_wpp, None, False, True)
        out[clip_key(w, h, n, seed, kind, qp, 1, no_wpp) + "/sao"] = [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in recs]
    for (w, h, n, seed, kind, qp) in ENCODER_CLIPS_FAST:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        for stage, (deblock, sao) in (("nodeblock", (0, False)), ("deblock", (1, False)), ("sao", (1, True))):
            recs = reference_encoder_recon(w, h, frames, qp, deblock, workdir, None, False, None, False, sao, "fast")
            out[clip_key(w, h, n, seed, kind, qp, deblock) + "/fast" + ("/sao" if sao else "")] = [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in recs]
    for (w, h, n, seed, kind, qp) in ENCODER_CLIPS_RDOQ:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        for stage, (deblock, sao) in (("nodeblock", (0, False)), ("deblock", (1, False)), ("sao", (1, True))):
            recs = reference_encoder_recon(w, h, frames, qp, deblock, workdir, None, False, None, False, sao, "medium", ("--pu-depth-intra", "1-3"))
            out[clip_key(w, h, n, seed, kind, qp, deblock) + "/medium-pu13" + ("/sao" if sao else "")] = [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in recs]
    out.update(medium_digests(workdir))
    out.update(bench_digests(workdir))
    for (w, h, n, seed, kind, qp, tiles, wpp) in ENCODER_CLIPS_TILES:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        for deblock in (0, 1):
            recs = reference_encoder_recon(w, h, frames, qp, deblock, workdir, None, False, tiles, wpp)
            out[clip_key(w, h, n, seed, kind, qp, deblock, False, tiles, wpp)] = [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in recs]
            # ... and tile by tile (raster tile order), so that a rank holding some of the tiles can check its share (bench.py --tiles)
            from kvazaar_amd import sharding
            grid = sharding.tile_grid(w, h, *(int(v) for v in tiles.split("x")))
            out[clip_key(w, h, n, seed, kind, qp, deblock, False, tiles, wpp) + "/per-tile"] = [
                [hashlib.sha256(sharding.crop_tile(r, w, h, t).tobytes()).hexdigest()[:24] for t in grid] for r in recs]
    return out


def bench_digests(workdir):
    """the ENCODER_CLIPS_BENCH entries of encoder_recon.json (callable on its own: `python -c "import make_golden as m; m.update_bench()"`)"""
    import ctu_common as cc
    out = {}
    for (w, h, n, seed, kind, qp) in ENCODER_CLIPS_BENCH:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        for deblock in (0, 1):
            recs = reference_encoder_recon(w, h, frames, qp, deblock, workdir)
            out[clip_key(w, h, n, seed, kind, qp, deblock)] = [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in recs]
    return out


def update_bench():
    import tempfile
    path = os.path.join(HERE, "encoder_recon.json")
    data = json.load(open(path))
    with tempfile.TemporaryDirectory() as d:
        data.update(bench_digests(d))
    json.dump(data, open(path, "w"), indent=0, sort_keys=True)


def medium_digests(workdir):
    """the ENCODER_CLIPS_MEDIUM entries of encoder_recon.json (callable on its own: `python -c "import make_golden as m; m.update_medium()"`)"""
    import ctu_common as cc
    out = {}
    for (w, h, n, seed, kind, qp) in ENCODER_CLIPS_MEDIUM:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        for stage, (deblock, sao) in (("nodeblock", (0, False)), ("deblock", (1, False)), ("sao", (1, True))):
            maps = [] if stage == "nodeblock" else None
            recs = reference_encoder_recon(w, h, frames, qp, deblock, workdir, maps, False, None, False, sao, "medium")
            out[clip_key(w, h, n, seed, kind, qp, deblock) + "/medium" + ("/sao" if sao else "")] = [hashlib.sha256(r.tobytes()).hexdigest()[:24] for r in recs]
            if maps is not None:  # CU depth and the first PU's intra mode per 8x8 cell
                out[clip_key(w, h, n, seed, kind, qp, deblock) + "/medium/cu"] = [cu_digest(d, m) for d, m in maps]
    return out


def update_medium():
    import tempfile
    path = os.path.join(HERE, "encoder_recon.json")
    data = json.load(open(path))
    with tempfile.TemporaryDirectory() as d:
        data.update(medium_digests(d))
    json.dump(data, open(path, "w"), indent=0, sort_keys=True)


def cu_digest(depth, mode):
    return hashlib.sha256(np.ascontiguousarray(depth, np.uint8).tobytes() + np.ascontiguousarray(mode, np.uint8).tobytes()).hexdigest()[:24]


def update_inter():
    """tests/golden/inter_recon.json: the reference encoder (oracle/_ref/kvazaar_ref, --preset ... --gop lp-g4d3t1, --threads 0 so that the ref_cudump.c
    interposer sees the LCUs in order) on the clips of tests/inter_common.py CASES: digest of every picture's --debug reconstruction and of its CU decisions,
    md5 of the bitstream"""
    import tempfile
    import inter_common as ic
    out = {}
    for case in ic.CASES:
        name, w, h, n, qp, preset, dbk, sao, owf, src = case
        frames = ic.case_frames(case)
        with tempfile.TemporaryDirectory() as d:
            rec, cu = ic.reference_encode(w, h, frames, qp, d, preset=preset, deblock=bool(dbk), sao=bool(sao), owf=owf)
            out[name] = dict(ic.digests(rec, cu), bitstream_md5=hashlib.md5(open(os.path.join(d, "out.hevc"), "rb").read()).hexdigest(),
                             clip_md5=hashlib.md5(b"".join(f.tobytes() for f in frames)).hexdigest())
        print(name, out[name]["bitstream_md5"], flush=True)
    json.dump(out, open(os.path.join(HERE, "inter_recon.json"), "w"), indent=0, sort_keys=True)


# the tiled inter configuration (BASELINE config 4 sharded by tile): (name, width, height, pictures, qp, tiles, clip seed, noise, pan)
INTER_TILE_CLIPS = [("tiles2x1-pan", 256, 128, 4, 22, "2x1", 21, 1.5, (1.25, -0.5)), ("tiles2x2-fast-pan-qp27", 256, 256, 4, 27, "2x2", 23, 2.0, (-5.0, 11.0))]


def update_inter_tiles():
    """tests/golden/inter_tiles.json: the reference encoder with `--tiles CxR --preset veryfast --gop lp-g4d3t1` (--threads 0 for the ref_cudump.c interposer, which writes
    frame positions): digests of every picture's final reconstruction and CU decisions.  tests/test_dist_cpu.py chains I -> B -> B -> B tile by tile on two ranks against it."""
    import tempfile
    import inter_common as ic
    out = {}
    clips = list(INTER_TILE_CLIPS)
    if "--with-4k" in sys.argv:  # BASELINE config 4's own clip under --tiles 4x2 (what `bench.py --preset veryfast-inter --tiles 4x2` verifies): minutes of the reference encoder
        clips.append(("baseline-c4-2160p-tiles4x2", 3840, 2160, 4, 22, "4x2", None, None, None))
    else:
        try:
            out.update({k: v for k, v in json.load(open(os.path.join(HERE, "inter_tiles.json"))).items() if k.startswith("baseline-")})
        except (OSError, ValueError):
            pass
    for (name, w, h, n, qp, tiles, seed, noise, pan) in clips:
        frames = ic.clip(w, h, n, seed, noise, pan) if seed is not None else ic.case_frames([c for c in ic.CASES if c[0] == "baseline-c4-2160p"][0])
        with tempfile.TemporaryDirectory() as d:
            rec, cu = ic.reference_encode(w, h, frames, qp, d, preset="veryfast", deblock=True, sao=True, owf=0, extra=("--tiles", tiles))
            out[name] = dict(ic.digests(rec, cu), bitstream_md5=hashlib.md5(open(os.path.join(d, "out.hevc"), "rb").read()).hexdigest())
        print(name, out[name]["bitstream_md5"], flush=True)
    json.dump(out, open(os.path.join(HERE, "inter_tiles.json"), "w"), indent=0, sort_keys=True)


def update_entropy():
    """tests/golden/entropy.json: the slice data of the reference encoder's bitstreams for tests/entropy_common.py CASES.  Per picture: sha256 of the slice NAL's payload
    behind the slice header (the bytes are taken from the REFERENCE bitstream; where the header ends follows from the substream sizes, which the header's own entry
    points must confirm -- asserted here), the substream sizes and the header bytes in front"""
    import tempfile
    import entropy_common as ec
    oracle = flatapi.load_oracle()
    out = {}
    only = [a for a in sys.argv[sys.argv.index("--entropy") + 1:] if not a.startswith("-")]  # `--entropy name ...`: (re)make these cases only, keep the others as they are
    if only:
        out = json.load(open(os.path.join(HERE, "entropy.json")))
    for case in ec.CASES + ec.BENCH_CASES:
        if only and case[0] not in only:
            continue
        with tempfile.TemporaryDirectory() as d:
            payloads = ec.reference_slice_payloads(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "kvazaar_ref"), case, d)
        pictures = []
        for payload, (data, sizes) in zip(payloads, ec.oracle_slice_data(oracle, case)):
            total = sum(sizes)
            ref_data, header = payload[len(payload) - total:], payload[:len(payload) - total]
            assert ec.header_ends_with_entry_points(header, sizes, "--no-wpp" not in case[8]), (case[0], "the slice header's entry points are not these substream sizes")
            assert ref_data == data, (case[0], "oracle/kvz_oracle_entropy.inc does not reproduce the reference's slice data")
            pictures.append({"sha": hashlib.sha256(ref_data).hexdigest()[:24], "sizes": sizes, "header": header.hex()})
        out[case[0]] = pictures
        print(case[0], [p["sizes"] for p in pictures], flush=True)
    json.dump(out, open(os.path.join(HERE, "entropy.json"), "w"), indent=0, sort_keys=True)


def update_entropy_inter():
    """tests/golden/entropy_inter.json: the slice data of the reference encoder's low-delay bitstreams (I and B pictures) for the small cases of tests/inter_common.py;
    taken from the reference's bytes, the split asserted as in update_entropy"""
    import tempfile
    import entropy_common as ec
    import inter_common as ic
    oracle = flatapi.load_oracle()
    out = {}
    for case in ic.CASES:
        name, w, h, n, qp, preset, dbk, sao, owf, src = case
        if name not in ic.ENTROPY_CASES + ic.ENTROPY_BENCH_CASES:
            continue
        frames = ic.case_frames(case)
        with tempfile.TemporaryDirectory() as d:
            ic.reference_encode(w, h, frames, qp, d, preset=preset, deblock=bool(dbk), sao=bool(sao), owf=owf, cu=False)
            payloads = ec.slice_payloads(open(os.path.join(d, "out.hevc"), "rb").read())
        pictures = []
        for payload, (data, sizes) in zip(payloads, ic.oracle_encode_bits(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)):
            total = sum(sizes)
            ref_data, header = payload[len(payload) - total:], payload[:len(payload) - total]
            assert ec.header_ends_with_entry_points(header, sizes, True), (name, "entry points")
            assert ref_data == data, (name, "slice data")
            pictures.append({"sha": hashlib.sha256(ref_data).hexdigest()[:24], "sizes": sizes})
        out[name] = pictures
        print(name, [sum(p["sizes"]) for p in pictures], flush=True)
    json.dump(out, open(os.path.join(HERE, "entropy_inter.json"), "w"), indent=0, sort_keys=True)


def main():
    if "--entropy-inter" in sys.argv:
        return update_entropy_inter()
    if "--inter-tiles" in sys.argv:
        return update_inter_tiles()
    if "--inter" in sys.argv:
        return update_inter()
    if "--entropy" in sys.argv:
        return update_entropy()
    ref = flatapi.load_ref(0)  # generic strategies
    oracle = flatapi.load_oracle()

    def scan_table(scan_idx, l2):
        n = 1 << (2 * l2)
        return np.ctypeslib.as_array(oracle.lib.kvz_oracle_scan_table(scan_idx, l2), shape=(n,)).copy()

    json.dump(strategy_digests(ref, scan_table), open(os.path.join(HERE, "strategy_cases.json"), "w"), indent=0, sort_keys=True)
    if "--strategy" in sys.argv:  # the per-function cases only (new cases in tests/cases.py): seconds
        return print("wrote strategy_cases.json")
    json.dump(deblock_digests(ref.lib.kvz_ref_deblock_frame), open(os.path.join(HERE, "deblock.json"), "w"), indent=0, sort_keys=True)
    json.dump(sao_digests(ref.lib.kvz_ref_sao_frame), open(os.path.join(HERE, "sao_frame.json"), "w"), indent=0, sort_keys=True)
    json.dump({"entropy_fbits": [float(ref.lib.kvz_ref_entropy_fbits(i)) for i in range(128)],
               "coeff_weights": {str(qp): int(ref.lib.kvz_ref_fast_coeff_weights(qp)) for qp in range(52)}},
              open(os.path.join(HERE, "model_constants.json"), "w"), sort_keys=True)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        json.dump(encoder_digests(d), open(os.path.join(HERE, "encoder_recon.json"), "w"), indent=0, sort_keys=True)
    print("wrote strategy_cases.json, deblock.json, sao_frame.json, model_constants.json, encoder_recon.json")


if __name__ == "__main__":
    main()
