"""Multi-process path on CPU (gloo, world_size 2): the sharding + timing plumbing bench.py uses under
torch.distributed.run, exercised with the oracle as the per-rank worker.  Property: the CTU pass over a clip is the
same whether one process does all frames or two ranks do half each (frames are independent), checked through a
checksum of per-frame checksums gathered across ranks."""
import hashlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ctu_common as cc
import flatapi
from kvazaar_amd import sharding

W, H, NFRAMES = 64, 64, 5


def _model():
    m = cc.CostModel()
    m.lambda_, m.lambda_sqrt, m.qp, m.coeff_weights = 5.745, 5.745 ** 0.5, 22, cc.COEFF_WEIGHTS_QP22
    for arr in (m.split_flag[0], m.split_flag[1], m.split_flag[2], m.part_size, m.intra_mode, m.chroma_mode, m.cbf_luma[0], m.cbf_luma[1],
                m.cbf_chroma[0], m.cbf_chroma[1]):
        arr[0], arr[1] = 0.75, 1.5
    return m


def _frame_digest(oracle, model, frame):
    o = cc.run_oracle(oracle, model, W, H, frame)
    h = hashlib.sha256()
    for k in ("rec", "coeff", "depth", "mode", "cost"):
        h.update(o[k].tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = flatapi.load_oracle()
    model = _model()
    frames = cc.yuv_frames(W, H, NFRAMES, 4321, "small")
    lo, hi = sharding.frames_for_rank(NFRAMES, rank, world)
    digests = torch.zeros((NFRAMES, 32), dtype=torch.uint8)

    def step():
        for i in range(lo, hi):
            digests[i] = torch.from_numpy(_frame_digest(oracle, model, frames[i]))
    dt = sharding.timed_steps(step, 1, dist)
    dist.all_reduce(digests.view(torch.uint8).to(torch.int32), op=dist.ReduceOp.SUM)  # smoke: collective works on CPU
    gathered = [torch.zeros_like(digests) for _ in range(world)]
    dist.all_gather(gathered, digests)
    total = sum(g.to(torch.int32) for g in gathered).to(torch.uint8)  # shards are disjoint, the rest is zero
    if rank == 0:
        out.put((hashlib.sha256(total.numpy().tobytes()).hexdigest(), dt))
    dist.destroy_process_group()


def test_frames_for_rank_partitions():
    for n in (1, 5, 8, 96, 97):
        for world in (1, 2, 3, 8):
            spans = [sharding.frames_for_rank(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_checksum_equals_single_process(oracle):
    frames = cc.yuv_frames(W, H, NFRAMES, 4321, "small")
    model = _model()
    single = np.stack([_frame_digest(oracle, model, f) for f in frames])
    want = hashlib.sha256(single.tobytes()).hexdigest()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got, dt = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got == want and dt > 0


def test_tile_grid_matches_reference_geometry():
    """encoder.c:383-404 uniform spacing; the 4K 4x2 split of BASELINE config 5 is 8 tiles of 15x17 CTUs"""
    t = sharding.tile_grid(3840, 2160, 4, 2)
    assert len(t) == 8 and all(w == 960 for _, _, w, _ in t)
    assert [h for _, _, _, h in t] == [1088] * 4 + [1072] * 4
    assert t[5] == (960, 1088, 960, 1072)
    # 416x240 --tiles 2x2: 7x4 CTUs -> columns of 3 and 4 CTUs, rows of 2 and 2
    assert sharding.tile_grid(416, 240, 2, 2) == [(0, 0, 192, 128), (192, 0, 224, 128), (0, 128, 192, 112), (192, 128, 224, 112)]
    with pytest.raises(ValueError):
        sharding.tile_grid(128, 128, 3, 1)
    # a partition: every pixel in exactly one tile
    for (w, h, c, r) in ((1920, 1080, 3, 2), (416, 240, 2, 2), (3840, 2160, 4, 2)):
        cover = np.zeros((h, w), np.uint8)
        for x, y, tw, th in sharding.tile_grid(w, h, c, r):
            cover[y:y + th, x:x + tw] += 1
        assert (cover == 1).all()


def test_crop_and_paste_tile_round_trip():
    w, h = 192, 128
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    out = np.zeros_like(frame)
    for tile in sharding.tile_grid(w, h, 3, 2):
        sub = sharding.crop_tile(frame, w, h, tile)
        assert sub.size == tile[2] * tile[3] * 3 // 2
        sharding.paste_tile(out, w, h, tile, sub)
    assert np.array_equal(out, frame)


def _tile_worker(rank, world, port, out):
    """tile-sharded job: rank r runs the CTU pass on its tiles of every frame (sub-pictures), rank 0 assembles the frame"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = flatapi.load_oracle()
    model = _model()
    w, h = 128, 128
    frame = cc.yuv_frames(w, h, 1, 77, "small")[0]
    tiles = sharding.tile_grid(w, h, 2, 2)
    lo, hi = sharding.frames_for_rank(len(tiles), rank, world)
    rec = np.zeros(w * h * 3 // 2, dtype=np.uint8)
    for t in tiles[lo:hi]:
        o = cc.run_oracle(oracle, model, t[2], t[3], sharding.crop_tile(frame, w, h, t))
        sharding.paste_tile(rec, w, h, t, o["rec"])
    total = torch.from_numpy(rec.astype(np.int32))
    dist.all_reduce(total, op=dist.ReduceOp.SUM)  # host-side gather of the tiles (disjoint supports)
    if rank == 0:
        out.put(total.numpy().astype(np.uint8).tobytes())
    dist.destroy_process_group()


def test_two_rank_tile_sharding_equals_single_process(oracle):
    w, h = 128, 128
    frame = cc.yuv_frames(w, h, 1, 77, "small")[0]
    model = _model()
    want = np.zeros(w * h * 3 // 2, dtype=np.uint8)
    for t in sharding.tile_grid(w, h, 2, 2):
        sharding.paste_tile(want, w, h, t, cc.run_oracle(oracle, model, t[2], t[3], sharding.crop_tile(frame, w, h, t))["rec"])
    # tiles really are independent pictures: the tiled reconstruction differs from the untiled one only because prediction stops at tile edges
    untiled = cc.run_oracle(oracle, model, w, h, frame)["rec"]
    assert np.array_equal(want[:64 * w].reshape(64, w)[:, :64], untiled[:64 * w].reshape(64, w)[:, :64])  # first tile == first CTU of the frame
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert got == want.tobytes()


def _exchange_worker(rank, world, port, out, geometry):
    """the inter configuration's collective: every rank contributes its tiles' reconstruction, every rank ends up with the whole reference frame"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, cols, rows = geometry
    rng = np.random.default_rng(11)
    frame = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)  # stands for the reconstructed picture; every rank knows it only to cut ITS tiles
    plan = sharding.exchange_plan(w, h, cols, rows, world)
    mine = sharding.tiles_of_rank(len(plan["tiles"]), rank, world)
    local = {ti: torch.from_numpy(sharding.crop_tile(frame, w, h, plan["tiles"][ti])) for ti in mine}
    got = sharding.allgather_reference_frame(dist, plan, rank, world, local, w, h)
    out.put((rank, bool(np.array_equal(got.numpy(), frame)), plan["recv_bytes_per_rank"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("geometry", [(416, 240, 2, 2), (832, 480, 3, 2), (3840, 2160, 4, 2)], ids=lambda g: f"{g[0]}x{g[1]}-tiles{g[2]}x{g[3]}")
def test_two_rank_reference_frame_allgather(geometry):
    """SURVEY.md 8e, sharded inter: all_gather_into_tensor of per-rank tiles into the full reference frame on every rank (gloo here, RCCL on the
    GPUs) == the single-process frame; uneven tile sizes (960x1088 / 960x1072 at 4K) and an odd tile count (3x2 over 2 ranks) included"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q, geometry)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert [r[1] for r in res] == [True, True]
    w, h, cols, rows = geometry
    tiles = sharding.tile_grid(w, h, cols, rows)
    assert res[0][2] == (len(tiles) + 1) // 2 * sharding.tile_slot_bytes(tiles)  # one peer's slots


def test_single_process_exchange_is_identity():
    w, h = 416, 240
    plan = sharding.exchange_plan(w, h, 2, 2, 1)
    frame = np.random.default_rng(3).integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    local = {i: torch.from_numpy(sharding.crop_tile(frame, w, h, t)) for i, t in enumerate(plan["tiles"])}
    assert np.array_equal(sharding.allgather_reference_frame(None, plan, 0, 1, local, w, h).numpy(), frame)


# ---- BASELINE config 4 sharded by tile: I -> B -> B -> B of a tiled clip on two ranks, every picture through pass -> loop filters -> exchange -> next picture ----
def _tiled_inter_worker(rank, world, port, clip, out):
    """each rank: the CTU pass of its tiles (the device sources in host simulation: kvz_hostsim_intra_frame for the I picture, kvz_hostsim_inter_tile -- the inter CTU pass with
    the tile's origin in the reference FRAME -- for the B pictures), the loop filters of its tiles (the oracle's picture-level deblocking + SAO decision + SAO: loop filters do
    not cross tiles), then ReferenceExchange.exchange(): all ranks' filtered tiles -> every rank's full reference frame (and the CU records the same way) for the next picture"""
    import inter_common as ic
    import tile_common as tc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sim = tc.load_hostsim()
    pictures, records = tc.tiled_inter_chain(clip, rank, world, dist, tc.hostsim_tile_pass(sim), sim)
    if rank == 0:
        out.put(ic.digests(pictures, records))
    dist.destroy_process_group()


@pytest.mark.parametrize("clip", ["tiles2x1-pan", "tiles2x2-fast-pan-qp27"])  # (tests/golden/inter_tiles.json also holds BASELINE config 4's own clip under --tiles 4x2: bench.py's check)
def test_two_rank_tiled_inter_chain_equals_reference_encoder(clip):
    """kvazaar --tiles CxR --preset veryfast --gop lp-g4d3t1 on two ranks, one tile (or two) each: every picture's final reconstruction and CU decisions equal the reference
    encoder's (tests/golden/inter_tiles.json) -- motion vectors leave the tile into the other rank's part of the reference frame, which only the exchange provides"""
    import json
    import golden.make_golden as mg
    from test_hostsim import HOSTSIM_SO
    if not os.path.exists(HOSTSIM_SO):
        pytest.skip("tests/hostsim/libkvz_hostsim.so not built")
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inter_tiles.json")))[clip]
    spec = [c for c in mg.INTER_TILE_CLIPS if c[0] == clip][0]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_tiled_inter_worker, args=(r, 2, port, spec, out)) for r in range(2)]
    [p.start() for p in procs]
    got = None
    for _ in range(300):  # a worker that died must not leave the test waiting for its answer
        try:
            got = out.get(timeout=2)
            break
        except Exception:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    [p.join(timeout=60) for p in procs]
    [p.kill() for p in procs if p.is_alive()]
    assert got is not None, "a rank failed"
    assert got["rec"] == want["rec"], "final pictures differ from the reference encoder's"
    assert got["cu"] == want["cu"], "CU decisions differ from the reference encoder's"


def test_bench_gpus_2_self_launch_starts_two_ranks():
    """`python bench.py --gpus 2` on its own starts two ranks (torch.distributed.run, 127.0.0.1) and rank 0 prints `n_gpus` = 2 from the process group: the launch
    path with the device work stubbed (KVZ_BENCH_STUB=1: gloo, sharding.timed_steps with its barrier + MAX over ranks).  A launcher that disagrees with --gpus is refused."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KVZ_BENCH_STUB="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "KVZ_HIP_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "7", "--no-extra"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    # MAX over ranks: rank 1's step sleeps 4 ms, rank 0's 2 ms
    assert out["ms_per_step"] >= 3.9
    assert abs(out["value"] - 14 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    # world size and --gpus must agree
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE" in r2.stderr
