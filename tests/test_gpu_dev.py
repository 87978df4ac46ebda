"""Parity of the device-resident batch entry points (include/kvz_hip_dev.h) with the oracle, block by block, bit-exact."""
import ctypes as C

import numpy as np
import pytest

from kvazaar_amd import dev as devapi
import flatapi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import kvazaar_amd
    return devapi.Dev(kvazaar_amd.load_library())


def blocks(rng, n, count, kind):
    if kind == "noise":
        return rng.integers(0, 256, (count, n * n), dtype=np.uint8)
    if kind == "extreme":
        return rng.choice(np.array([0, 255], dtype=np.uint8), (count, n * n))
    base = rng.integers(0, 256, (count, 1), dtype=np.int32) + np.arange(n * n, dtype=np.int32)[None, :] % n  # speed_tests.c style gradient
    return np.clip(base, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("n", [8, 16, 32, 64])
@pytest.mark.parametrize("cost", ["sad", "satd"])
def test_dev_cost_nxn(oracle, dev, cost, n):
    rng = np.random.default_rng(100 + n)
    count = 333  # not a multiple of anything: exercises the ragged last workgroup
    for kind in ("noise", "extreme", "gradient"):
        a, b = blocks(rng, n, count, kind), blocks(rng, n, count, "noise")
        da, db, do = dev.put(a), dev.put(b), dev.empty(4 * count)
        getattr(dev.lib, f"kvz_hip_dev_{cost}_nxn")(n, da, db, count, do)
        got = dev.get(do, (count,), np.uint32)
        f = getattr(oracle, f"{cost}_nxn")
        want = np.array([f(n, flatapi.ptr(a[i]), flatapi.ptr(b[i])) for i in range(count)], dtype=np.uint32)
        dev.free(da, db, do)
        assert np.array_equal(got, want), (kind, np.flatnonzero(got != want)[:8])


def test_dev_satd_4x4(oracle, dev):
    rng = np.random.default_rng(4)
    count = 1001
    a, b = blocks(rng, 4, count, "noise"), blocks(rng, 4, count, "extreme")
    da, db, do = dev.put(a), dev.put(b), dev.empty(4 * count)
    dev.lib.kvz_hip_dev_satd_nxn(4, da, db, count, do)
    got = dev.get(do, (count,), np.uint32)
    want = np.array([oracle.satd_nxn(4, flatapi.ptr(a[i]), flatapi.ptr(b[i])) for i in range(count)], dtype=np.uint32)
    dev.free(da, db, do)
    assert np.array_equal(got, want)


def test_dev_empty_batch(dev):
    dev.lib.kvz_hip_dev_sad_nxn(8, None, None, 0, None)
    dev.lib.kvz_hip_dev_satd_nxn(16, None, None, 0, None)
    dev.lib.kvz_hip_dev_sync()


@pytest.mark.parametrize("name", sorted(devapi.TRANSFORM_KINDS))
@pytest.mark.parametrize("mfma", [0, 1, 2])
def test_dev_transform(oracle, dev, name, mfma):
    kind = devapi.TRANSFORM_KINDS[name]
    n = devapi.TRANSFORM_SIZE[kind]
    rng = np.random.default_rng(kind)
    count = 67
    x = np.empty((count, n * n), dtype=np.int16)
    x[:20] = rng.integers(-255, 256, (20, n * n))            # residuals
    x[20:40] = rng.integers(-32768, 32768, (20, n * n))      # full range: exercises the int16 wrap / clip of both passes
    x[40:] = rng.choice(np.array([-32768, 32767, 0], dtype=np.int16), (count - 40, n * n))
    di, dt, do = dev.put(x), dev.empty(x.nbytes), dev.empty(x.nbytes)
    dev.lib.kvz_hip_dev_transform(kind, di, dt, do, count, mfma)
    got = dev.get(do, x.shape, np.int16)
    want = np.empty_like(x)
    for i in range(count):
        oracle.transform(kind, 8, flatapi.ptr(x[i]), flatapi.ptr(want[i]))
    dev.free(di, dt, do)
    assert np.array_equal(got, want), np.argwhere(got != want)[:8]


@pytest.mark.parametrize("log2w", [2, 3, 4, 5])
def test_dev_angular(oracle, dev, log2w):
    rng = np.random.default_rng(log2w)
    w, count = 1 << log2w, 203
    above = rng.integers(0, 256, (count, 2 * w + 1), dtype=np.uint8)
    left = rng.integers(0, 256, (count, 2 * w + 1), dtype=np.uint8)
    left[:, 0] = above[:, 0]
    for mode in range(2, 35):
        da, dl, do = dev.put(above), dev.put(left), dev.empty(count * w * w)
        dev.lib.kvz_hip_dev_angular_pred(log2w, mode, da, dl, count, do)
        got = dev.get(do, (count, w * w), np.uint8)
        want = np.empty_like(got)
        for i in range(count):
            oracle.angular_pred(log2w, mode, flatapi.ptr(above[i]), flatapi.ptr(left[i]), flatapi.ptr(want[i]))
        dev.free(da, dl, do)
        assert np.array_equal(got, want), mode


@pytest.mark.parametrize("size", [(64, 64), (192, 136), (416, 240), (1920, 1080)])
def test_dev_deblock_frames(oracle, dev, size):
    """device deblocking of a batch of frames == the oracle's kvz_filter_deblock_lcu restatement (itself pinned against the
    compiled reference in tests/test_oracle_vs_ref.py), random CU quadtrees"""
    import ctypes as C
    import deblock_common as dc
    w, h = size
    rng = np.random.default_rng(w + h)
    kinds = ("smooth", "steps", "noise", "steps")
    frames, depths = [], []
    for k in kinds:
        f, _ = dc.test_picture(w, h, rng, k)
        frames.append(f)
        depths.append(dc.random_depth_map(w, h, rng))
    for qp, b_off, t_off in ((22, 0, 0), (37, 2, -1), (51, 0, 3)):
        dfr, ddp = dev.put(np.stack(frames)), dev.put(np.stack(depths))
        dev.lib.kvz_hip_dev_deblock_frames.restype = None
        dev.lib.kvz_hip_dev_deblock_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        dev.lib.kvz_hip_dev_deblock_frames(dfr, w, h, len(frames), ddp, qp, b_off, t_off)
        got = dev.get(dfr, (len(frames), w * h * 3 // 2), np.uint8)
        dev.free(dfr, ddp)
        for i in range(len(frames)):
            want = dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, w, h, qp, b_off, t_off, frames[i], depths[i])
            assert np.array_equal(got[i], want), (kinds[i], qp, np.flatnonzero(got[i] != want)[:8])


def test_dev_picture_checksums(oracle, dev):
    """kvz_hip_dev_picture_checksums over a batch == the oracle's per-plane checksum (nal-generic.c:57-82) of every frame; sizes
    whose planes cross the 256-sample mask periods and whose wavefronts straddle plane / frame boundaries"""
    import ctypes as C
    rng = np.random.default_rng(12)
    for (w, h, n) in ((8, 8, 3), (72, 40, 5), (416, 240, 4), (1920, 1080, 2)):
        frames = rng.integers(0, 256, (n, w * h * 3 // 2), dtype=np.uint8)
        dfr, dout = dev.put(frames), dev.empty(4 * 3 * n)
        dev.lib.kvz_hip_dev_picture_checksums.restype = None
        dev.lib.kvz_hip_dev_picture_checksums.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        dev.lib.kvz_hip_dev_picture_checksums(dfr, w, h, n, dout)
        got = dev.get(dout, (n, 3), np.uint32)
        dev.free(dfr, dout)
        for f in range(n):
            y, u, v = frames[f][:w * h], frames[f][w * h:w * h * 5 // 4], frames[f][w * h * 5 // 4:]
            want = [oracle.plane_checksum(flatapi.ptr(y), h, w, w), oracle.plane_checksum(flatapi.ptr(u), h // 2, w // 2, w // 2),
                    oracle.plane_checksum(flatapi.ptr(v), h // 2, w // 2, w // 2)]
            assert list(got[f]) == want, (w, h, f)


@pytest.mark.parametrize("bw,rng_", [(8, 4), (16, 16), (32, 9), (64, 32)])
def test_dev_sad_surface(oracle, dev, bw, rng_):
    """every candidate of the motion cost surface == kvz_image_calc_sad (edge-replicated reference), blocks in the picture
    interior and touching all four borders / corners so that the window leaves the frame"""
    import ctypes as C
    w, h = 200, 136
    rng = np.random.default_rng(bw)
    cur = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ref = np.roll(cur, (3, -5), (0, 1)) ^ rng.integers(0, 4, (h, w), dtype=np.uint8)
    pos = [(0, 0), (w - bw, 0), (0, h - bw), (w - bw, h - bw), ((w - bw) // 2 // 8 * 8, (h - bw) // 2 // 8 * 8), (8, h - bw), (w - bw, 8)]
    xy = np.array(pos, dtype=np.int16)
    side = 2 * rng_ + 1
    dc, dr, dxy, dout = dev.put(cur), dev.put(ref), dev.put(xy), dev.empty(4 * len(pos) * side * side)
    dev.lib.kvz_hip_dev_sad_surface.restype = None
    dev.lib.kvz_hip_dev_sad_surface.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    dev.lib.kvz_hip_dev_sad_surface(dc, dr, w, h, bw, rng_, dxy, len(pos), dout)
    got = dev.get(dout, (len(pos), side, side), np.uint32)
    dev.free(dc, dr, dxy, dout)
    picks = [(-rng_, -rng_), (rng_, rng_), (0, 0), (-rng_, rng_), (1, -2), (rng_ // 2, -rng_)] + [tuple(rng.integers(-rng_, rng_ + 1, 2)) for _ in range(40)]
    for b, (x, y) in enumerate(pos):
        for dx, dy in picks:
            want = oracle.image_calc_sad(flatapi.ptr(cur), w, flatapi.ptr(ref), w, h, w, x, y, x + int(dx), y + int(dy), bw, bw)
            assert got[b, dy + rng_, dx + rng_] == want, (b, dx, dy)


@pytest.mark.parametrize("size", [(64, 64), (136, 72), (416, 240), (1920, 1080)])
def test_dev_sao_frames(oracle, dev, size):
    """device SAO over a batch of frames == the oracle's kvz_sao_reconstruct restatement (pinned against the compiled reference
    in tests/test_oracle_vs_ref.py), random per-CTU parameters"""
    import ctypes as C
    import sao_common as sc
    w, h = size
    rng = np.random.default_rng(2 * w + h)
    nf, n = 3, ((w + 63) // 64) * ((h + 63) // 64)
    frames = rng.integers(0, 256, (nf, w * h * 3 // 2), dtype=np.uint8)
    lumas = [sc.random_params(rng, n, False) for _ in range(nf)]
    chromas = [sc.random_params(rng, n, True) for _ in range(nf)]
    din, dout = dev.put(frames), dev.empty(frames.nbytes)
    dl = dev.put(np.frombuffer(b"".join(bytes(x) for x in lumas), dtype=np.uint8))
    dc_ = dev.put(np.frombuffer(b"".join(bytes(x) for x in chromas), dtype=np.uint8))
    dev.lib.kvz_hip_dev_sao_frames.restype = None
    dev.lib.kvz_hip_dev_sao_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    dev.lib.kvz_hip_dev_sao_frames(din, dout, w, h, nf, dl, dc_)
    got = dev.get(dout, frames.shape, np.uint8)
    dev.free(din, dout, dl, dc_)
    for f in range(nf):
        want = sc.run_cpu(oracle.lib.kvz_oracle_sao_frame, w, h, frames[f], lumas[f], chromas[f])
        assert np.array_equal(got[f], want), (f, np.flatnonzero(got[f] != want)[:8])


def test_dev_paste_tiles_assembles_the_reference_frame(dev):
    """kvz_hip_dev_paste_tiles (the receive side of the tile-sharded inter configuration's exchange): every tile's planar picture, one slot each, pasted into the
    full frame by ONE launch == the numpy paste of kvazaar_amd/sharding.py, for BASELINE config 5's geometry (960x1088 / 960x1072 tiles) and a ragged small one"""
    import ctypes as C
    from kvazaar_amd import sharding
    lib = dev.lib
    lib.kvz_hip_dev_paste_tiles.restype = C.c_int
    lib.kvz_hip_dev_paste_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(3)
    for (w, h, cols, rows) in ((3840, 2160, 4, 2), (200, 136, 3, 2)):
        tiles = sharding.tile_grid(w, h, cols, rows)
        slot = sharding.tile_slot_bytes(tiles)
        order = rng.permutation(len(tiles))  # slot index of every tile: any order
        slots = np.zeros(len(tiles) * slot, np.uint8)
        want = np.zeros(w * h * 3 // 2, np.uint8)
        table = []
        for ti, t in enumerate(tiles):
            sub = rng.integers(0, 256, t[2] * t[3] * 3 // 2, dtype=np.uint8)
            slots[order[ti] * slot:order[ti] * slot + sub.size] = sub
            sharding.paste_tile(want, w, h, t, sub)
            table += [t[0], t[1], t[2], t[3], int(order[ti])]
        table = np.ascontiguousarray(np.array(table, np.int32))
        d_slots, d_frame = dev.put(slots), dev.empty(want.size)
        assert lib.kvz_hip_dev_paste_tiles(d_frame, w, h, d_slots, slot, table.ctypes.data, len(tiles), None) == 0
        lib.kvz_hip_dev_sync()
        got = dev.get(d_frame, want.shape, np.uint8)
        dev.free(d_slots, d_frame)
        assert np.array_equal(got, want), (w, h, cols, rows)
