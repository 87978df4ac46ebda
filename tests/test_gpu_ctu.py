"""Batched all-intra CTU pass on the MI355X (-m gpu): kvz_hip_intra_frames (include/kvz_hip_batch.h) against the
oracle restatement of kvazaar's search (oracle/kvz_oracle_ctu.c) -- reconstruction, coefficients, CU depths, intra
modes and double-precision RD costs all bit-exact -- plus size-independent properties at BASELINE's full sizes."""
import ctypes as C

import numpy as np
import pytest

import ctu_common as cc
import flatapi

flatapi.sys.path.insert(0, flatapi.os.path.join(flatapi.ROOT, "tests", "golden"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hiplib():
    import kvazaar_amd
    lib = kvazaar_amd.load_library()
    assert lib.kvz_hip_device_count() >= 1
    return lib


def _model(hiplib, oracle, qp):
    if not flatapi.os.path.exists(flatapi.refshim_path()):
        return cc.hip_cost_model(hiplib, qp)
    # the product's own cost-model builder must equal the one derived from the reference's tables
    import test_ctu_pipeline as t
    ref = flatapi.load_ref(0)
    m = cc.hip_cost_model(hiplib, qp, ref.lib.kvz_ref_fast_coeff_weights(qp))
    assert t.oracle_model(oracle, ref, qp).key() == m.key()
    return m


def _run_batch(hiplib, model, w, h, frames):
    b = cc.HipBatch(hiplib, w, h, len(frames))
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run(model)
        return [b.download(i) for i in range(len(frames))]
    finally:
        b.close()


@pytest.mark.parametrize("size", [(64, 64), (416, 240), (72, 88), (200, 136)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_hip_ctu_equals_oracle(oracle, hiplib, size):
    w, h = size
    model = _model(hiplib, oracle, 22)
    frames = cc.yuv_frames(w, h, 3, 1234, "small")
    got = _run_batch(hiplib, model, w, h, frames)
    for i, f in enumerate(frames):
        want = cc.run_oracle(oracle, model, w, h, f)
        assert not cc.compare(want, got[i]), (size, i, cc.compare(want, got[i]))


def test_hip_ctu_adversarial_and_qps(oracle, hiplib):
    w, h = 192, 136
    frames = list(cc.adversarial_frames(w, h).values())
    for qp in (10, 22, 28, 37, 51):  # fast coefficient cost below 28, the residual coder in counting mode from 28 on
        model = _model(hiplib, oracle, qp)
        got = _run_batch(hiplib, model, w, h, frames)
        for i, f in enumerate(frames):
            want = cc.run_oracle(oracle, model, w, h, f)
            assert not cc.compare(want, got[i]), (qp, i, cc.compare(want, got[i]))


@pytest.mark.parametrize("qp", [4, 12, 22])
def test_hip_ctu_residual_coder_large_levels(oracle, hiplib, qp):
    """The residual coder in counting mode where it is busiest: CABAC coefficient cost and the 32x32 search forced on (presets `faster` /
    `fast`) at QPs whose levels run into the escape codes with a moving Rice parameter (encode_coding_tree.c:224-246) -- adversarial and
    noise pictures, every output bit-exact against the oracle."""
    w, h = 192, 136
    frames = list(cc.adversarial_frames(w, h).values()) + cc.yuv_frames(w, h, 2, 77 + qp, "small")
    model = _model(hiplib, oracle, qp)
    model.coeff_cabac = 1
    model.search_32x32 = 1
    got = _run_batch(hiplib, model, w, h, frames)
    for i, f in enumerate(frames):
        want = cc.run_oracle(oracle, model, w, h, f)
        assert not cc.compare(want, got[i]), (qp, i, cc.compare(want, got[i]))


def test_hip_ctu_1080p_frame_equals_oracle(oracle, hiplib):
    """BASELINE config 2 geometry (1920x1080: 30x17 CTUs, partial bottom row), one frame against the oracle"""
    w, h = 1920, 1080
    model = _model(hiplib, oracle, 22)
    frames = cc.yuv_frames(w, h, 1, 1, "large")
    got = _run_batch(hiplib, model, w, h, frames)
    want = cc.run_oracle(oracle, model, w, h, frames[0])
    assert not cc.compare(want, got[0]), cc.compare(want, got[0])


def test_hip_ctu_4k_batch_independence(oracle, hiplib):
    """4K (BASELINE configs 3-5 geometry): frame 0 is the picture of tests/golden/encoder_recon.json's 3840x2160 clip -- the reference
    encoder's reconstruction and CU maps (test_encoder_parity.py checks the same digests and the deblocked picture) -- and frames of a
    batch are independent: a frame encodes identically alone and inside a batch, repeated runs are deterministic."""
    import hashlib
    import json
    import make_golden as mg
    w, h = 3840, 2160
    model = _model(hiplib, oracle, 22)
    frames = cc.yuv_frames(w, h, 2, 2, "large")
    batch = _run_batch(hiplib, model, w, h, frames)
    golden = json.load(open(flatapi.os.path.join(flatapi.ROOT, "tests", "golden", "encoder_recon.json")))
    assert hashlib.sha256(batch[0]["rec"].tobytes()).hexdigest()[:24] == golden[mg.clip_key(w, h, 1, 2, "large", 22, 0)][0]
    assert mg.cu_digest(batch[0]["depth"].reshape(h // 8, w // 8), batch[0]["mode"].reshape(h // 8, w // 8)) == golden[mg.clip_key(w, h, 1, 2, "large", 22, 0) + "/cu"][0]
    again = _run_batch(hiplib, model, w, h, frames)
    alone = _run_batch(hiplib, model, w, h, frames[1:])
    for i in range(2):
        assert not cc.compare(batch[i], again[i])
    assert not cc.compare(batch[1], alone[0])
    for i in range(2):
        y, ry = frames[i][:w * h].astype(np.float64), batch[i]["rec"][:w * h].astype(np.float64)
        psnr = 10 * np.log10(255 ** 2 / np.mean((y - ry) ** 2))
        assert 38.0 < psnr < 46.0, psnr
        assert np.isfinite(batch[i]["cost"]).all() and (batch[i]["cost"] > 0).all()
        assert set(np.unique(batch[i]["depth"])) <= {0, 1, 2, 3}
        assert batch[i]["mode"].max() <= 34


def test_hip_ctu_large_batch_stress(oracle, hiplib):
    """A full-machine batch (36 x 1080p = 18 360 CTUs, more than the ~768 resident workgroups, so the in-order ticket
    schedule hands CTUs between workgroups and XCDs all the time), run repeatedly: every frame must equal the oracle's
    result for its picture, every time.  Guards the inter-workgroup hand-off and the barrier discipline of the kernel."""
    w, h = 1920, 1080
    model = _model(hiplib, oracle, 22)
    distinct = cc.yuv_frames(w, h, 3, 5, "large")
    want = [cc.run_oracle(oracle, model, w, h, f) for f in distinct]
    n = 36
    b = cc.HipBatch(hiplib, w, h, n)
    try:
        for i in range(n):
            b.upload(i, distinct[i % 3])
        for rep in range(4):
            b.run(model)
            for i in range(n):
                got = b.download(i)
                assert not cc.compare(want[i % 3], got), (rep, i, cc.compare(want[i % 3], got))
    finally:
        b.close()


def test_batch_deblock_after_ctu_pass(oracle, hiplib):
    """kvz_hip_batch_deblock on the batch's own reconstruction + CU depths == the oracle's deblocking of the oracle's pass
    (the two stages kvazaar runs back to back per LCU, encoderstate.c:669-671)"""
    import deblock_common as dc
    w, h = 192, 136
    model22 = _model(hiplib, oracle, 22)
    frames = cc.yuv_frames(w, h, 3, 31, "small")
    batch = cc.HipBatch(hiplib, w, h, len(frames))
    for i, f in enumerate(frames):
        batch.upload(i, f)
    batch.run(model22)
    batch.deblock(22)
    for i, f in enumerate(frames):
        o = cc.run_oracle(oracle, model22, w, h, f)
        want = dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, w, h, 22, 0, 0, o["rec"], o["depth"].reshape(h // 8, w // 8))
        got = batch.download(i)["rec"]
        assert np.array_equal(got, want), i
        assert not np.array_equal(want, o["rec"])
        ys, cs = w * h, w * h // 4
        sums = [oracle.plane_checksum(flatapi.ptr(want), h, w, w), oracle.plane_checksum(flatapi.ptr(want, offset=ys), h // 2, w // 2, w // 2),
                oracle.plane_checksum(flatapi.ptr(want, offset=ys + cs), h // 2, w // 2, w // 2)]
        assert list(batch.checksums()[i]) == sums
        # --hash md5 (nal.c:88-101): per plane the RFC 1321 digest of the plane's bytes (python's hashlib as the independent known answer)
        import hashlib
        planes = (want[:ys], want[ys:ys + cs], want[ys + cs:])
        assert [bytes(batch.md5()[i][p]).hex() for p in range(3)] == [hashlib.md5(pl.tobytes()).hexdigest() for pl in planes]
    batch.close()


def test_c_host_example_matches_oracle_chain(oracle, hiplib, tmp_path):
    """examples/batch_intra.c -- a plain C host over the C ABI -- compiled with gcc and run as a process: its per-frame checksums
    must be those of the oracle's CTU pass + deblocking + picture checksum"""
    import re
    import subprocess
    import deblock_common as dc
    import kvazaar_amd
    root = flatapi.ROOT
    exe = str(tmp_path / "batch_intra")
    libdir = flatapi.os.path.dirname(kvazaar_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-I" + flatapi.os.path.join(root, "include"), flatapi.os.path.join(root, "examples", "batch_intra.c"),
                           "-L" + libdir, "-l:" + flatapi.os.path.basename(kvazaar_amd.LIB_PATH), "-Wl,-rpath," + libdir, "-o", exe])
    w, h = 128, 72
    frames = cc.yuv_frames(w, h, 2, 5, "small")
    yuv = tmp_path / "in.yuv"
    yuv.write_bytes(b"".join(f.tobytes() for f in frames))
    out = subprocess.run([exe, str(yuv), str(w), str(h), "22"], capture_output=True, text=True, check=True).stdout
    got = [tuple(int(v, 16) for v in m) for m in re.findall(r"checksum Y ([0-9a-f]+) U ([0-9a-f]+) V ([0-9a-f]+)", out)]
    model22 = _model(hiplib, oracle, 22)
    want = []
    for f in frames:
        o = cc.run_oracle(oracle, model22, w, h, f)
        r = dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, w, h, 22, 0, 0, o["rec"], o["depth"].reshape(h // 8, w // 8))
        ys, cs = w * h, w * h // 4
        want.append((oracle.plane_checksum(flatapi.ptr(r), h, w, w), oracle.plane_checksum(flatapi.ptr(r, offset=ys), h // 2, w // 2, w // 2),
                     oracle.plane_checksum(flatapi.ptr(r, offset=ys + cs), h // 2, w // 2, w // 2)))
    assert got == want, out


def test_two_geometries_side_by_side_on_their_shares_of_the_device(oracle, hiplib):
    """kvz_hip_batch_set_device_share: batches of two geometries (the two tile heights of a --tiles grid) launched back to back, each on half the workgroup slots, give
    what each gives alone -- every picture against the oracle -- and a share of (1, 1) afterwards restores the whole device"""
    model = _model(hiplib, oracle, 22)
    model.no_wpp = 1  # tiles: one coder per tile in raster order (cfg.c:925-978)
    geo = [(192, 136), (192, 120)]
    frames = [cc.yuv_frames(w, h, 24, 77 + k, "small") for k, (w, h) in enumerate(geo)]
    batches = [cc.HipBatch(hiplib, w, h, len(f)) for (w, h), f in zip(geo, frames)]
    try:
        for b, f in zip(batches, frames):
            for i, fr in enumerate(f):
                b.upload(i, fr)
            b.set_device_share(1, 2)
        for b in batches:  # asynchronous: the two persistent launches are resident together
            b.launch(model)
        for b in batches:
            b.sync()
        for (w, h), b, f in zip(geo, batches, frames):
            for i in (0, 11, 23):
                want = cc.run_oracle(oracle, model, w, h, f[i])
                assert not cc.compare(want, b.download(i)), ((w, h), i)
        batches[0].set_device_share(1, 1)
        batches[0].run(model)
        assert not cc.compare(cc.run_oracle(oracle, model, geo[0][0], geo[0][1], frames[0][5]), batches[0].download(5))
    finally:
        for b in batches:
            b.close()


def test_upload_all_async_feeds_the_next_pass(oracle, hiplib):
    """kvz_hip_batch_upload_all_async: a batch that has run a pass on one set of pictures gets another set from pinned host memory on its upload queue; the next pass
    (launched right behind, with no host synchronisation in between) waits for the copy and its results are those of the NEW pictures, frame for frame"""
    from kvazaar_amd.batch import HipBatch, pinned_bytes, pinned_free
    w, h, n = 200, 136, 6
    model = _model(hiplib, oracle, 27)
    first = cc.yuv_frames(w, h, n, 3, "small")
    second = cc.yuv_frames(w, h, n, 11, "small")
    assert any(not np.array_equal(a, b) for a, b in zip(first, second))
    b = HipBatch(hiplib, w, h, n)
    fb = w * h * 3 // 2
    ptr_, view = pinned_bytes(hiplib, n * fb)
    try:
        for i, f in enumerate(first):
            b.upload(i, f)
        b.launch(model)
        for i, f in enumerate(second):
            view[i * fb:(i + 1) * fb] = f
        b.upload_all_async(ptr_)  # queued behind the pass in flight
        b.launch(model)
        b.sync()
        for i, f in enumerate(second):
            assert not cc.compare(cc.run_oracle(oracle, model, w, h, f), b.download(i)), i
    finally:
        b.close()
        pinned_free(hiplib, ptr_)
