"""kvz_hip_batch_entropy_code on the MI355X: kvazaar's entropy coder in its real mode (kvazaar_amd/csrc/kvz_entropy.hpp) on the device-resident results of the CTU pass.
The bytes of every picture must be the slice data the REFERENCE ENCODER wrote (tests/golden/entropy.json, made from kvazaar_ref's bitstreams): ultrafast at several QPs,
--no-wpp, partial CTUs, a one-CTU-wide picture, noise / flat pictures (a bin list that outgrows the first capacity), SAO syntax from the device's own SAO decision,
`medium` with RDOQ levels and NxN CUs; then whole batches at BASELINE's picture sizes against the oracle's coder."""
import hashlib
import json
import os

import numpy as np
import pytest

import ctu_common as cc
import entropy_common as ec
import flatapi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "entropy.json")))


def device_model(lib, case):
    from kvazaar_amd.batch import cost_model
    name, w, h, n, seed, kind, qp, preset, extra = case
    m = cost_model(lib, qp, cc.coeff_weights(qp))
    if preset == "medium":
        m.coeff_cabac, m.search_32x32, m.rdoq, m.search_nxn = 1, 1, 1, 1
    if "--no-wpp" in extra:
        m.no_wpp = 1
    return m


def split(data, sizes):
    out, at = [], 0
    for row in sizes:
        total = int(row.sum())
        out.append((bytes(data[at:at + total]), [int(v) for v in row]))
        at += total
    assert at == len(data)
    return out


@pytest.mark.parametrize("case", ec.CASES + ec.BENCH_CASES, ids=[c[0] for c in ec.CASES + ec.BENCH_CASES])
def test_device_slice_data_equals_the_reference_encoders(case):
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch
    lib = kvazaar_amd.load_library()
    name, w, h, n, seed, kind, qp, preset, extra = case
    frames = cc.yuv_frames(w, h, n, seed, kind)
    model = device_model(lib, case)
    b = HipBatch(lib, w, h, n)
    for i, f in enumerate(frames):
        b.upload(i, f)
    b.run(model)
    sao = preset != "ultrafast"
    if sao:
        b.loop_filters(model, deblock=True, sao=True)
    data, sizes = b.entropy_code(model, sao=sao)
    got = split(data, sizes)
    b.close()
    for (bytes_, row), g in zip(got, GOLDEN[name]):
        assert row == g["sizes"]
        assert hashlib.sha256(bytes_).hexdigest()[:24] == g["sha"]


def test_model_matches_the_oracle_tests_model():
    """the medium switches of device_model are those of tests/test_encoder_parity.py _medium_model (what entropy.json was made with)"""
    import kvazaar_amd
    from test_encoder_parity import _medium_model, oracle_model
    oracle = flatapi.load_oracle()
    case = [c for c in ec.CASES if c[0] == "medium"][0]
    a, b = device_model(kvazaar_amd.load_library(), case), _medium_model(oracle_model(oracle, case[6]))
    for field in ("coeff_cabac", "search_32x32", "rdoq", "search_nxn", "no_wpp", "qp"):
        assert getattr(a, field) == getattr(b, field), field
    assert bytes(a.ctx_init) == bytes(b.ctx_init)


@pytest.mark.parametrize("w,h,n,seed,qp", [(1920, 1080, 8, 1, 22), (3840, 2160, 4, 2, 22)], ids=["1080p-x8", "2160p-x4"])
def test_device_slice_data_at_baseline_sizes(w, h, n, seed, qp):
    """BASELINE configs 2 and 5's pictures: every substream of every picture equals the oracle's coder run on the device's own CTU-pass results (which
    tests/test_gpu_ctu_batch.py pins to the reference encoder), several pictures per launch, scratch budget small enough to force chunks"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    from test_encoder_parity import oracle_model
    lib = kvazaar_amd.load_library()
    oracle = flatapi.load_oracle()
    frames = cc.yuv_frames(w, h, n, seed, "large")
    model = cost_model(lib, qp, cc.coeff_weights(qp))
    b = HipBatch(lib, w, h, n)
    for i, f in enumerate(frames):
        b.upload(i, f)
    b.run(model)
    os.environ["KVZ_HIP_ENTROPY_SCRATCH_MB"] = "160" if w == 1920 else "300"  # three / two pictures per chunk
    try:
        data, sizes = b.entropy_code(model)
    finally:
        del os.environ["KVZ_HIP_ENTROPY_SCRATCH_MB"]
    got = split(data, sizes)
    om = oracle_model(oracle, qp)
    for i in range(n):
        o = b.download(i)
        want, want_sizes = ec.oracle_entropy(oracle, om, w, h, o)
        assert got[i][1] == want_sizes, i
        assert got[i][0] == want, i
    b.close()


def test_device_slice_data_of_tiles():
    """pictures that are tiles of a slice (kvz_hip_batch_entropy_code_tiles): all but the slice's last end in end_of_subset_one_bit -- against the oracle's tile coder;
    the integrated encoder's --tiles bitstreams are checked in tests/test_e2e_dropin.py"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    from test_encoder_parity import oracle_model
    lib = kvazaar_amd.load_library()
    oracle = flatapi.load_oracle()
    w, h, n, qp = 208, 120, 4, 22
    frames = cc.yuv_frames(w, h, n, 7, "small")
    model = cost_model(lib, qp, cc.coeff_weights(qp))
    model.no_wpp = 1
    b = HipBatch(lib, w, h, n)
    for i, f in enumerate(frames):
        b.upload(i, f)
    b.run(model)
    flags = [1, 1, 0, 1]
    data, sizes = b.entropy_code(model, not_last=flags)
    got = split(data, sizes)
    om = oracle_model(oracle, qp)
    om.no_wpp = 1
    for i in range(n):
        want, want_sizes = ec.oracle_entropy(oracle, om, w, h, b.download(i), not_last=flags[i])
        assert got[i] == (want, want_sizes), i
    assert got[2] != split(*b.entropy_code(model, not_last=[1, 1, 1, 1]))[2]
    b.close()


def test_coder_starts_the_next_batchs_pass():
    """kvz_hip_batch_entropy_code_then: batch A's slice data is what the plain call writes, and batch B's pass -- started by A's coder once its third stage is
    queued -- leaves what a pass launched the usual way leaves (two geometries, so that nothing of A can be mistaken for B)"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    lib = kvazaar_amd.load_library()
    model = cost_model(lib, 22, cc.coeff_weights(22))
    fa, fb = cc.yuv_frames(416, 240, 3, 5, "small"), cc.yuv_frames(256, 192, 2, 6, "small")
    a, b = HipBatch(lib, 416, 240, 3), HipBatch(lib, 256, 192, 2)
    for i, f in enumerate(fa):
        a.upload(i, f)
    for i, f in enumerate(fb):
        b.upload(i, f)
    a.run(model)
    b.run(model)
    want_b = [b.download(i) for i in range(2)]
    want_data, want_sizes = a.entropy_code(model)
    want_data = bytes(want_data)
    b.run(cost_model(lib, 37, cc.coeff_weights(37)))  # (other results in B's buffers: the pass below has to overwrite them)
    a.run(model)
    data, sizes = a.entropy_code(model, then=(b, model))
    b.sync()
    got_b = [b.download(i) for i in range(2)]
    assert bytes(data) == want_data and np.array_equal(sizes, want_sizes)
    for g, w_ in zip(got_b, want_b):
        assert cc.compare(g, w_) == []
    a.close()
    b.close()


def test_deferred_download_in_a_two_batch_pipeline():
    """kvz_hip_batch_entropy_defer_download: the calls return with the slice data's download queued; two batches coded in turn (the second call compacts while the first
    batch's bytes are still on their way down: each batch has a compaction buffer of its own in this mode) deliver, after their sync, the bytes of the plain calls"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    lib = kvazaar_amd.load_library()
    model = cost_model(lib, 22, cc.coeff_weights(22))
    fa, fb = cc.yuv_frames(416, 240, 4, 5, "small"), cc.yuv_frames(416, 240, 4, 9, "small")
    a, b = HipBatch(lib, 416, 240, 4), HipBatch(lib, 416, 240, 4)
    for i in range(4):
        a.upload(i, fa[i])
        b.upload(i, fb[i])
    a.run(model)
    b.run(model)
    want = [(bytes(d), s.copy()) for d, s in (a.entropy_code(model), b.entropy_code(model))]
    assert want[0][0] != want[1][0]
    for x in (a, b):
        x.entropy_defer_download(True)
    for turn in range(3):
        a.launch(model)
        da, sa = a.entropy_code(model, then=(b, model))
        db, sb = b.entropy_code(model)
        a.sync()
        b.sync()
        assert bytes(da) == want[0][0] and np.array_equal(sa, want[0][1]), turn
        assert bytes(db) == want[1][0] and np.array_equal(sb, want[1][1]), turn
    a.close()
    b.close()
