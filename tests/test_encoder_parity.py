"""End-to-end parity with the reference ENCODER: tests/golden/encoder_recon.json holds digests of the reconstruction that
`kvazaar --preset ultrafast -p 1 -q QP --debug` (the CLI compiled from the reference tree, tests/golden/make_golden.py) writes for
seeded clips.  The oracle's CTU pass (+ picture-level deblocking) must produce exactly those pictures -- which pins the search
restatement (CU quadtree, modes, coefficients, adaptive CABAC contexts, WPP context hand-off) against the real encoder, not
only against per-function outputs; QPs on both sides of the switch from the fast coefficient cost to the CABAC model -- and so must the device sources (host simulation here, the MI355X under -m gpu)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import ctu_common as cc
import deblock_common as dc
import flatapi
from test_hostsim import hostsim  # noqa: F401  (fixture)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402

GOLDEN = json.load(open(os.path.join(HERE, "golden", "encoder_recon.json")))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def _entropy_table():
    return cc.model_constants()["entropy_fbits"]


def oracle_model(oracle, qp):
    import ctypes as C
    fb = (C.c_float * 128)(*_entropy_table())
    m = cc.CostModel()
    f = oracle.lib.kvz_oracle_intra_cost_model
    f.restype = None
    f.argtypes = [C.c_int, C.c_float * 128, C.c_uint64, C.POINTER(cc.CostModel)]
    f(qp, fb, cc.coeff_weights(qp), C.byref(m))
    return m


def _cu(o, w, h):
    return mg.cu_digest(o["depth"].reshape(h // 8, w // 8), o["mode"].reshape(h // 8, w // 8))


def _oracle_chain(oracle, model, w, h, frame, qp, deblock):
    o = cc.run_oracle(oracle, model, w, h, frame)
    if not deblock:
        return o["rec"]
    return dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, w, h, qp, 0, 0, o["rec"], o["depth"].reshape(h // 8, w // 8))


def _oracle_outputs(oracle, model, w, h, frames, qp):
    """one oracle pass per frame -> (digests before deblocking, CU-map digests, digests after deblocking)"""
    raw, cu, deb = [], [], []
    for f in frames:
        o = cc.run_oracle(oracle, model, w, h, f)
        raw.append(_sha(o["rec"]))
        cu.append(_cu(o, w, h))
        deb.append(_sha(dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, w, h, qp, 0, 0, o["rec"], o["depth"].reshape(h // 8, w // 8))))
    return raw, cu, deb


@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS, ids=lambda c: f"{c[0]}x{c[1]}-qp{c[5]}")
def test_oracle_chain_reproduces_reference_encoder(oracle, clip):
    """every clip of the fixture, up to the 3840x2160 pictures the north-star target is stated on"""
    w, h, n, seed, kind, qp = clip
    model = oracle_model(oracle, qp)
    frames = cc.yuv_frames(w, h, n, seed, kind)
    raw, cu, deb = _oracle_outputs(oracle, model, w, h, frames, qp)
    assert raw == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0)], clip
    assert deb == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1)], clip
    # ... and the decisions behind the pixels: CU depth and intra mode of every 8x8 cell as the encoder's cu_array holds them
    assert cu == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/cu"]


def _tile_pictures(w, h, frame, tiles):
    from kvazaar_amd import sharding
    grid = sharding.tile_grid(w, h, *(int(v) for v in tiles.split("x")))
    return grid, [sharding.crop_tile(frame, w, h, t) for t in grid]


@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_TILES, ids=lambda c: f"{c[0]}x{c[1]}-qp{c[5]}-tiles{c[6]}{'-wpp' if c[7] else ''}")
def test_oracle_tiles_reproduce_reference_encoder(oracle, clip):
    """--tiles CxR (BASELINE config 5 at its real geometry among them): every tile searched, reconstructed and deblocked as a picture of
    its own -- WPP off unless --wpp, as kvazaar does (cfg.c:925-978) -- and pasted back == the reference CLI's whole-frame reconstruction"""
    from kvazaar_amd import sharding
    w, h, n, seed, kind, qp, tiles, wpp = clip
    model = oracle_model(oracle, qp)
    model.no_wpp = 0 if wpp else 1
    raw, deb = [], []
    for f in cc.yuv_frames(w, h, n, seed, kind):
        grid, subs = _tile_pictures(w, h, f, tiles)
        full_raw, full_deb = np.zeros_like(f), np.zeros_like(f)
        for t, sub in zip(grid, subs):
            o = cc.run_oracle(oracle, model, t[2], t[3], sub)
            sharding.paste_tile(full_raw, w, h, t, o["rec"])
            sharding.paste_tile(full_deb, w, h, t, dc.run_cpu(oracle.lib.kvz_oracle_deblock_frame, t[2], t[3], qp, 0, 0, o["rec"],
                                                              o["depth"].reshape(t[3] // 8, t[2] // 8)))
        raw.append(_sha(full_raw))
        deb.append(_sha(full_deb))
    assert raw == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0, False, tiles, wpp)]
    assert deb == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1, False, tiles, wpp)]


@pytest.mark.gpu
@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_TILES, ids=lambda c: f"{c[0]}x{c[1]}-qp{c[5]}-tiles{c[6]}{'-wpp' if c[7] else ''}")
def test_hip_tiles_reproduce_reference_encoder(clip):
    """the product on the MI355X on tile-sharded pictures exactly as bench.py --tiles runs them: tiles of one geometry share a batch,
    no-WPP raster chains unless --wpp; deblocked and assembled == the reference CLI's reconstruction (no oracle in between)"""
    import kvazaar_amd
    from kvazaar_amd import sharding
    lib = kvazaar_amd.load_library()
    w, h, n, seed, kind, qp, tiles, wpp = clip
    model = cc.hip_cost_model(lib, qp, cc.coeff_weights(qp))
    model.no_wpp = 0 if wpp else 1
    frames = cc.yuv_frames(w, h, n, seed, kind)
    grid = sharding.tile_grid(w, h, *(int(v) for v in tiles.split("x")))
    raw, deb = [np.zeros_like(f) for f in frames], [np.zeros_like(f) for f in frames]
    by_geometry = {}
    for t in grid:
        by_geometry.setdefault((t[2], t[3]), []).append(t)
    for (tw, th), ts in sorted(by_geometry.items()):
        b = cc.HipBatch(lib, tw, th, n * len(ts))
        try:
            for i, f in enumerate(frames):
                for j, t in enumerate(ts):
                    b.upload(i * len(ts) + j, sharding.crop_tile(f, w, h, t))
            b.run(model)
            for i in range(n):
                for j, t in enumerate(ts):
                    sharding.paste_tile(raw[i], w, h, t, b.download(i * len(ts) + j)["rec"])
            b.deblock(qp)
            for i in range(n):
                for j, t in enumerate(ts):
                    sharding.paste_tile(deb[i], w, h, t, b.download(i * len(ts) + j)["rec"])
        finally:
            b.close()
    assert [_sha(r) for r in raw] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0, False, tiles, wpp)]
    assert [_sha(r) for r in deb] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1, False, tiles, wpp)]


@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_NO_WPP, ids=lambda c: f"{c[0]}x{c[1]}-qp{c[5]}")
def test_no_wpp_context_flow_reproduces_reference_encoder(oracle, hostsim, clip):
    """--no-wpp (what --tiles implies in kvazaar): one coder runs through the picture, a row starts from the contexts the row above
    ended with.  Oracle and the device sources (host simulation) against the reference CLI's reconstruction."""
    w, h, n, seed, kind, qp = clip
    model = oracle_model(oracle, qp)
    model.no_wpp = 1
    frames = cc.yuv_frames(w, h, n, seed, kind)
    want = GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0, True)]
    assert [_sha(cc.run_oracle(oracle, model, w, h, f)["rec"]) for f in frames] == want
    if w * h <= 416 * 240:
        assert [_sha(cc.run_hostsim(hostsim.lib, model, w, h, f)["rec"]) for f in frames] == want


@pytest.mark.gpu
@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_NO_WPP, ids=lambda c: f"{c[0]}x{c[1]}-qp{c[5]}")
def test_hip_batch_no_wpp_reproduces_reference_encoder(clip):
    import kvazaar_amd
    lib = kvazaar_amd.load_library()
    w, h, n, seed, kind, qp = clip
    model = cc.hip_cost_model(lib, qp, cc.coeff_weights(qp))
    model.no_wpp = 1
    frames = cc.yuv_frames(w, h, n, seed, kind)
    b = cc.HipBatch(lib, w, h, n)
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run(model)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0, True)]
    finally:
        b.close()


def test_frozen_contexts_do_not_reproduce_the_encoder(oracle):
    """the adaptive contexts matter: with every context frozen at its slice-start state the pass is still a valid encode,
    but not kvazaar's"""
    w, h, n, seed, kind, qp = 416, 240, 3, 1234, "small", 22
    assert (w, h, n, seed, kind, qp) in mg.ENCODER_CLIPS
    model = oracle_model(oracle, qp)
    model.adaptive = 0
    frames = cc.yuv_frames(w, h, n, seed, kind)
    got = [_sha(_oracle_chain(oracle, model, w, h, f, qp, 0)) for f in frames]
    assert got != GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0)]


def test_fuzz_of_the_oracle_against_the_reference_encoder():
    """tools/fuzz_intra_oracle.py: random pictures, every way a multiple of 8 can cut a CTU, QP 0..51, the searches of ultrafast / faster / fast / medium, deblocking and
    WPP on / off -- the oracle's picture must be what the compiled reference encoder writes with --debug"""
    if not os.path.exists(os.path.join(flatapi.ROOT, "oracle", "_ref", "kvazaar_ref")):
        pytest.skip("oracle/_ref not built (the GPU box): the committed digests are the check there")
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(flatapi.ROOT, "tools", "fuzz_intra_oracle.py"), "60", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 of 90 rounds differ" in r.stdout


def test_encoder_fixture_matches_reference_build(tmp_path):
    """where oracle/_ref exists, the committed fixture is what the reference CLI produces today (two small clips)"""
    if not os.path.exists(os.path.join(flatapi.ROOT, "oracle", "_ref", "kvazaar_ref")):
        pytest.skip("oracle/_ref not built")
    for (w, h, n, seed, kind, qp) in mg.ENCODER_CLIPS[:5]:
        frames = cc.yuv_frames(w, h, n, seed, kind)
        for deblock in (0, 1):
            maps = [] if deblock == 0 else None
            recs = mg.reference_encoder_recon(w, h, frames, qp, deblock, str(tmp_path), maps)
            assert [_sha(r) for r in recs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, deblock)]
            if maps is not None:
                assert [mg.cu_digest(d, m) for d, m in maps] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, deblock) + "/cu"]


def test_entropy_fixture_matches_reference_build():
    if not os.path.exists(flatapi.refshim_path()):
        pytest.skip("oracle/_ref not built")
    ref = flatapi.load_ref(0)
    assert [float(ref.lib.kvz_ref_entropy_fbits(i)) for i in range(128)] == _entropy_table()
    for qp in (10, 17, 22, 27, 37):
        assert ref.lib.kvz_ref_fast_coeff_weights(qp) == cc.coeff_weights(qp)


@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS[:5] + mg.ENCODER_CLIPS[7:9], ids=lambda c: f"{c[0]}x{c[1]}-qp{c[5]}")
def test_hostsim_pass_reproduces_reference_encoder(oracle, hostsim, clip):
    """the device sources compiled for the host (tests/hostsim): their CTU pass is kvazaar's, picture for picture"""
    hs = hostsim
    w, h, n, seed, kind, qp = clip
    model = oracle_model(oracle, qp)
    frames = cc.yuv_frames(w, h, n, seed, kind)
    outs = [cc.run_hostsim(hs.lib, model, w, h, f) for f in frames]
    assert [_sha(o["rec"]) for o in outs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0)]
    assert [_cu(o, w, h) for o in outs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/cu"]


@pytest.mark.gpu
@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS, ids=lambda c: f"{c[0]}x{c[1]}-qp{c[5]}")
def test_hip_batch_reproduces_reference_encoder(clip):
    """the product on the MI355X, no oracle in between: kvz_hip_intra_frames (+ kvz_hip_batch_deblock) with the model the
    library builds itself == the reference encoder's reconstruction"""
    import kvazaar_amd
    lib = kvazaar_amd.load_library()
    w, h, n, seed, kind, qp = clip
    model = cc.hip_cost_model(lib, qp, cc.coeff_weights(qp))
    frames = cc.yuv_frames(w, h, n, seed, kind)
    b = cc.HipBatch(lib, w, h, n)
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run(model)
        outs = [b.download(i) for i in range(n)]
        assert [_sha(o["rec"]) for o in outs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0)]
        assert [_cu(o, w, h) for o in outs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/cu"]
        b.deblock(qp)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1)]
    finally:
        b.close()


# ---- preset `fast`: 32x32 CUs searched (--pu-depth-intra 1-3), CABAC coefficient cost at every QP ------------------------------------------
def _fast_model(model):
    model.search_32x32 = 1
    model.coeff_cabac = 1
    return model


@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_FAST, ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_oracle_fast_preset_reproduces_reference_encoder(oracle, clip):
    w, h, n, seed, kind, qp = clip
    model = _fast_model(oracle_model(oracle, qp))
    frames = cc.yuv_frames(w, h, n, seed, kind)
    raw, _, deb = _oracle_outputs(oracle, model, w, h, frames, qp)
    assert raw == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/fast"]
    assert deb == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1) + "/fast"]


@pytest.mark.parametrize("clip", [c for c in mg.ENCODER_CLIPS_FAST if c[0] * c[1] <= 416 * 240], ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_hostsim_fast_preset_equals_oracle(oracle, hostsim, clip):
    """the device sources with the 32x32 search (CtuProgramT<.., true>) on the host: every output equals the oracle's, costs included"""
    w, h, n, seed, kind, qp = clip
    for cab in (1, 0):  # 0: --pu-depth-intra 1-3 on top of the fast coefficient estimate (not a preset, but a legal configuration)
        model = oracle_model(oracle, qp)
        model.search_32x32, model.coeff_cabac = 1, cab
        for f in cc.yuv_frames(w, h, n, seed, kind):
            assert not cc.compare(cc.run_oracle(oracle, model, w, h, f), cc.run_hostsim(hostsim.lib, model, w, h, f)), (clip, cab)


@pytest.mark.gpu
@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_FAST, ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_hip_fast_preset_reproduces_reference_encoder(oracle, clip):
    """the product on the MI355X with search_32x32: CTU pass, deblocking, SAO == `kvazaar --preset fast -p 1`, stage by stage; first frame also output by output against the oracle"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    lib = kvazaar_amd.load_library()
    w, h, n, seed, kind, qp = clip
    model = _fast_model(cost_model(lib, qp))
    frames = cc.yuv_frames(w, h, n, seed, kind)
    b = HipBatch(lib, w, h, n)
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run(model)
        outs = [b.download(i) for i in range(n)]
        assert [_sha(o["rec"]) for o in outs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/fast"]
        if w * h <= 832 * 480:
            assert not cc.compare(cc.run_oracle(oracle, _fast_model(oracle_model(oracle, qp)), w, h, frames[0]), outs[0])
        b.loop_filters(model, deblock=True, sao=True)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1) + "/fast/sao"]
    finally:
        b.close()


# ---- `--rdoq` (preset `medium` without its NxN partitions: --preset medium --pu-depth-intra 1-3): kvz_rdoq in every quantisation of the CTU pass ----------
def _rdoq_model(model):
    model.search_32x32 = 1
    model.coeff_cabac = 1
    model.rdoq = 1
    return model


@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_RDOQ, ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_oracle_rdoq_reproduces_reference_encoder(oracle, clip):
    w, h, n, seed, kind, qp = clip
    model = _rdoq_model(oracle_model(oracle, qp))
    frames = cc.yuv_frames(w, h, n, seed, kind)
    raw, _, deb = _oracle_outputs(oracle, model, w, h, frames, qp)
    assert raw == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/medium-pu13"]
    assert deb == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1) + "/medium-pu13"]


# ---- `--preset medium` itself: the above + the 4x4 NxN partitions of 8x8 CUs (pu-depth-intra 1-4; model.search_nxn) ----------
def _medium_model(model):
    model = _rdoq_model(model)
    model.search_nxn = 1
    return model


@pytest.mark.parametrize("clip", [c for c in mg.ENCODER_CLIPS_MEDIUM if c[0] * c[1] <= 1920 * 1080], ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_oracle_medium_reproduces_reference_encoder(oracle, clip):
    """`kvazaar --preset medium -p 1 --debug`, picture for picture: before the loop filters, the CU depth / first-PU mode maps behind it, after deblocking and
    after SAO (`--sao full`, the preset's own) -- with the NxN partition taken by every CU (noise at QP 12) down to a few per cent of the 8x8 CUs"""
    from test_sao_decision import oracle_sao_chain
    w, h, n, seed, kind, qp = clip
    model = _medium_model(oracle_model(oracle, qp))
    frames = cc.yuv_frames(w, h, n, seed, kind)
    raw, cu, deb = _oracle_outputs(oracle, model, w, h, frames, qp)
    assert raw == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/medium"]
    assert cu == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/medium/cu"]
    assert deb == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1) + "/medium"]
    if w * h <= 832 * 480:
        assert [_sha(oracle_sao_chain(oracle, model, w, h, f)[0]) for f in frames] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1) + "/medium/sao"]


def test_oracle_medium_takes_the_nxn_partition(oracle):
    """the fixture exercises what it is for: the noise picture at QP 12 codes every 8x8 CU as four 4x4 PUs with modes of their own"""
    import ctypes as C
    w, h, qp = 192, 136, 12
    f = cc.yuv_frames(w, h, 4, 0, "adversarial")[1]
    model = _medium_model(oracle_model(oracle, qp))
    o = cc.outputs(w, h)
    part, mode4 = np.zeros((h // 8) * (w // 8), np.uint8), np.zeros((h // 4) * (w // 4), np.uint8)
    ys, cs = w * h, w * h // 4
    fn = oracle.lib.kvz_oracle_intra_frame_nxn
    fn.restype = None
    fn(C.byref(model), w, h, cc.ptr(f[:ys]), cc.ptr(f[ys:ys + cs]), cc.ptr(f[ys + cs:]), cc.ptr(o["rec"]), cc.ptr(o["rec"], offset=ys), cc.ptr(o["rec"], offset=ys + cs),
       cc.ptr(o["coeff"]), cc.ptr(o["depth"]), cc.ptr(o["mode"]), o["cost"].ctypes.data_as(C.POINTER(C.c_double)), cc.ptr(part), cc.ptr(mode4))
    assert part.sum() > 0.9 * part.size and (o["depth"][part == 1] == 3).all()
    m4 = mode4.reshape(h // 4, w // 4)
    assert np.array_equal(m4[::2, ::2].reshape(-1), o["mode"])          # cu_mode holds the first PU's mode
    assert (m4[::2, ::2] != m4[1::2, 1::2]).mean() > 0.5                # ... and the PUs of a CU choose their own


@pytest.mark.parametrize("clip", [c for c in mg.ENCODER_CLIPS_MEDIUM if c[0] * c[1] <= 416 * 240], ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_hostsim_medium_equals_oracle(oracle, hostsim, clip):
    """the device sources with depth 4 of the search (the NxN partition of 8x8 CUs, CtuProgramT<true, true, true> with model.search_nxn) on the host: every output
    equals the oracle's -- pictures, levels, costs, the NxN flags and the 4x4-granular modes -- for `medium` itself and for the partition on top of the `ultrafast`
    search (no RDOQ, fast coefficient cost below QP 28: the instantiation's other switch positions)"""
    w, h, n, seed, kind, qp = clip
    frames = cc.yuv_frames(w, h, n, seed, kind)
    for variant in ("medium", "nxn-only"):
        model = oracle_model(oracle, qp)
        if variant == "medium":
            model = _medium_model(model)
        else:
            model.search_nxn = 1
        for f in (frames if variant == "medium" else frames[:2]):
            o = cc.run_oracle_nxn(oracle, model, w, h, f)
            assert not cc.compare(o, cc.run_hostsim_nxn(hostsim.lib, model, w, h, f)), (clip, variant)


@pytest.mark.gpu
@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_MEDIUM, ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_hip_medium_reproduces_reference_encoder(oracle, clip):
    """the product on the MI355X with the whole of `medium` in the CTU pass (32x32 search, RDOQ, NxN partitions): CTU pass, CU maps, deblocking, SAO ==
    `kvazaar --preset medium -p 1 --debug`, stage by stage; on the smaller clips every output also equals the oracle's (levels, costs, partitions, 4x4 modes)"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    lib = kvazaar_amd.load_library()
    w, h, n, seed, kind, qp = clip
    model = _medium_model(cost_model(lib, qp))
    frames = cc.yuv_frames(w, h, n, seed, kind)
    b = HipBatch(lib, w, h, n)
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run(model)
        outs = [b.download(i) for i in range(n)]
        assert [_sha(o["rec"]) for o in outs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/medium"]
        assert [_cu(o, w, h) for o in outs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/medium/cu"]
        if w * h <= 416 * 240:
            for i in range(n):
                outs[i]["part"], outs[i]["mode4"] = b.download_partitions(i)
                assert not cc.compare(cc.run_oracle_nxn(oracle, _medium_model(oracle_model(oracle, qp)), w, h, frames[i]), outs[i]), (clip, i)
        b.loop_filters(model, deblock=True, sao=False)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1) + "/medium"]
        b.run(model)
        b.loop_filters(model, deblock=True, sao=True)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1) + "/medium/sao"]
    finally:
        b.close()


@pytest.mark.parametrize("clip", [c for c in mg.ENCODER_CLIPS_RDOQ if c[0] * c[1] <= 416 * 240], ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_hostsim_rdoq_equals_oracle(oracle, hostsim, clip):
    """the device sources with kvz_rdoq in the quantisation stage (CtuProgramT<true, true, true>) on the host: every output equals the oracle's, costs included;
    also without the 32x32 search (--rdoq on top of `faster`)"""
    w, h, n, seed, kind, qp = clip
    for s32 in (1, 0):
        model = _rdoq_model(oracle_model(oracle, qp))
        model.search_32x32 = s32
        for f in cc.yuv_frames(w, h, n, seed, kind)[:1 if s32 == 0 else n]:
            assert not cc.compare(cc.run_oracle(oracle, model, w, h, f), cc.run_hostsim(hostsim.lib, model, w, h, f)), (clip, s32)


@pytest.mark.gpu
@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_RDOQ, ids=lambda c: f"{c[0]}x{c[1]}-{c[4]}-qp{c[5]}")
def test_hip_rdoq_reproduces_reference_encoder(oracle, clip):
    """the product on the MI355X with kvz_rdoq in the CTU pass: CTU pass, deblocking, SAO == `kvazaar --preset medium --pu-depth-intra 1-3 -p 1`, stage by stage"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    lib = kvazaar_amd.load_library()
    w, h, n, seed, kind, qp = clip
    model = _rdoq_model(cost_model(lib, qp))
    frames = cc.yuv_frames(w, h, n, seed, kind)
    b = HipBatch(lib, w, h, n)
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run(model)
        outs = [b.download(i) for i in range(n)]
        assert [_sha(o["rec"]) for o in outs] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0) + "/medium-pu13"]
        if w * h <= 416 * 240:
            assert not cc.compare(cc.run_oracle(oracle, _rdoq_model(oracle_model(oracle, qp)), w, h, frames[0]), outs[0])
        b.loop_filters(model, deblock=True, sao=True)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1) + "/medium-pu13/sao"]
    finally:
        b.close()


# ---- the pictures bench.py keeps resident, ALL of them (the first 8 frames of the 1080p clip, the first 4 of the 4K one): bench.py hashes one copy of each ----------
def test_oracle_reproduces_every_bench_picture_1080p(oracle):
    w, h, n, seed, kind, qp = mg.ENCODER_CLIPS_BENCH[0]
    model = oracle_model(oracle, qp)
    frames = cc.yuv_frames(w, h, n, seed, kind)
    assert [_sha(cc.run_oracle(oracle, model, w, h, f)["rec"]) for f in frames] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0)]


@pytest.mark.gpu
@pytest.mark.parametrize("clip", mg.ENCODER_CLIPS_BENCH, ids=lambda c: f"{c[0]}x{c[1]}-n{c[2]}")
def test_hip_reproduces_every_bench_picture(clip):
    """CTU pass and deblocking of all 8 / 4 distinct pictures of the bench batches == the reference encoder's reconstructions"""
    import kvazaar_amd
    from kvazaar_amd.batch import HipBatch, cost_model
    lib = kvazaar_amd.load_library()
    w, h, n, seed, kind, qp = clip
    model = cost_model(lib, qp, cc.coeff_weights(qp))
    frames = cc.yuv_frames(w, h, n, seed, kind)
    b = HipBatch(lib, w, h, n)
    try:
        for i, f in enumerate(frames):
            b.upload(i, f)
        b.run(model)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 0)]
        b.deblock(qp)
        assert [_sha(b.download(i)["rec"]) for i in range(n)] == GOLDEN[mg.clip_key(w, h, n, seed, kind, qp, 1)]
    finally:
        b.close()
