"""kvz_hip_dev_inter_ctu_pass -- the CTU pass of B pictures on the MI355X (kvazaar_amd/csrc/kvz_inter_ctu.hpp) -- against the sequence oracle
(oracle/kvz_oracle_inter.inc, equal to the reference encoder CU for CU: tests/test_inter_oracle.py): every B picture of a clip is searched on the device from the
oracle's reference picture and reference CU info, and the device's reconstruction and every CU decision must equal the oracle's picture for picture; then the
device chains its own pictures (its reconstruction deblocked on the device becomes the next picture's reference).  The host simulation of the same sources
(tests/hostsim, CPU) is checked the same way without a GPU."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

import ctu_common as cc
import flatapi
import inter_common as ic


from kvazaar_amd.inter import InterParams  # noqa: E402  (kvz_hip_inter_params; sets struct_size)


FAST_COST_CASES = ["pan", "ultrafast", "vertical-pan-owf", "static-qp17", "no-loop-filters", "survey-416x240"]  # every picture QP below 28: kvz_fast_coeff_cost
CABAC_COST_CASES = ["noisy-qp27", "cabac-coeff-cost-qp32", "fast-pan-owf-qp37", "ultrafast-fast-pan-owf-qp30"]     # picture QPs from 28 on: the residual coder in counting mode
EDGE_CASES = ["ultrafast-8mod16", "superfast-8mod16-qp33"]  # 8x8 inter CUs where the picture edge forces the split below pu-depth-inter's 16x16 (search.c:702-713)
FASTER_CASES = ["faster-pan", "faster-qp32", "faster-owf-qp27"]  # `--preset faster`: quarter-sample steps in the fractional search, CABAC coefficient cost at every QP


@pytest.fixture(scope="module")
def oracle():
    return flatapi.load_oracle()


def params_of(case, qp, poc):
    name, w, h, n, base_qp, preset, dbk, sao, owf, src = case
    p = ic.PRESETS[preset]
    return InterParams(qp=int(qp), poc=poc, mv_constraint=int(owf > 0), sao=int(sao), deblock=int(dbk), fme_level=p["fme_level"], pu_depth_inter_max=p["pu_depth_inter_max"], no_wpp=0, fast_residual_cost=p["fast_residual_cost"])


@pytest.fixture(scope="module")
def hostsim_lib():
    d = os.path.join(flatapi.ROOT, "tests", "hostsim")
    so = os.path.join(d, "libkvz_hostsim.so")
    srcs = [os.path.join(d, "hostsim.cpp")] + [os.path.join(flatapi.ROOT, "kvazaar_amd", "csrc", f) for f in os.listdir(os.path.join(flatapi.ROOT, "kvazaar_amd", "csrc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, os.path.join(d, "hostsim.cpp")])
    return C.CDLL(so)


@pytest.mark.parametrize("name", ["pan", "ultrafast", "vertical-pan-owf", "no-loop-filters", "noisy-qp27", "cabac-coeff-cost-qp32", "fast-pan-owf-qp37", "ultrafast-fast-pan-owf-qp30",
                                  "faster-pan", "faster-qp32", "faster-owf-qp27", "ultrafast-8mod16", "superfast-8mod16-qp33"])
def test_host_simulation_of_the_device_program_equals_the_oracle(oracle, hostsim_lib, name):
    case = [c for c in ic.CASES if c[0] == name][0]
    _, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames = ic.case_frames(case)
    rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
    mc = cc.model_constants()
    fb = np.array(mc["entropy_fbits"], np.float32)
    f = hostsim_lib.kvz_hostsim_inter_frame
    f.restype = None
    f.argtypes = [C.c_int] * 4 + [C.c_uint64, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 5
    p = ic.PRESETS[preset]
    for k in range(1, n):
        rec = np.zeros(w * h * 3 // 2, np.uint8)
        out = np.zeros((h // 4, w // 4), ic.CU_DTYPE)
        f(w, h, int(qps[k]), k, int(mc["coeff_weights"][str(int(qps[k]))]), fb.ctypes.data, int(owf > 0), int(sao), int(dbk), p["fme_level"], p["pu_depth_inter_max"], 0, p["fast_residual_cost"],
          np.ascontiguousarray(frames[k]).ctypes.data, np.ascontiguousarray(rf[k - 1]).ctypes.data, np.ascontiguousarray(cu[k - 1]).ctypes.data, rec.ctypes.data, out.ctypes.data)
        assert ic.first_difference(out[None], cu[k][None]) is None, k
        assert np.array_equal(rec, rs[k]), k


def test_fuzz_of_the_device_program_against_the_oracle(hostsim_lib):
    """tools/fuzz_inter.py: random clips, sizes that cut CTUs (incl. 8 mod 16), --qp 10..44, ultrafast / superfast / veryfast / faster, GOPs of 2 / 3 / 4 / 8, fast pans,
    loop filters / motion restriction / WPP on and off -- the simulated device program must equal the oracle on every B picture"""
    import sys
    r = subprocess.run([sys.executable, os.path.join(flatapi.ROOT, "tools", "fuzz_inter.py"), "60", "9"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 of 60 rounds differ" in r.stdout


def device_pass(lib, dev, w, h, srcs, refs, ref_cus, prm):
    n = len(srcs)
    lib.kvz_hip_dev_inter_ctu_pass.restype = C.c_int
    lib.kvz_hip_dev_inter_ctu_pass.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]
    fs, cells = w * h * 3 // 2, (w // 4) * (h // 4)
    d_src, d_ref, d_rcu = dev.put(np.concatenate(srcs)), dev.put(np.concatenate(refs)), dev.put(np.concatenate([c.reshape(-1) for c in ref_cus]))
    d_rec, d_cu = dev.empty(n * fs), dev.empty(n * cells * ic.CU_DTYPE.itemsize)
    rc = lib.kvz_hip_dev_inter_ctu_pass(d_src, d_ref, d_rcu, d_rec, d_cu, None, w, h, n, C.addressof(prm))
    assert rc == 0, rc
    rec = dev.get(d_rec, (n, fs), np.uint8)
    cu = dev.get(d_cu, (n, h // 4, w // 4), ic.CU_DTYPE)
    dev.free(d_src, d_ref, d_rcu, d_rec, d_cu)
    return rec, cu


@pytest.mark.gpu
@pytest.mark.parametrize("name", FAST_COST_CASES + CABAC_COST_CASES + FASTER_CASES + EDGE_CASES + ["two-gops"])
def test_device_pass_equals_oracle_picture_by_picture(oracle, name):
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    case = [c for c in ic.CASES if c[0] == name][0]
    _, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames = ic.case_frames(case)
    rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
    for k in range(1, n):
        rec, got = device_pass(lib, dev, w, h, [frames[k]], [rf[k - 1]], [cu[k - 1]], params_of(case, qps[k], k))
        d = ic.first_difference(got, cu[k][None])
        assert d is None, (k, {a: (b if a not in ("ours", "ref") else b.tolist()) for a, b in d.items()})
        assert np.array_equal(rec[0], rs[k]), k


@pytest.mark.gpu
def test_device_pass_on_baseline_config_4(oracle):
    """3840x2160 `--preset veryfast --gop lp-g4d3t1 -q 22`, SURVEY's own clip: the three B pictures, each from the oracle's reference"""
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    case = [c for c in ic.CASES if c[0] == "baseline-c4-2160p"][0]
    _, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames = ic.case_frames(case)
    rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
    assert ic.digests(rf, cu)["cu"] == json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inter_recon.json")))["baseline-c4-2160p"]["cu"]
    for k in range(1, n):
        rec, got = device_pass(lib, dev, w, h, [frames[k]], [rf[k - 1]], [cu[k - 1]], params_of(case, qps[k], k))
        assert ic.first_difference(got, cu[k][None]) is None, k
        assert np.array_equal(rec[0], rs[k]), k


CHAIN_CASES = ["deblock-only", "ultrafast", "pan", "vertical-pan-owf", "static-qp17", "no-loop-filters", "survey-416x240", "survey-1080p", "baseline-c4-2160p",
               "noisy-qp27", "cabac-coeff-cost-qp32", "ultrafast-fast-pan-owf-qp30", "faster-pan", "faster-qp32"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CHAIN_CASES)
def test_device_chains_its_own_pictures(oracle, name):
    """A whole sequence on the device from the I picture's search result on: its loop filters, then every B picture from the device's own previous picture -- CTU pass ->
    kvz_hip_dev_cu_dbk_from_info -> kvz_hip_dev_loop_filters_inter (deblocking with motion-based strengths, the SAO decision on the partly deblocked picture, SAO) -> reference
    of the next picture.  Every picture before and after its loop filters, every CU decision and every SAO decision must equal the oracle's (= the reference encoder's), up to
    BASELINE config 4's own 3840x2160 sequence."""
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    case = [c for c in ic.CASES if c[0] == name][0]
    _, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames = ic.case_frames(case)
    rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
    lib.kvz_hip_dev_inter_ctu_pass.restype = C.c_int
    lib.kvz_hip_dev_inter_ctu_pass.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]
    lib.kvz_hip_dev_cu_dbk_from_info.restype = None
    lib.kvz_hip_dev_cu_dbk_from_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.kvz_hip_dev_loop_filters_inter.restype = C.c_int
    lib.kvz_hip_dev_loop_filters_inter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 3
    fs, cells = w * h * 3 // 2, (w // 4) * (h // 4)
    d_rec, d_cu, d_dbk = dev.empty(fs), dev.empty(cells * ic.CU_DTYPE.itemsize), dev.empty(cells * 20)
    d_ref, d_rcu = dev.empty(fs), dev.empty(cells * ic.CU_DTYPE.itemsize)

    def loop_filters(k, d_src):
        lib.kvz_hip_dev_cu_dbk_from_info(d_cu, cells, d_dbk)
        assert lib.kvz_hip_dev_loop_filters_inter(d_src, d_rec, w, h, 1, d_dbk, int(qps[k]), int(k > 0), int(dbk), 0, 0, int(sao), 0, None, None, None) == 0
        assert np.array_equal(dev.get(d_rec, (fs,), np.uint8), rf[k]), k

    # the I picture: the oracle's search result (the all-intra pass has its own tests), filtered here
    d_src = dev.put(frames[0])
    dev.copy_in(d_rec, rs[0]); dev.copy_in(d_cu, cu[0].reshape(-1))
    loop_filters(0, d_src)
    dev.free(d_src)
    for k in range(1, n):
        d_ref, d_rec = d_rec, d_ref      # the filtered picture is the next reference
        d_rcu, d_cu = d_cu, d_rcu
        d_src = dev.put(frames[k])
        prm = params_of(case, qps[k], k)
        assert lib.kvz_hip_dev_inter_ctu_pass(d_src, d_ref, d_rcu, d_rec, d_cu, None, w, h, 1, C.addressof(prm)) == 0
        assert np.array_equal(dev.get(d_rec, (fs,), np.uint8), rs[k]), k
        assert ic.first_difference(dev.get(d_cu, (1, h // 4, w // 4), ic.CU_DTYPE), cu[k][None]) is None, k
        loop_filters(k, d_src)
        dev.free(d_src)
    dev.free(d_ref, d_rcu, d_rec, d_cu, d_dbk)


@pytest.mark.gpu
def test_device_pass_on_several_sequences_at_once(oracle):
    """picture k of three independent sequences in one launch (the ticket list interleaves their CTUs) == each of them alone"""
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    w, h, n, qp = 264, 200, 3, 22
    seqs = []
    for seed in (31, 32, 33):
        frames = ic.clip(w, h, n, seed, 1.5, (1.0 + seed % 3, -0.75))
        seqs.append((frames,) + ic.oracle_encode(oracle, w, h, frames, qp, preset="veryfast", deblock=True, sao=True, mv_constraint=True))
    case = ("x", w, h, n, qp, "veryfast", 1, 1, 2, None)
    for k in range(1, n):
        rec, got = device_pass(lib, dev, w, h, [s[0][k] for s in seqs], [s[2][k - 1] for s in seqs], [s[3][k - 1] for s in seqs], params_of(case, seqs[0][4][k], k))
        for i, s in enumerate(seqs):
            assert ic.first_difference(got[i][None], s[3][k][None]) is None, (k, i)
            assert np.array_equal(rec[i], s[1][k]), (k, i)


@pytest.mark.gpu
def test_device_pass_rejects_what_it_does_not_cover():
    import kvazaar_amd
    lib = kvazaar_amd.load_library()
    lib.kvz_hip_dev_inter_ctu_pass.restype = C.c_int
    lib.kvz_hip_dev_inter_ctu_pass.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]
    ok = InterParams(qp=25, poc=1, mv_constraint=0, sao=1, deblock=1, fme_level=2, pu_depth_inter_max=3, no_wpp=0, fast_residual_cost=28)
    for bad in (dict(qp=52), dict(qp=-1), dict(fme_level=5), dict(fast_residual_cost=52), dict(poc=0), dict(pu_depth_inter_max=4)):
        p = InterParams(**{**{n: getattr(ok, n) for n, _ in InterParams._fields_}, **bad})
        assert lib.kvz_hip_dev_inter_ctu_pass(None, None, None, None, None, None, 64, 64, 1, C.addressof(p)) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("clip", ["tiles2x1-pan", "tiles2x2-fast-pan-qp27"])
def test_device_tile_pass_in_the_tiled_chain(clip):
    """kvazaar --tiles CxR --preset veryfast --gop lp-g4d3t1: the DEVICE's inter CTU pass with the tile geometry of kvz_hip_inter_params (the pictures are tiles, the reference
    a whole frame: motion vectors and the co-located starting point reach into the other tiles; no TMVP) inside the chain of tests/tile_common.py -- every tile's
    reconstruction and CU records equal the device sources in host simulation, and the assembled pictures / CU decisions of the sequence equal the REFERENCE ENCODER's
    (tests/golden/inter_tiles.json).  Several copies of the tile per launch, so that workgroups of different sequences interleave."""
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    import golden.make_golden as mg
    import tile_common as tc
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    sim = tc.load_hostsim()
    host = tc.hostsim_tile_pass(sim)
    copies = 3

    def device_tile_pass(tw, th, pq, k, src, ref_frame, ref_cu, w, h, tx, ty):
        prm = InterParams(qp=int(pq), poc=k, mv_constraint=0, sao=1, deblock=1, fme_level=2, pu_depth_inter_max=3, no_wpp=1, fast_residual_cost=28,
                          ref_width=w, ref_height=h, tile_x=tx, tile_y=ty, no_tmvp=1)
        rec, cu = device_pass(lib, dev, tw, th, [src] * copies, [ref_frame] * copies, [ref_cu] * copies, prm)
        want_rec, want_cu = host(tw, th, pq, k, src, ref_frame, ref_cu, w, h, tx, ty)
        for i in range(copies):
            assert ic.first_difference(cu[i][None], want_cu[None]) is None, (k, tx, ty, i)
            assert np.array_equal(rec[i], want_rec), (k, tx, ty, i)
        return rec[0], cu[0]

    spec = [c for c in mg.INTER_TILE_CLIPS if c[0] == clip][0]
    pictures, records = tc.tiled_inter_chain(spec, 0, 1, None, device_tile_pass, sim)
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inter_tiles.json")))[clip]
    got = ic.digests(pictures, records)
    assert got["rec"] == want["rec"] and got["cu"] == want["cu"]


@pytest.mark.gpu
def test_device_pass_refuses_a_tile_outside_its_frame():
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    lib.kvz_hip_dev_inter_ctu_pass.restype = C.c_int
    lib.kvz_hip_dev_inter_ctu_pass.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]
    d = dev.empty(64 * 64 * 3)
    for bad in (dict(ref_width=128, ref_height=64, tile_x=128, tile_y=0), dict(ref_width=128, ref_height=64, tile_x=4, tile_y=0), dict(ref_width=100, ref_height=64, tile_x=0, tile_y=0)):
        prm = InterParams(qp=22, poc=1, mv_constraint=0, sao=1, deblock=1, fme_level=2, pu_depth_inter_max=3, no_wpp=1, fast_residual_cost=28, no_tmvp=1, **bad)
        assert lib.kvz_hip_dev_inter_ctu_pass(d, d, d, d, d, None, 64, 64, 1, C.addressof(prm)) == -1
    dev.free(d)
