"""The inter-search oracle (oracle/kvz_oracle_inter.inc: kvazaar's CTU search of B pictures in the low-delay GOP of BASELINE config 4, with the loop filters and
the picture-to-picture reference chain) against the reference encoder:
  * the committed digests of oracle/_ref/kvazaar_ref's own output on every clip of tests/inter_common.py CASES (tests/golden/inter_recon.json, written by
    tests/golden/make_golden.py --inter): reconstruction (--debug) and CU decisions (type, depth, skip / merge, merge index, motion vectors, MVP indices, intra
    mode) of every picture -- up to BASELINE config 4's own 3840x2160;
  * the compiled reference itself where oracle/_ref is built (this container), CU for CU.
No GPU involved: this pins the checker the device inter pass will be compared with."""
import ctypes as C
import json
import os
import tempfile

import numpy as np
import pytest

import flatapi
import inter_common as ic

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inter_recon.json")))


@pytest.fixture(scope="module")
def oracle():
    return flatapi.load_oracle()


def qp_of(oracle, qp, frame, gop=(4, 3), period=64, ra8=1):
    f = oracle.lib.kvz_oracle_lowdelay_qp
    f.restype = C.c_int
    f.argtypes = [C.c_int] * 6
    return f(qp, gop[0], gop[1], frame, period, ra8)


def test_picture_qp_of_the_lowdelay_gop(oracle):
    """what `kvazaar --preset veryfast --gop lp-g4d3t1 -q <qp>` runs its pictures at (state->qp printed from a debug build of the reference): the layer offsets
    3 2 3 1, the I picture at -1, and the QP model of kvz_gop_ra8's entries left behind by the preset's "gop 8" (+1 on the third picture from --qp 24 on)"""
    assert [qp_of(oracle, 22, f) for f in range(9)] == [21, 25, 24, 25, 23, 25, 24, 25, 23]
    assert [qp_of(oracle, 24, f) for f in range(9)] == [23, 27, 26, 28, 25, 27, 26, 28, 25]
    assert [qp_of(oracle, 24, f, ra8=0) for f in range(5)] == [23, 27, 26, 27, 25]   # no preset before --gop: the model fields are zero
    assert [qp_of(oracle, 37, f) for f in range(5)] == [36, 40, 42, 43, 40]            # 39 * 0.25 - 6.25 = 3.5 -> clipped to 3; 40 * .25 - 6.25 = 3.75 -> 3; 38 * .245 - 7 = 2.31
    assert qp_of(oracle, 22, 64) == 21 and qp_of(oracle, 22, 65) == 25                  # --period 64: the GOP restarts at every I picture
    assert qp_of(oracle, 51, 1) == 51 and qp_of(oracle, 0, 0) == 0                      # CLIP_TO_QP


@pytest.mark.parametrize("case", ic.CASES, ids=[c[0] for c in ic.CASES])
def test_oracle_reproduces_the_reference_encoder(oracle, case):
    name, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames = ic.case_frames(case)
    import hashlib
    assert hashlib.md5(b"".join(f.tobytes() for f in frames)).hexdigest() == GOLDEN[name]["clip_md5"], "the synthetic clip itself changed"
    rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
    got = ic.digests(rf, cu)
    assert got["cu"] == GOLDEN[name]["cu"], [i for i in range(n) if got["cu"][i] != GOLDEN[name]["cu"][i]]
    assert got["rec"] == GOLDEN[name]["rec"], [i for i in range(n) if got["rec"][i] != GOLDEN[name]["rec"][i]]
    if not dbk and not sao:
        assert np.array_equal(rs, rf)  # without loop filters the search's reconstruction is the output
    assert (cu["type"][0] == 1).all() and (cu["type"][1:] != 0).all()


def test_cases_cover_every_decision_kind():
    """the fixtures are only worth something if the clips make the encoder take every path: counted on the oracle's output of three of them"""
    oracle = flatapi.load_oracle()
    seen = dict(intra=0, skipped=0, merged=0, amvp=0, bipred=0, l1_only=0, depth1=0, depth3=0, mvp1=0, fractional=0)
    for case in [c for c in ic.CASES if c[0] in ("noisy-qp27", "fast-pan-owf-qp37", "static-qp17")]:
        name, w, h, n, qp, preset, dbk, sao, owf, src = case
        _, _, cu, _ = ic.oracle_encode(oracle, w, h, ic.case_frames(case), qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
        b = cu[1:]
        inter = b["type"] == 2
        amvp = inter & (b["merged"] == 0) & (b["skipped"] == 0)
        seen["intra"] += int((b["type"] == 1).sum()); seen["skipped"] += int((inter & (b["skipped"] == 1)).sum()); seen["merged"] += int((inter & (b["merged"] == 1)).sum())
        seen["amvp"] += int(amvp.sum()); seen["bipred"] += int((inter & (b["mv_dir"] == 3)).sum()); seen["l1_only"] += int((inter & (b["mv_dir"] == 2)).sum())
        seen["depth1"] += int((inter & (b["depth"] == 1)).sum()); seen["depth3"] += int((inter & (b["depth"] == 3)).sum())
        seen["mvp1"] += int((amvp & (b["mv_cand"][..., 0] == 1)).sum()); seen["fractional"] += int((amvp & ((b["mv"][..., 0, 0] & 3) != 0)).sum())
    assert all(v > 0 for k, v in seen.items() if k != "l1_only"), seen   # L1-only motion cannot win: the L1 AMVP candidate is dropped and merge candidates are L0 or both


@pytest.mark.parametrize("name", ["pan", "fast-pan-owf-qp37", "ultrafast-fast-pan-owf-qp30", "two-gops"])
def test_oracle_vs_compiled_reference_cu_by_cu(oracle, name):
    if not os.path.exists(os.path.join(flatapi.ROOT, "oracle", "_ref", "kvazaar_ref")):
        pytest.skip("oracle/_ref not built (the GPU box): the committed digests above are the check there")
    case = [c for c in ic.CASES if c[0] == name][0]
    _, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames = ic.case_frames(case)
    with tempfile.TemporaryDirectory() as d:
        rrec, rcu = ic.reference_encode(w, h, frames, qp, d, preset=preset, deblock=bool(dbk), sao=bool(sao), owf=owf)
    rs, rf, cu, qps = ic.oracle_encode(oracle, w, h, frames, qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
    assert ic.first_difference(cu, rcu) is None
    assert np.array_equal(rf, rrec)
    assert ic.digests(rrec, rcu) == {k: GOLDEN[name][k] for k in ("rec", "cu")}


def test_fuzz_of_the_oracle_against_the_reference_encoder():
    """tools/fuzz_inter_oracle.py: random clips and switch settings through the oracle and through the compiled reference encoder (how the forced-split rule of
    search.c:702-713 was found: 8x8 inter CUs at the edge of pictures whose size is 8 mod 16 under `ultrafast` / `superfast`)"""
    if not os.path.exists(os.path.join(flatapi.ROOT, "oracle", "_ref", "kvazaar_ref")):
        pytest.skip("oracle/_ref not built (the GPU box): the committed digests are the check there")
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(flatapi.ROOT, "tools", "fuzz_inter_oracle.py"), "40", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 of 40 rounds differ" in r.stdout


def test_default_threading_gives_the_constrained_result():
    """kvazaar's output depends on --owf only through the motion-vector restriction of overlapped pictures (search_inter.c:85): the default CLI (threads and owf
    auto) and --threads 0 --owf 2 write the same reconstruction, which is why the fixtures can be recorded single-threaded"""
    if not os.path.exists(os.path.join(flatapi.ROOT, "oracle", "_ref", "kvazaar_ref")):
        pytest.skip("oracle/_ref not built")
    case = [c for c in ic.CASES if c[0] == "survey-416x240"][0]
    _, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames = ic.case_frames(case)
    import hashlib
    import subprocess
    with tempfile.TemporaryDirectory() as d:
        rrec, _ = ic.reference_encode(w, h, frames, qp, d, preset=preset, owf=2, cu=False)
        exe = os.path.join(flatapi.ROOT, "oracle", "_ref", "kvazaar_ref")
        subprocess.run([exe, "-i", os.path.join(d, "in.yuv"), "--input-res", f"{w}x{h}", "--preset", preset, "--gop", "lp-g4d3t1", "-q", str(qp), "--debug", os.path.join(d, "rd.yuv"),
                        "-o", os.path.join(d, "o.hevc")], check=True, capture_output=True)
        assert np.array_equal(np.fromfile(os.path.join(d, "rd.yuv"), np.uint8).reshape(n, -1), rrec)
        md5 = hashlib.md5(open(os.path.join(d, "o.hevc"), "rb").read()).hexdigest()
    assert md5 == GOLDEN["survey-416x240"]["bitstream_md5"] == "1e7a81653d1157ce05e9ffd85c7cd5cf"  # SURVEY.md App. C's inter stream


import hashlib as _hashlib
ENTROPY_GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "entropy_inter.json")))


@pytest.mark.parametrize("name", ic.ENTROPY_CASES)
def test_oracle_slice_data_of_low_delay_sequences_equals_the_reference_encoders(oracle, name):
    """oracle/kvz_oracle_entropy.inc e_b_picture -- kvz_encode_coding_tree with the inter syntax (skip / merge / inter_pred_idc / MVD / MVP index / rqt_root_cbf), in real mode --
    and the I picture's coder inside a sequence (SAO decisions, picture QPs of the GOP): every picture's slice data and entry points are the reference encoder's
    (tests/golden/entropy_inter.json, taken from kvazaar_ref's bitstreams by make_golden.py --entropy-inter)"""
    case = [c for c in ic.CASES if c[0] == name][0]
    _, w, h, n, qp, preset, dbk, sao, owf, src = case
    got = ic.oracle_encode_bits(oracle, w, h, ic.case_frames(case), qp, preset=preset, deblock=bool(dbk), sao=bool(sao), mv_constraint=owf > 0)
    assert len(got) == len(ENTROPY_GOLDEN[name])
    for (data, sizes), g in zip(got, ENTROPY_GOLDEN[name]):
        assert sizes == g["sizes"]
        assert _hashlib.sha256(data).hexdigest()[:24] == g["sha"]
