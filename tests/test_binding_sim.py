"""The kvazaar-side binding of the batched pass (integration/kvazaar/search_lcu_hip.c) checked WITHOUT a GPU: oracle/_ref/kvazaar_hipsim is the integrated
encoder with the kvz_hip_batch_* calls served by the oracle's CTU pass (oracle/batch_sim.c, test infrastructure) instead of the device.  What is under test is the
binding's own logic -- eligibility, picture gathering, CU-array / coded-block-flag rebuild incl. NxN CUs, tile views, coefficient hand-over: the bitstream must be the
reference encoder's, byte for byte.  The same cases run against the device under -m gpu (tests/test_e2e_dropin.py, oracle/_ref/kvazaar_hip)."""
import hashlib
import os
import subprocess

import pytest

import ctu_common as cc
import flatapi
from kvazaar_amd import synth

REF = os.path.join(flatapi.ROOT, "oracle", "_ref")


def _encode(binary, yuv, res, out, extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([os.path.join(REF, binary), "-i", yuv, "--input-res", res, "-o", out] + extra, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return hashlib.md5(open(out, "rb").read()).hexdigest()


CASES = [("416x240", 2, 1234, "small", ["--preset", "medium", "-p", "1"]),                           # BASELINE config 3's preset as it is: rdoq, NxN, SAO
         ("416x240", 2, 1234, "small", ["--preset", "medium", "-p", "1", "-q", "12"]),               # ... where a sixth of the 8x8 CUs go NxN
         ("192x136", 4, 0, "adversarial", ["--preset", "medium", "-p", "1", "-q", "12"]),            # the noise picture: every CU NxN
         ("416x240", 2, 1234, "small", ["--preset", "ultrafast", "-p", "1", "--pu-depth-intra", "2-4"]),
         ("416x240", 2, 1234, "small", ["--preset", "medium", "-p", "1", "-q", "32", "--tiles", "2x2"]),
         ("416x240", 3, 1234, "small", ["--preset", "ultrafast", "-p", "1", "-q", "32", "--no-wpp"])]


@pytest.mark.parametrize("res,frames,seed,kind,opts", CASES, ids=[" ".join(c[4][1:2] + c[4][4:]).replace(" ", "_") + "_" + c[0] for c in CASES])
def test_binding_writes_the_reference_bitstream(tmp_path, res, frames, seed, kind, opts):
    if not os.path.exists(os.path.join(REF, "kvazaar_hipsim")):
        pytest.skip("oracle/_ref/kvazaar_hipsim not built (oracle/Makefile, where /root/reference exists)")
    w, h = (int(v) for v in res.split("x"))
    yuv = str(tmp_path / "syn.yuv")
    if kind == "adversarial":
        open(yuv, "wb").write(b"".join(f.tobytes() for f in cc.yuv_frames(w, h, frames, seed, kind)))
    else:
        synth.write_yuv(yuv, w, h, frames, seed, kind)
    want = _encode("kvazaar_ref", yuv, res, str(tmp_path / "ref.hevc"), opts + ["--threads", "4"])
    got = _encode("kvazaar_hipsim", yuv, res, str(tmp_path / "sim.hevc"), opts + ["--threads", "4"],
                  {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_TRACE": str(tmp_path / "trace")})
    assert got == want
    assert int(open(str(tmp_path / "trace")).read().split()[0]) >= frames, "the batched search was not used"


def test_binding_reproduces_the_survey_md5_at_1080p(tmp_path):
    """1920x1080 x 8 frames --preset ultrafast -p 1 (BASELINE config 2's geometry through the binding's slot table and cbf rebuild)"""
    if not os.path.exists(os.path.join(REF, "kvazaar_hipsim")):
        pytest.skip("oracle/_ref/kvazaar_hipsim not built")
    yuv = str(tmp_path / "syn.yuv")
    assert synth.write_yuv(yuv, 1920, 1080, 8, 1, "large") == synth.MD5["1920x1080"]
    got = _encode("kvazaar_hipsim", yuv, "1920x1080", str(tmp_path / "sim.hevc"), ["--preset", "ultrafast", "-p", "1", "--threads", "8"],
                  {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1"})
    assert got == "dce84d2200dc0e54e2e029425d1682e1"


# ---- B pictures: the binding's inter path (search_lcu_inter: reference picture and CU-array hand-over, cu_info_t rebuild incl. motion, coefficients) ----
INTER_CASES = [("416x240", 8, ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0"], 7),                          # BASELINE config 4's options
               ("416x240", 6, ["--preset", "ultrafast", "--gop", "lp-g4d3t1", "--owf", "0", "-q", "20"], 5),            # fme 0, PUs down to 16x16, no SAO
               ("416x240", 5, ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0", "--no-wpp"], 4),             # one coder through the picture
               ("416x240", 10, ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0", "--period", "8"], 8),       # a second I picture: the chain restarts
               ("416x240", 6, ["--preset", "superfast", "--gop", "lp-g8d4t1", "--owf", "0", "--no-sao"], 5),            # another low-delay GOP: other picture QPs
               ("416x240", 6, ["--preset", "ultrafast", "--gop", "lp-g4d3t1", "--owf", "0", "-q", "24"], 5),            # picture QPs on both sides of fast-residual-cost 28
               ("416x240", 6, ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0", "-q", "32"], 5),            # coefficients priced by the residual coder's counting mode
               ("416x240", 6, ["--preset", "faster", "--gop", "lp-g4d3t1", "--owf", "0"], 5),                        # subme 4, fast-residual-cost 0
               ("200x136", 5, ["--preset", "ultrafast", "--gop", "lp-g4d3t1", "--owf", "0", "-q", "25"], 4)]        # 8 mod 16: 8x8 inter CUs where the edge forces the split (search.c:702-713)


@pytest.mark.parametrize("res,frames,opts,device_pictures", INTER_CASES, ids=["veryfast", "ultrafast-qp20", "veryfast-no-wpp", "veryfast-period8", "superfast-g8-no-sao", "ultrafast-qp24-mixed", "veryfast-qp32", "faster", "ultrafast-8mod16"])
def test_binding_inter_pictures_write_the_reference_bitstream(tmp_path, res, frames, opts, device_pictures):
    """B pictures of a low-delay GOP through the binding: the reference picture (after kvazaar's own loop filters) and its cu_array go in, cu_info_t / reconstruction /
    coefficients of every LCU come back; kvazaar's entropy coder must then write the reference encoder's bitstream"""
    if not os.path.exists(os.path.join(REF, "kvazaar_hipsim")):
        pytest.skip("oracle/_ref/kvazaar_hipsim not built (oracle/Makefile, where /root/reference exists)")
    w, h = (int(v) for v in res.split("x"))
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, w, h, frames, 1234, "small")
    want = _encode("kvazaar_ref", yuv, res, str(tmp_path / "ref.hevc"), opts + ["--threads", "4"])
    got = _encode("kvazaar_hipsim", yuv, res, str(tmp_path / "sim.hevc"), opts + ["--threads", "4"],
                  {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_INTER_TRACE": str(tmp_path / "trace")})
    assert got == want
    assert int(open(str(tmp_path / "trace")).read()) >= device_pictures, "the inter pass was not used"
    if opts[1] == "veryfast" and len(opts) == 6 and frames == 8:
        assert got == "1e7a81653d1157ce05e9ffd85c7cd5cf"  # 416x240 x 8 --preset veryfast --gop lp-g4d3t1


def test_binding_leaves_b_pictures_with_a_custom_coeff_table_to_kvazaar(tmp_path):
    """--fast-coeff-table: the inter pass prices with the built-in weights, so B pictures under a custom table must stay with kvz_search_lcu (the I pictures take the table's
    weights through the cost model).  encoder.c:168 clears cfg.fast_coeff_table_fn after parsing, so the binding has to look at the parsed table: the bitstream is the
    reference encoder's, and no B picture went through the pass."""
    if not os.path.exists(os.path.join(REF, "kvazaar_hipsim")):
        pytest.skip("oracle/_ref/kvazaar_hipsim not built (oracle/Makefile, where /root/reference exists)")
    table = str(tmp_path / "weights.txt")
    with open(table, "w") as f:
        for qp in range(50):  # fast_coeff_cost.c:56-72: fifty lines of four weights
            f.write(f"{0.16 + 0.001 * qp:.6f} {4.2 + 0.01 * qp:.6f} {3.1 + 0.02 * qp:.6f} {6.5 + 0.03 * qp:.6f}\n")
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, 416, 240, 6, 1234, "small")
    opts = ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0", "--fast-coeff-table", table, "--threads", "4"]
    want = _encode("kvazaar_ref", yuv, "416x240", str(tmp_path / "ref.hevc"), opts)
    got = _encode("kvazaar_hipsim", yuv, "416x240", str(tmp_path / "sim.hevc"), opts, {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_INTER_TRACE": str(tmp_path / "trace")})
    assert got == want
    assert not os.path.exists(str(tmp_path / "trace")) or int(open(str(tmp_path / "trace")).read() or 0) == 0, "B pictures went through the inter pass under a custom table"
    plain = _encode("kvazaar_ref", yuv, "416x240", str(tmp_path / "plain.hevc"), [o for o in opts if o not in ("--fast-coeff-table", table)])
    assert plain != want, "the table does not change the encode: the case tests nothing"


# ---- the device's entropy coder behind the binding (KVZ_HIP_BATCH_ENTROPY=1): levels stay on the device, kvz_encode_coding_tree is skipped, the row coders' streams are
# replaced by the device's substreams before the slice header takes its entry points from them ----
ENTROPY_CASES = [(8, ["--preset", "ultrafast", "-p", "1"], True),                                  # BASELINE config 1: the survey's md5
                 (8, ["--preset", "ultrafast", "-p", "1", "--owf", "7", "--threads", "8"], True),  # pictures gathered, several coded per call
                 (3, ["--preset", "ultrafast", "-p", "1", "--no-wpp"], True),                       # one substream per picture
                 (2, ["--preset", "ultrafast", "-p", "1", "-q", "32", "--pu-depth-intra", "2-4"], True),  # NxN CUs, levels priced with the CABAC model
                 (2, ["--preset", "ultrafast", "-p", "1", "--tiles", "2x2"], True),                 # tiles: every tile a substream, all but the last end in end_of_subset_one_bit
                 (2, ["--preset", "veryfast", "-p", "1"], True),                                    # SAO syntax from the device's own SAO decision
                 (2, ["--preset", "medium", "-p", "1"], True),                                      # BASELINE config 3's preset: RDOQ levels, NxN CUs, SAO
                 (2, ["--preset", "veryfast", "-p", "1", "--sao", "edge"], False)]                  # a SAO mode the device's decision does not model: the host codes


@pytest.mark.parametrize("frames,opts,on_device", ENTROPY_CASES, ids=["ultrafast", "owf7", "no-wpp", "nxn-qp32", "tiles", "veryfast-sao", "medium", "sao-edge-falls-back"])
def test_binding_with_device_entropy_coding_writes_the_reference_bitstream(tmp_path, frames, opts, on_device):
    if not os.path.exists(os.path.join(REF, "kvazaar_hipsim")):
        pytest.skip("oracle/_ref/kvazaar_hipsim not built (oracle/Makefile, where /root/reference exists)")
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, 416, 240, frames, 1234, "small")
    opts = opts + ([] if "--threads" in opts else ["--threads", "4"])
    want = _encode("kvazaar_ref", yuv, "416x240", str(tmp_path / "ref.hevc"), opts)
    got = _encode("kvazaar_hipsim", yuv, "416x240", str(tmp_path / "sim.hevc"), opts,
                  {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_ENTROPY": "1", "KVZ_HIP_ENTROPY_TRACE": str(tmp_path / "trace")})
    assert got == want
    if frames == 8 and "--owf" not in opts:
        assert got == "9aeb72382ab3092e285ce3f97f51d4ea"
    coded = int(open(str(tmp_path / "trace")).read()) if os.path.exists(str(tmp_path / "trace")) else 0
    assert (coded >= frames) if on_device else coded == 0


def test_fuzz_of_the_binding_against_the_reference_encoder():
    """tools/fuzz_binding.py: random command lines (every preset, all-intra or a low-delay GOP, --qp, --owf, --no-wpp, tiles, SAO modes, RDOQ, PU depths, --subme,
    --fast-residual-cost, motion-search switches, rate control ...) through kvazaar_hipsim and kvazaar_ref: whatever the binding decides -- take the pictures or leave them to
    kvz_search_lcu -- the file must be the reference encoder's"""
    if not os.path.exists(os.path.join(REF, "kvazaar_hipsim")):
        pytest.skip("oracle/_ref/kvazaar_hipsim not built (oracle/Makefile, where /root/reference exists)")
    import sys
    r = subprocess.run([sys.executable, os.path.join(flatapi.ROOT, "tools", "fuzz_binding.py"), "40", "8"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "0 of 40 rounds differ" in r.stdout
