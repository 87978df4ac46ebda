"""Drop-in end to end (-m gpu): the REAL kvazaar encoder (reference sources built by oracle/Makefile into
oracle/_ref/) with the hip strategy registered through integration/kvazaar/strategies/hip/*.c must produce the
byte-identical bitstream of the generic strategy on the same YUV (SURVEY.md section 0 "end-to-end parity bar",
8c md5s).  Every strategy call of the encode goes through libkvz_hip.so's synchronous per-call path."""
import hashlib
import os
import subprocess
import time

import pytest

import flatapi
from kvazaar_amd import synth

pytestmark = pytest.mark.gpu
REF = os.path.join(flatapi.ROOT, "oracle", "_ref")
GOLDEN_416x240_8F = "9aeb72382ab3092e285ce3f97f51d4ea"  # 416x240x8 --preset ultrafast -p 1


def _need_hip_encoder():
    """oracle/_ref is git-ignored but travels to the GPU box with gpurun; under -m gpu its absence is a failure, not a skip -- a
    green run without the bitstream tests would look like a green run with them"""
    if not os.path.exists(os.path.join(REF, "kvazaar_hip")) or not os.path.exists(os.path.join(REF, "kvazaar_ref")):
        pytest.fail("oracle/_ref/kvazaar_hip / kvazaar_ref not built: run `python -c 'import __graft_entry__ as g; g.build()'` where "
                    "/root/reference exists (the built files ship with the snapshot)")


def _encode(binary, yuv, out, extra, env=None, res="416x240"):
    e = dict(os.environ)
    e.update(env or {})
    t = time.time()
    r = subprocess.run([os.path.join(REF, binary), "-i", yuv, "--input-res", res, "-o", out] + extra,
                       env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), time.time() - t, r.stderr


@pytest.mark.parametrize("frames,preset", [(2, ["--preset", "ultrafast", "-p", "1"]),            # BASELINE config 1
                                           (1, ["--preset", "medium", "-p", "1"]),               # config 3: rdoq, sao, 4x4 DST
                                           (3, ["--preset", "veryfast", "--gop", "lp-g4d3t1"]),  # config 4: ME, FME, bipred
                                           (1, ["--preset", "ultrafast", "-p", "1", "--tiles", "2x2"]),  # config 5 (tiles)
                                           # --scaling-list default: quant / dequant / quantize_residual take the per-coefficient tables (quant-generic.c:59-60, 309-333)
                                           (1, ["--preset", "ultrafast", "-p", "1", "--scaling-list", "default"]),
                                           (2, ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--scaling-list", "default"])],  # ... the inter lists too
                         ids=["ultrafast-intra", "medium-intra", "veryfast-inter", "ultrafast-tiles", "ultrafast-scaling-list", "veryfast-inter-scaling-list"])
def test_bitstream_identical_to_generic(tmp_path, frames, preset):
    _need_hip_encoder()
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, 416, 240, frames, 1234, "small")
    common = preset + ["--threads", "4"]
    md5_gen, t_gen, _ = _encode("kvazaar_ref", yuv, str(tmp_path / "gen.hevc"), common + ["--no-cpuid"])
    md5_hip, t_hip, err = _encode("kvazaar_hip", yuv, str(tmp_path / "hip.hevc"), common, {"KVZ_HIP_STATS": "1", "KVZ_HIP_DROPIN": "1"})  # the per-call strategies are opt-in (hip-common.h)
    assert os.path.getsize(str(tmp_path / "hip.hevc")) > 2000
    print(f"generic {t_gen:.2f}s  hip {t_hip:.2f}s  {[l for l in err.splitlines() if 'kvz_hip' in l]}")
    assert "strategy calls served" in err and " 0 strategy calls" not in err, "hip strategy was not exercised"
    assert md5_hip == md5_gen


def test_golden_md5_416x240(tmp_path):
    """the survey's recorded md5 for BASELINE config 1 (8 frames) reproduced through the hip strategy"""
    _need_hip_encoder()
    yuv = str(tmp_path / "syn.yuv")
    assert synth.write_yuv(yuv, 416, 240, 8, 1234, "small") == synth.MD5["416x240"]
    md5_hip, t, err = _encode("kvazaar_hip", yuv, str(tmp_path / "hip.hevc"), ["--preset", "ultrafast", "-p", "1", "--threads", "8"], {"KVZ_HIP_DROPIN": "1", "KVZ_HIP_STATS": "1"})
    assert "strategy calls served" in err and " 0 strategy calls" not in err, "hip strategy was not exercised"
    assert md5_hip == GOLDEN_416x240_8F


@pytest.mark.parametrize("frames,extra", [(3, ["-q", "22"]), (2, ["-q", "32"]), (2, ["-q", "27", "--no-deblock"]), (1, ["-q", "22", "--tiles", "2x2", "--wpp"]), (2, ["-q", "22", "--tiles", "2x2"]), (2, ["-q", "32", "--no-wpp"])],
                         ids=["qp22", "qp32-cabac-coeff-cost", "qp27-nodeblock", "tiles-wpp", "tiles", "no-wpp"])
def test_batched_search_bitstream_identical(tmp_path, frames, extra):
    """The BATCHED pass inside the real encoder: integration/kvazaar/search_lcu_hip.c stands in front of kvz_search_lcu and
    fills cu_array / reconstruction / coefficients of every LCU from one kvz_hip_intra_frames() run per picture; deblocking and
    the entropy coder are the reference's own.  The bitstream must be the reference encoder's, byte for byte -- i.e. every CU
    depth, mode, coded block flag and coefficient the device decided is kvazaar's.  (The per-call strategies are switched off
    here: this is about the batched path.)  Tiles: each tile is a picture of its own for the pass; kvazaar switches WPP off
    when tiles are requested (cfg.c:925-978), which the pass follows (kvz_hip_intra_cost_model::no_wpp)."""
    _need_hip_encoder()
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, 416, 240, frames, 1234, "small")
    common = ["--preset", "ultrafast", "-p", "1", "--threads", "4"] + extra
    md5_ref, _, _ = _encode("kvazaar_ref", yuv, str(tmp_path / "ref.hevc"), common)
    md5_plain, _, _ = _encode("kvazaar_hip", yuv, str(tmp_path / "plain.hevc"), common, {"KVZ_HIP_DISABLE": "1"})
    md5_batch, t, _ = _encode("kvazaar_hip", yuv, str(tmp_path / "batch.hevc"), common, {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_TRACE": str(tmp_path / "trace")})
    assert md5_plain == md5_ref
    assert md5_batch == md5_ref
    assert int(open(str(tmp_path / "trace")).read().split()[0]) >= frames, "the batched search was not used"


@pytest.mark.parametrize("preset,extra", [("superfast", ["-q", "22"]), ("veryfast", ["-q", "32"]), ("faster", ["-q", "22"]), ("faster", ["-q", "37", "--no-wpp"]),
                                          ("fast", ["-q", "22"]), ("fast", ["-q", "32", "--tiles", "2x2"]), ("medium", ["-q", "27", "--pu-depth-intra", "1-3"]),
                                          ("medium", ["-q", "22"]), ("medium", ["-q", "12"]), ("ultrafast", ["-q", "22", "--pu-depth-intra", "2-4"])],
                         ids=["superfast-qp22", "veryfast-qp32", "faster-qp22", "faster-qp37-nowpp", "fast-qp22", "fast-qp32-tiles", "medium-pu13-qp27",
                              "medium-qp22", "medium-qp12-nxn", "ultrafast-pu24-nxn"])
def test_batched_search_other_all_intra_presets(tmp_path, preset, extra):
    """All-intra superfast / veryfast (= the ultrafast search + `--sao full`), faster (fast-residual-cost 0: coefficients priced with the
    CABAC model at every QP), fast (--pu-depth-intra 1-3 on top: 32x32 CUs searched), medium without its NxN partitions (--rdoq on top: kvz_rdoq in every
    quantisation of the device's search) and medium as it is (BASELINE config 3's preset: + 8x8 CUs tried as four 4x4 PUs) with the device searching whole pictures: the reference's SAO decision then runs on the host on the device's
    reconstruction, and the bitstream must still be the reference encoder's, byte for byte."""
    _need_hip_encoder()
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, 416, 240, 2, 1234, "small")
    common = ["--preset", preset, "-p", "1", "--threads", "4"] + extra
    md5_ref, _, _ = _encode("kvazaar_ref", yuv, str(tmp_path / "ref.hevc"), common)
    md5_batch, t, _ = _encode("kvazaar_hip", yuv, str(tmp_path / "batch.hevc"), common, {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_TRACE": str(tmp_path / "trace")})
    assert md5_batch == md5_ref
    assert int(open(str(tmp_path / "trace")).read().split()[0]) >= 2, "the batched search was not used"


def test_batched_search_gathers_pictures_in_flight(tmp_path):
    """--owf 7: eight pictures in flight ask for their results at about the same time; the binding gathers them into ONE device pass per group
    (search_lcu_hip.c "frame batching") and the bitstream is still the reference's"""
    _need_hip_encoder()
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, 416, 240, 16, 1234, "small")
    common = ["--preset", "ultrafast", "-p", "1", "--threads", "8", "--owf", "7"]
    md5_ref, _, _ = _encode("kvazaar_ref", yuv, str(tmp_path / "ref.hevc"), common)
    md5_batch, _, _ = _encode("kvazaar_hip", yuv, str(tmp_path / "batch.hevc"), common,
                              {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_TRACE": str(tmp_path / "trace"), "KVZ_HIP_BATCH_WINDOW_US": "40000"})
    assert md5_batch == md5_ref
    pictures, passes, largest = (int(v) for v in open(str(tmp_path / "trace")).read().split())
    assert pictures == 16 and passes < 16 and largest > 1, (pictures, passes, largest)


def test_batched_search_golden_md5_416x240(tmp_path):
    """the survey's recorded md5 for BASELINE config 1 (8 frames) reproduced with the device searching whole pictures"""
    _need_hip_encoder()
    yuv = str(tmp_path / "syn.yuv")
    assert synth.write_yuv(yuv, 416, 240, 8, 1234, "small") == synth.MD5["416x240"]
    md5, _, _ = _encode("kvazaar_hip", yuv, str(tmp_path / "b.hevc"), ["--preset", "ultrafast", "-p", "1", "--threads", "8"],
                        {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1"})
    assert md5 == GOLDEN_416x240_8F


# the bitstream md5s of the reference encoder at BASELINE's own picture sizes (identical for generic and AVX2)
GOLDEN_AT_SIZE = [("1920x1080", 8, 1, ["--preset", "ultrafast", "-p", "1"], "dce84d2200dc0e54e2e029425d1682e1"),
                  ("3840x2160", 4, 2, ["--preset", "ultrafast", "-p", "1", "--tiles", "4x2"], "3de25813427ad6fc374bb50351d48dd9")]


@pytest.mark.parametrize("res,frames,seed,opts,md5", GOLDEN_AT_SIZE, ids=["1080p-x8-ultrafast", "2160p-x4-tiles4x2"])
def test_batched_search_golden_md5_at_baseline_sizes(tmp_path, res, frames, seed, opts, md5):
    """BASELINE configs 2 and 5 at their real geometry through the binding (integration/kvazaar/search_lcu_hip.c: slot table, cbf rebuild, tile views of
    960x1088 / 960x1072 luma): the device searches every picture / tile, the reference's entropy coder writes the survey's bitstream byte for byte"""
    _need_hip_encoder()
    w, h = (int(v) for v in res.split("x"))
    yuv = str(tmp_path / "syn.yuv")
    assert synth.write_yuv(yuv, w, h, frames, seed, "large") == synth.MD5[res]
    got, t, _ = _encode("kvazaar_hip", yuv, str(tmp_path / "b.hevc"), opts + ["--threads", "16"],
                        {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_TRACE": str(tmp_path / "trace")}, res=res)
    assert got == md5
    assert int(open(str(tmp_path / "trace")).read().split()[0]) >= frames, "the batched search was not used"


# ---- BASELINE config 4: B pictures searched by the inter CTU pass on the device, inside the real encoder ----
INTER_CASES = [("416x240", 8, 1234, "small", ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0"], 7),
               ("416x240", 6, 1234, "small", ["--preset", "ultrafast", "--gop", "lp-g4d3t1", "--owf", "0", "-q", "20"], 5),
               ("416x240", 5, 1234, "small", ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0", "--no-wpp"], 4),
               ("416x240", 10, 1234, "small", ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0", "--period", "8"], 8),
               ("416x240", 6, 1234, "small", ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0", "-q", "32"], 5),   # picture QPs 35-38: coefficients priced with the CABAC model
               ("416x240", 6, 1234, "small", ["--preset", "ultrafast", "--gop", "lp-g4d3t1", "--owf", "0", "-q", "24"], 5),  # picture QPs on both sides of fast-residual-cost 28
               ("416x240", 6, 1234, "small", ["--preset", "faster", "--gop", "lp-g4d3t1", "--owf", "0"], 5),                 # subme 4 (quarter-sample search), fast-residual-cost 0
               ("1920x1080", 5, 1, "large", ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0"], 4),
               ("3840x2160", 4, 2, "large", ["--preset", "veryfast", "--gop", "lp-g4d3t1", "--owf", "0"], 3)]


@pytest.mark.parametrize("res,frames,seed,kind,opts,device_pictures", INTER_CASES,
                         ids=["veryfast-416x240", "ultrafast-qp20", "veryfast-no-wpp", "veryfast-period8", "veryfast-qp32", "ultrafast-qp24-mixed", "faster", "veryfast-1080p", "baseline-c4-2160p"])
def test_inter_pass_inside_the_encoder_bitstream_identical(tmp_path, res, frames, seed, kind, opts, device_pictures):
    """`--preset veryfast --gop lp-g4d3t1` with every picture searched on the device: the I picture by the batched intra pass, the B pictures by
    kvz_hip_dev_inter_ctu_pass (integration/kvazaar/search_lcu_hip.c search_lcu_inter), each from kvazaar's own deblocked + SAO-filtered reference picture and its
    cu_array.  kvazaar's loop filters and entropy coder then run on the device's CU records, reconstruction and coefficients, and the bitstream must be the
    reference encoder's byte for byte -- every motion vector, merge index, skip flag, intra mode, coded block flag and coefficient of every CU.  (--owf 0: the pass
    needs the complete reference picture when a picture's first LCU is searched.)"""
    _need_hip_encoder()
    w, h = (int(v) for v in res.split("x"))
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, w, h, frames, seed, kind)
    common = opts + ["--threads", "8"]
    md5_ref, _, _ = _encode("kvazaar_ref", yuv, str(tmp_path / "ref.hevc"), common, res=res)
    md5_dev, t, _ = _encode("kvazaar_hip", yuv, str(tmp_path / "dev.hevc"), common,
                            {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_INTER_TRACE": str(tmp_path / "trace")}, res=res)
    assert md5_dev == md5_ref
    assert int(open(str(tmp_path / "trace")).read()) >= device_pictures, "the inter pass was not used"
    if res == "416x240" and frames == 8:
        assert md5_dev == "1e7a81653d1157ce05e9ffd85c7cd5cf"


# ---- the device's entropy coder inside the real encoder (KVZ_HIP_BATCH_ENTROPY=1) ----
@pytest.mark.parametrize("res,frames,seed,kind,opts", [("416x240", 8, 1234, "small", ["--preset", "ultrafast", "-p", "1"]),
                                                        ("416x240", 3, 1234, "small", ["--preset", "ultrafast", "-p", "1", "--no-wpp"]),
                                                        ("416x240", 2, 1234, "small", ["--preset", "ultrafast", "-p", "1", "-q", "32", "--pu-depth-intra", "2-4"]),
                                                        ("1920x1080", 8, 1, "large", ["--preset", "ultrafast", "-p", "1", "--owf", "7"]),
                                                        ("416x240", 3, 1234, "small", ["--preset", "ultrafast", "-p", "1", "--tiles", "2x2"]),
                                                        ("416x240", 3, 1234, "small", ["--preset", "veryfast", "-p", "1"]),
                                                        ("416x240", 2, 1234, "small", ["--preset", "medium", "-p", "1"]),
                                                        ("1920x1080", 4, 1, "large", ["--preset", "medium", "-p", "1", "--owf", "3"]),
                                                        ("3840x2160", 4, 2, "large", ["--preset", "ultrafast", "-p", "1", "--tiles", "4x2"])],
                         ids=["survey-416x240", "no-wpp", "nxn-qp32", "1080p-x8-owf7", "tiles-2x2", "veryfast-sao", "medium", "medium-1080p", "2160p-tiles-4x2"])
def test_device_entropy_coding_inside_the_encoder_bitstream_identical(tmp_path, res, frames, seed, kind, opts):
    """The slice data of every picture written on the device (kvz_hip_batch_entropy_code behind integration/kvazaar/search_lcu_hip.c): the levels never leave the device,
    kvz_encode_coding_tree does not run, the row coders' streams are replaced by the device's substreams before kvazaar writes the slice header (whose entry points are
    their sizes), parameter sets, SEI and NAL framing.  The file must be the reference encoder's, byte for byte."""
    _need_hip_encoder()
    w, h = (int(v) for v in res.split("x"))
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, w, h, frames, seed, kind)
    common = opts + ["--threads", "8"]
    md5_ref, _, _ = _encode("kvazaar_ref", yuv, str(tmp_path / "ref.hevc"), common, res=res)
    md5_dev, t, _ = _encode("kvazaar_hip", yuv, str(tmp_path / "dev.hevc"), common,
                            {"KVZ_HIP_DISABLE": "1", "KVZ_HIP_BATCH_SEARCH": "1", "KVZ_HIP_BATCH_ENTROPY": "1", "KVZ_HIP_ENTROPY_TRACE": str(tmp_path / "trace")}, res=res)
    assert md5_dev == md5_ref
    assert int(open(str(tmp_path / "trace")).read()) >= frames, "the device entropy coder was not used"
    if res == "3840x2160":
        assert md5_dev == "3de25813427ad6fc374bb50351d48dd9"  # BASELINE config 5
    if res == "416x240" and frames == 8:
        assert md5_dev == GOLDEN_416x240_8F
    if res == "1920x1080" and "ultrafast" in opts:
        assert md5_dev == "dce84d2200dc0e54e2e029425d1682e1"
