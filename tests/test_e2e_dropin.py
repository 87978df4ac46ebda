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
import synth

pytestmark = pytest.mark.gpu
REF = os.path.join(flatapi.ROOT, "oracle", "_ref")
GOLDEN_416x240_8F = "9aeb72382ab3092e285ce3f97f51d4ea"  # 416x240x8 --preset ultrafast -p 1


def _encode(binary, yuv, out, extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    t = time.time()
    r = subprocess.run([os.path.join(REF, binary), "-i", yuv, "--input-res", "416x240", "-o", out] + extra,
                       env=e, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    return hashlib.md5(open(out, "rb").read()).hexdigest(), time.time() - t, r.stderr


@pytest.mark.parametrize("frames,preset", [(2, ["--preset", "ultrafast", "-p", "1"]),            # BASELINE config 1
                                           (1, ["--preset", "medium", "-p", "1"]),               # config 3: rdoq, sao, 4x4 DST
                                           (3, ["--preset", "veryfast", "--gop", "lp-g4d3t1"]),  # config 4: ME, FME, bipred
                                           (1, ["--preset", "ultrafast", "-p", "1", "--tiles", "2x2"])],  # config 5 (tiles)
                         ids=["ultrafast-intra", "medium-intra", "veryfast-inter", "ultrafast-tiles"])
def test_bitstream_identical_to_generic(tmp_path, frames, preset):
    if not os.path.exists(os.path.join(REF, "kvazaar_hip")):
        pytest.skip("oracle/_ref/kvazaar_hip not built")
    yuv = str(tmp_path / "syn.yuv")
    synth.write_yuv(yuv, 416, 240, frames, 1234, "small")
    common = preset + ["--threads", "4"]
    md5_gen, t_gen, _ = _encode("kvazaar_ref", yuv, str(tmp_path / "gen.hevc"), common + ["--no-cpuid"])
    md5_hip, t_hip, err = _encode("kvazaar_hip", yuv, str(tmp_path / "hip.hevc"), common, {"KVZ_HIP_STATS": "1"})
    assert os.path.getsize(str(tmp_path / "hip.hevc")) > 2000
    print(f"generic {t_gen:.2f}s  hip {t_hip:.2f}s  {[l for l in err.splitlines() if 'kvz_hip' in l]}")
    assert "strategy calls served" in err and " 0 strategy calls" not in err, "hip strategy was not exercised"
    assert md5_hip == md5_gen


def test_golden_md5_416x240(tmp_path):
    """the survey's recorded md5 for BASELINE config 1 (8 frames) reproduced through the hip strategy"""
    if not os.path.exists(os.path.join(REF, "kvazaar_hip")):
        pytest.skip("oracle/_ref/kvazaar_hip not built")
    yuv = str(tmp_path / "syn.yuv")
    assert synth.write_yuv(yuv, 416, 240, 8, 1234, "small") == synth.MD5["416x240"]
    md5_hip, t, err = _encode("kvazaar_hip", yuv, str(tmp_path / "hip.hevc"), ["--preset", "ultrafast", "-p", "1", "--threads", "8"])
    assert md5_hip == GOLDEN_416x240_8F
