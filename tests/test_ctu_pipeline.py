"""The batched all-intra CTU pass without a GPU: the CTU program of kvazaar_amd/csrc/kvz_ctu.hpp, compiled for the
host by tests/hostsim (256 "threads" looped per phase), against the independent recursive restatement of kvazaar's
search in oracle/kvz_oracle_ctu.c -- bit-exact reconstruction, coefficients, CU depths, modes and RD costs."""
import ctypes as C

import numpy as np
import pytest

import ctu_common as cc
import flatapi
from test_hostsim import hostsim  # noqa: F401  (fixture)


def oracle_model(oracle, reflib, qp):
    fb = (C.c_float * 128)(*[reflib.lib.kvz_ref_entropy_fbits(i) for i in range(128)])
    m = cc.CostModel()
    f = oracle.lib.kvz_oracle_intra_cost_model
    f.restype = None
    f.argtypes = [C.c_int, C.c_float * 128, C.c_uint64, C.POINTER(cc.CostModel)]
    f(qp, fb, reflib.lib.kvz_ref_fast_coeff_weights(qp), C.byref(m))
    return m


def builtin_model(qp, weights=cc.COEFF_WEIGHTS_QP22):
    """the product's own cost-model builder (kvz_batch.hpp cost_model_init) needs the HIP library; the oracle's builder
    with a Python copy of the entropy table is used for the no-GPU tests instead"""
    raise NotImplementedError


@pytest.fixture(scope="module")
def model22(oracle):
    import test_oracle_vs_ref  # noqa: F401
    if not flatapi.os.path.exists(flatapi.refshim_path()):
        pytest.skip("oracle/_ref not built")
    return oracle_model(oracle, flatapi.load_ref(0), 22)


def test_cost_model_weights_constant(oracle):
    if not flatapi.os.path.exists(flatapi.refshim_path()):
        pytest.skip("oracle/_ref not built")
    assert flatapi.load_ref(0).lib.kvz_ref_fast_coeff_weights(22) == cc.COEFF_WEIGHTS_QP22


@pytest.mark.parametrize("size", [(64, 64), (128, 64), (416, 240), (72, 88)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_hostsim_ctu_equals_oracle_synthetic(oracle, hostsim, model22, size):
    w, h = size
    for i, yuv in enumerate(cc.yuv_frames(w, h, 2, 1234, "small")):
        a, b = cc.run_oracle(oracle, model22, w, h, yuv), cc.run_hostsim(hostsim.lib, model22, w, h, yuv)
        assert not cc.compare(a, b), (size, i, cc.compare(a, b))


def test_hostsim_ctu_equals_oracle_adversarial(oracle, hostsim, model22):
    w, h = 192, 136
    for name, yuv in cc.adversarial_frames(w, h).items():
        a, b = cc.run_oracle(oracle, model22, w, h, yuv), cc.run_hostsim(hostsim.lib, model22, w, h, yuv)
        assert not cc.compare(a, b), (name, cc.compare(a, b))


@pytest.mark.parametrize("qp", [10, 27, 37])
def test_hostsim_ctu_equals_oracle_qps(oracle, hostsim, qp):
    if not flatapi.os.path.exists(flatapi.refshim_path()):
        pytest.skip("oracle/_ref not built")
    m = oracle_model(oracle, flatapi.load_ref(0), qp)
    w, h = 128, 128
    yuv = cc.yuv_frames(w, h, 1, 77, "large")[0]
    a, b = cc.run_oracle(oracle, m, w, h, yuv), cc.run_hostsim(hostsim.lib, m, w, h, yuv)
    assert not cc.compare(a, b), (qp, cc.compare(a, b))


def test_pipeline_quality_is_sane(oracle, model22):
    """the restated search must behave like an encoder: PSNR at QP 22 in the range kvazaar itself reaches on this clip
    (39.65 dB luma for frame 0 of the 416x240 clip with --preset ultrafast -p 1, deblocking on)"""
    w, h = 416, 240
    yuv = cc.yuv_frames(w, h, 1, 1234, "small")[0]
    o = cc.run_oracle(oracle, model22, w, h, yuv)
    y, ry = yuv[:w * h].astype(np.float64), o["rec"][:w * h].astype(np.float64)
    psnr = 10 * np.log10(255 ** 2 / np.mean((y - ry) ** 2))
    assert 39.0 < psnr < 40.5, psnr
    assert set(np.unique(o["depth"])) <= {0, 1, 2, 3}


def test_fuzz_of_the_ctu_program_against_the_oracle(hostsim):
    """tools/fuzz_ctu_sim.py: random pictures and sizes, QP 0..51, every switch of the cost model (CABAC coefficient cost, 32x32 CUs, RDOQ, NxN partitions, WPP, frozen
    contexts) -- the device sources in host simulation must equal the oracle on every output"""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(flatapi.ROOT, "tools", "fuzz_ctu_sim.py"), "120", "6"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 of 120 cases differ" in r.stdout
