"""strategies-encode.h:49-65 kvz_encode_coeff_nxn, the per-call form of the entropy coder's residual syntax: kvz_hip_coeff_nxn_bins (kvazaar_amd/csrc/kvz_entropy.hpp
CoeffBinsOp) returns one block's bins as records; integration/kvazaar/strategies/hip/encode-hip.c feeds them to kvazaar's arithmetic coder.  Checked here:
  the oracle's records, run through the oracle's arithmetic coder, are the bytes the REFERENCE's kvz_encode_coeff_nxn leaves in a real bitstream for the block
  (oracle/ref_shim.c kvz_ref_encode_coeff_nxn_bytes, live where oracle/_ref exists; tests/golden/encode_coeff_nxn.json everywhere);
  the device op compiled for the host and -- under -m gpu -- the device itself return the oracle's records up to how runs of bypass bins are cut into records (the device
  gathers consecutive runs into records of up to 16 bins: fewer steps for the coder, the same bins), and those records code to the reference's bytes too."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import flatapi
from flatapi import A, ptr

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_PATH = os.path.join(HERE, "golden", "encode_coeff_nxn.json")
CAP = 8192


def blocks():
    """(label, width, type, scan_mode, levels): sparse and dense blocks, large levels (escape codes beyond 16 bins), single coefficients at the corners, the scan orders
    kvz_get_scan_order can ask for (horizontal / vertical only for 4x4 and 8x8 luma and 4x4 chroma)"""
    rng = np.random.default_rng(77)
    for w in (4, 8, 16, 32):
        for t in (0, 2):
            for scan in ((0, 1, 2) if (w <= 8 and (t == 0 or w == 4)) else (0,)):
                for kind in ("sparse", "dense", "huge", "corner", "last", "band"):
                    c = np.zeros((w, w), np.int16)
                    if kind == "sparse":
                        m = rng.random((w, w)) < 0.08
                        c[m] = rng.integers(-3, 4, m.sum())
                    elif kind == "dense":
                        c[:] = rng.integers(-9, 10, (w, w))
                    elif kind == "huge":
                        m = rng.random((w, w)) < 0.3
                        c[m] = rng.integers(-32768, 32768, m.sum())
                    elif kind == "corner":
                        c[0, 0] = -1
                    elif kind == "last":
                        c[w - 1, w - 1] = 2
                    else:
                        c[: max(1, w // 4), :] = rng.integers(-40, 41, (max(1, w // 4), w))
                    if not c.any():
                        c[0, 0] = 1
                    yield (f"{w}x{w}-t{t}-s{scan}-{kind}", w, t, scan, A(c.reshape(-1)))


def context_states(seed):
    """150 packed context states (state << 1 | mps), any of the 126 a regular context can be in"""
    return A(np.random.default_rng(seed).integers(0, 126, 150).astype(np.uint8))


def records_of(lib, w, t, scan, c):
    rec = A(np.zeros(CAP, np.uint32))
    n = lib.coeff_nxn_bins(ptr(c), w, t, scan, ptr(rec), CAP)
    assert 0 < n <= CAP
    return rec[:n].copy()


def canonical(rec):
    """a record list as the bins it stands for: context-coded and terminating records as they are, every maximal run of bypass records as (bins, value)"""
    out, run_n, run_v = [], 0, 0
    for r in (int(v) for v in rec):
        kind = r >> 30
        if kind == 1:
            n = (r >> 16) & 0x3f
            run_v, run_n = (run_v << n) | (r & 0xffff), run_n + n
            continue
        if run_n:
            out.append(("ep", run_n, run_v))
            run_n, run_v = 0, 0
        out.append(("ctx", r & 0xff, (r >> 8) & 1) if kind == 0 else ("trm", r & 1))
    if run_n:
        out.append(("ep", run_n, run_v))
    return out


def coded(oracle, ctx, rec):
    f = oracle.lib.kvz_oracle_code_records
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    out = np.zeros(1 << 17, np.uint8)
    rec = np.ascontiguousarray(rec)
    n = f(ctx.ctypes.data, rec.ctypes.data, len(rec), out.ctypes.data, out.size)
    return out[:n].tobytes()


@pytest.fixture(scope="module")
def oracle():
    lib = flatapi.load_oracle()
    lib.lib.kvz_oracle_coeff_nxn_bins.restype = C.c_int
    return lib


def digest_all(oracle):
    out = {}
    for i, (label, w, t, scan, c) in enumerate(blocks()):
        out[label] = hashlib.sha256(coded(oracle, context_states(i), records_of(oracle, w, t, scan, c))).hexdigest()[:24]
    return out


def test_oracle_records_code_to_the_reference_bytes_golden(oracle):
    assert digest_all(oracle) == json.load(open(GOLDEN_PATH))


@pytest.mark.skipif(not os.path.exists(flatapi.refshim_path()), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_records_code_to_the_reference_bytes_live(oracle):
    ref = flatapi.load_ref(0)
    f = ref.lib.kvz_ref_encode_coeff_nxn_bytes
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    for i, (label, w, t, scan, c) in enumerate(blocks()):
        ctx = context_states(i)
        out = np.zeros(1 << 17, np.uint8)
        n = f(ctx.ctypes.data, c.ctypes.data, w, t, scan, out.ctypes.data, out.size)
        assert out[:n].tobytes() == coded(oracle, ctx, records_of(oracle, w, t, scan, c)), label


def test_host_simulation_of_the_device_op_returns_the_oracles_records(oracle):
    import subprocess
    d = os.path.join(flatapi.ROOT, "tests", "hostsim")
    so = os.path.join(d, "libkvz_hostsim.so")
    srcs = [os.path.join(d, "hostsim.cpp")] + [os.path.join(flatapi.ROOT, "kvazaar_amd", "csrc", f) for f in os.listdir(os.path.join(flatapi.ROOT, "kvazaar_amd", "csrc"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, os.path.join(d, "hostsim.cpp")])
    sim = flatapi.FlatLib(so, "kvz_hostsim_")
    golden = json.load(open(GOLDEN_PATH))
    for i, (label, w, t, scan, c) in enumerate(blocks()):
        rec = records_of(sim, w, t, scan, c)
        assert canonical(rec) == canonical(records_of(oracle, w, t, scan, c)), label
        assert all(((int(r) >> 16) & 0x3f) <= 16 for r in rec if (int(r) >> 30) == 1), label
        assert hashlib.sha256(coded(oracle, context_states(i), rec)).hexdigest()[:24] == golden[label], label


@pytest.mark.gpu
def test_device_op_returns_the_oracles_records(oracle):
    hip = flatapi.FlatLib(flatapi.hip_path(), "kvz_hip_")
    golden = json.load(open(GOLDEN_PATH))
    for i, (label, w, t, scan, c) in enumerate(blocks()):
        rec = records_of(hip, w, t, scan, c)
        assert canonical(rec) == canonical(records_of(oracle, w, t, scan, c)), label
        assert hashlib.sha256(coded(oracle, context_states(i), rec)).hexdigest()[:24] == golden[label]


if __name__ == "__main__":  # writes the fixture: the digests are taken from the REFERENCE's bytes (the live test above must pass in the same run)
    import sys
    o = flatapi.load_oracle()
    o.lib.kvz_oracle_coeff_nxn_bins.restype = C.c_int
    ref = flatapi.load_ref(0)
    f = ref.lib.kvz_ref_encode_coeff_nxn_bytes
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    gold = {}
    for i, (label, w, t, scan, c) in enumerate(blocks()):
        ctx = context_states(i)
        out = np.zeros(1 << 17, np.uint8)
        n = f(ctx.ctypes.data, c.ctypes.data, w, t, scan, out.ctypes.data, out.size)
        assert out[:n].tobytes() == coded(o, ctx, records_of(o, w, t, scan, c)), label
        gold[label] = hashlib.sha256(out[:n].tobytes()).hexdigest()[:24]
    json.dump(gold, open(GOLDEN_PATH, "w"), indent=0, sort_keys=True)
    print("wrote", GOLDEN_PATH, len(gold), "blocks")
