"""The entropy coder's alternative forms against the oracle, on the CPU (device sources in host simulation, tools/fuzz_entropy.py on a few seeded rounds each):
  * stage 1 as the phased walk (the product: coefficient groups / block openings / tree nodes, deferred flags and blocks in a queue) and as the serial walk
    (KVZ_HIP_ENTROPY_BINS=serial) write lists that code to the oracle's bytes;
  * the coder moves its code value out 32 bits at a time, a lane when it must or when a neighbour must (kvz_entropy.hpp WideCoder): as late as a lane must (the default
    here), as early as it may (KVZ_HOSTSIM_WIDE_EARLY), and in byte units (KVZ_HOSTSIM_WIDE_UNIT=8), where a carry into an all-ones unit -- one case in 2^32 with 32-bit
    units -- happens all the time;
  * emulation prevention as the device does it -- by position, in 256 chunks -- against the serial rule of bitstream.c:212-223."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hostsim_built():
    """tests/hostsim/libkvz_hostsim.so, current (the recipe of tests/test_hostsim.py)"""
    d = os.path.join(ROOT, "tests", "hostsim")
    so = os.path.join(d, "libkvz_hostsim.so")
    csrc = os.path.join(ROOT, "kvazaar_amd", "csrc")
    srcs = [os.path.join(d, "hostsim.cpp")] + [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, os.path.join(d, "hostsim.cpp")])
    return so


@pytest.mark.parametrize("env", [{}, {"KVZ_HIP_ENTROPY_BINS": "serial"}, {"KVZ_HOSTSIM_WIDE_EARLY": "1"}, {"KVZ_HOSTSIM_WIDE_UNIT": "8"},
                                 {"KVZ_HOSTSIM_WIDE_UNIT": "8", "KVZ_HIP_ENTROPY_BINS": "serial"}],
                         ids=["phased", "serial", "units-early", "byte-units", "serial-byte-units"])
def test_forms_code_the_oracles_bytes(env, hostsim_built):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_entropy.py"), "6", "31"], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "differences: 0" in r.stdout, r.stdout[-2000:]


def serial_escape(raw):
    out, zeros = bytearray(), 0
    for b in raw:
        if zeros == 2 and b < 4:
            out.append(3)
            zeros = 0
        zeros = zeros + 1 if b == 0 else 0
        out.append(b)
    return bytes(out)


def test_emulation_prevention_by_position_equals_the_serial_rule(hostsim_built):
    import ctypes as C

    import numpy as np
    sim = C.CDLL(hostsim_built)
    f = sim.kvz_hostsim_escape
    f.restype = C.c_long
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    rng = np.random.default_rng(5)
    cases = [bytes(n) for n in (0, 1, 2, 3, 4, 5, 6, 7, 255, 256, 257, 513, 1000)]                       # runs of zeros only, across every chunk border
    cases += [bytes([0, 0, v]) for v in range(6)] + [bytes([0] * k + [v]) for k in range(2, 9) for v in (0, 1, 3, 4)]
    for n in (9, 100, 255, 256, 257, 258, 511, 512, 513, 1024, 5000, 26000):
        for alphabet in ((0, 1), (0, 0, 0, 3, 4), (0, 0, 1, 2, 3, 4, 255), tuple(range(256))):
            cases.append(bytes(rng.choice(np.array(alphabet, np.uint8), n).tolist()))
    bad = []
    for raw in cases:
        want = serial_escape(raw)
        src = np.frombuffer(raw + b"\0", np.uint8).copy()
        out = np.zeros(len(raw) * 3 // 2 + 8, np.uint8)
        n = f(src.ctypes.data, len(raw), out.ctypes.data)
        if n != len(want) or out[:n].tobytes() != want:
            bad.append((len(raw), n, len(want)))
    assert not bad, bad[:10]
