"""The entropy coder's alternative forms against the oracle, on the CPU (device sources in host simulation, tools/fuzz_entropy.py on a few seeded rounds each):
  * stage 1 as the phased walk (the product: coefficient groups / block openings / tree nodes, deferred flags and blocks in a queue) and as the serial walk
    (KVZ_HIP_ENTROPY_BINS=serial) write lists that code to the oracle's bytes;
  * the coder may move a byte out whenever eight bits have gathered (bits_left <= 15), not only at kvz_cabac_write's bits_left < 12: with KVZ_HOSTSIM_EARLY_WRITE every
    lane does so as early as it may -- the rule the device's wavefront-aligned byte output rests on (kvz_entropy.hpp entropy_move_bytes)."""
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


@pytest.mark.parametrize("env", [{}, {"KVZ_HIP_ENTROPY_BINS": "serial"}, {"KVZ_HOSTSIM_EARLY_WRITE": "1"}, {"KVZ_HOSTSIM_EARLY_WRITE": "1", "KVZ_HIP_ENTROPY_BINS": "serial"}],
                         ids=["phased", "serial", "phased-early-bytes", "serial-early-bytes"])
def test_forms_code_the_oracles_bytes(env, hostsim_built):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_entropy.py"), "6", "31"], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "differences: 0" in r.stdout, r.stdout[-2000:]
