"""Kernel logic without a GPU: the device ops of kvazaar_amd/csrc/kvz_ops.hpp and the per-call op sequences of
kvz_api_impl.hpp, compiled for the host by tests/hostsim (every op run as a plain loop over its items), must
match the oracle bit-for-bit on all seeded cases.  The GPU build of the same sources is checked against the
same cases in tests/test_gpu_parity.py (-m gpu)."""
import os
import subprocess

import numpy as np
import pytest

import cases
import flatapi

HOSTSIM_DIR = os.path.join(flatapi.ROOT, "tests", "hostsim")
HOSTSIM_SO = os.path.join(HOSTSIM_DIR, "libkvz_hostsim.so")


@pytest.fixture(scope="session")
def hostsim():
    srcs = [os.path.join(HOSTSIM_DIR, "hostsim.cpp")] + [os.path.join(flatapi.ROOT, "kvazaar_amd", "csrc", f)
                                                         for f in os.listdir(os.path.join(flatapi.ROOT, "kvazaar_amd", "csrc"))]
    if not os.path.exists(HOSTSIM_SO) or any(os.path.getmtime(s) > os.path.getmtime(HOSTSIM_SO) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", HOSTSIM_SO,
                               os.path.join(HOSTSIM_DIR, "hostsim.cpp")])
    return flatapi.FlatLib(HOSTSIM_SO, "kvz_hostsim_")


@pytest.mark.parametrize("gen", cases.ALL_GENERATORS, ids=lambda g: g.__name__)
def test_hostsim_equals_oracle(oracle, hostsim, gen):
    bad, n = [], 0
    for label, run in gen():
        n += 1
        if run(oracle) != run(hostsim):
            bad.append(label)
    assert not bad, f"{len(bad)}/{n} cases differ: {bad[:12]}"


def test_hostsim_find_last_scanpos(oracle, hostsim):
    def st(scan_idx, l2):
        n = 1 << (2 * l2)
        return np.ctypeslib.as_array(oracle.lib.kvz_oracle_scan_table(scan_idx, l2), shape=(n,)).copy()
    bad = [label for label, run in cases.cases_find_last_scanpos(st) if run(oracle) != run(hostsim)]
    assert not bad, bad[:10]
