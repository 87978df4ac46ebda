"""The committed golden fixtures (tests/golden/, produced by the compiled reference with tests/golden/make_golden.py) against
the oracle -- runs anywhere, no reference tree, no GPU -- and, under -m gpu, against the HIP library directly."""
import json
import os
import sys

import numpy as np
import pytest

import cases
import flatapi

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402


def _fixture(name):
    return json.load(open(os.path.join(HERE, "golden", name)))


def _scan_table(oracle):
    def f(scan_idx, l2):
        n = 1 << (2 * l2)
        return np.ctypeslib.as_array(oracle.lib.kvz_oracle_scan_table(scan_idx, l2), shape=(n,)).copy()
    return f


def _check(lib, oracle):
    want = _fixture("strategy_cases.json")
    got = mg.strategy_digests(lib, _scan_table(oracle))
    assert set(got) == set(want), sorted(set(got) ^ set(want))[:10]
    bad = [k for k in want if got[k] != want[k]]
    assert not bad, f"{len(bad)}/{len(want)} differ: {bad[:12]}"


def test_oracle_reproduces_reference_fixtures(oracle):
    _check(oracle, oracle)
    assert len(_fixture("strategy_cases.json")) > 2000


def test_oracle_deblock_reproduces_reference_fixtures(oracle):
    assert mg.deblock_digests(oracle.lib.kvz_oracle_deblock_frame) == _fixture("deblock.json")


def test_oracle_sao_frame_reproduces_reference_fixtures(oracle):
    assert mg.sao_digests(oracle.lib.kvz_oracle_sao_frame) == _fixture("sao_frame.json")


@pytest.mark.gpu
def test_hip_reproduces_reference_fixtures(oracle):
    import kvazaar_amd
    kvazaar_amd.load_library()
    _check(flatapi.FlatLib(kvazaar_amd.LIB_PATH, "kvz_hip_"), oracle)
