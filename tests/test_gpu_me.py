"""kvz_hip_dev_pu_search (the motion search of a PU, whole: starting points, early termination, hexagon search, half-pel refinement, every accept / reject
decision of the reference) on the MI355X against the oracle's restatement of the same functions (oracle/kvz_oracle_inter.inc, pinned against the reference
encoder CU for CU by tests/test_inter_oracle.py):
  * on random PUs with random candidate lists, both fme levels, with and without the overlapped-picture restriction, probes far outside the picture;
  * on EVERY motion search the sequence oracle runs while it encodes clips of tests/inter_common.py CASES -- inputs and results recorded by its tracer -- so the
    device is checked on exactly the searches of a real encode, against results that are part of an encode equal to kvazaar's.
The CPU half (the contract function reproduces the traced searches) needs no GPU."""
import ctypes as C

import numpy as np
import pytest

import flatapi
import inter_common as ic


@pytest.fixture(scope="module")
def oracle():
    return flatapi.load_oracle()


def device_pu_search(lib, dev, cur, ref, w, h, pus, params):
    lib.kvz_hip_dev_pu_search.restype = C.c_int
    lib.kvz_hip_dev_pu_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    d_cur, d_ref, d_pus = dev.put(cur), dev.put(ref), dev.put(pus)
    d_out = dev.empty(len(pus) * ic.ME_RESULT.itemsize)
    assert lib.kvz_hip_dev_pu_search(d_cur, d_ref, w, h, d_pus, len(pus), 32, C.addressof(params), d_out) == 0
    got = dev.get(d_out, (len(pus),), ic.ME_RESULT)
    dev.free(d_cur, d_ref, d_pus, d_out)
    return got


def random_pus(rng, w, h, n):
    pus = np.zeros(n, ic.ME_PU)
    for i in range(n):
        s = int(rng.choice([8, 16, 32]))
        p = pus[i]
        p["w"] = p["h"] = s
        p["x"], p["y"] = int(rng.integers(0, (w - s) // 8 + 1)) * 8, int(rng.integers(0, (h - s) // 8 + 1)) * 8
        spread = int(rng.choice([2, 8, 40, 200]))
        p["mv_cand"] = rng.integers(-spread, spread + 1, (2, 2))
        if rng.random() < 0.3:
            p["mv_cand"][1] = p["mv_cand"][0]
        p["has_start"] = int(rng.random() < 0.6)
        p["start_mv"] = rng.integers(-spread * 2, spread * 2 + 1, 2)
        p["num_merge"] = int(rng.integers(0, 6))
        p["merge_dir"] = rng.choice([1, 2, 3], 5)
        p["merge_mv"] = rng.integers(-spread, spread + 1, (5, 2))
        if i % 17 == 0:   # a corner PU whose candidates point far outside
            p["x"], p["y"] = (0, 0) if i % 2 else (w - s, h - s)
            p["start_mv"] = (-900, -700) if i % 2 else (800, 1100)
            p["has_start"] = 1
    return pus


def test_contract_function_reproduces_every_traced_search(oracle):
    """CPU: kvz_oracle_pu_motion_search on the recorded inputs == what the sequence encoder computed (the recorder and the contract are the same code paths)"""
    total = 0
    for name in ("fast-pan-owf-qp37", "ultrafast-fast-pan-owf-qp30", "two-gops", "faster-owf-qp27"):
        case = [c for c in ic.CASES if c[0] == name][0]
        _, w, h, n, qp, preset, dbk, sao, owf, src = case
        frames, rf, qps, pus, res, poc = ic.traced_encode(oracle, case)
        for f in range(1, n):
            sel = np.flatnonzero(poc == f)
            prm = ic.MeParams(lambda_sqrt=ic.lambda_sqrt(int(qps[f])), mv_constraint=int(owf > 0), sao=int(sao), deblock=int(dbk), fme_level=ic.PRESETS[preset]["fme_level"])
            got = ic.oracle_pu_search(oracle, frames[f][:w * h], rf[f - 1][:w * h], w, h, pus[sel], prm)
            assert len(ic.me_results_differ(got, res[sel], prm.fme_level)) == 0
            total += len(sel)
    assert total > 2500


@pytest.mark.gpu
@pytest.mark.parametrize("fme_level,constraint", [(2, 0), (2, 1), (0, 0), (0, 1), (4, 0), (4, 1), (1, 0), (3, 1)])
def test_device_search_equals_oracle_on_random_pus(oracle, fme_level, constraint):
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    w, h = 352, 288
    rng = np.random.default_rng(77 + fme_level + constraint)
    frames = ic.clip(w, h, 2, 21, 2.0, (3.25, -1.5))
    cur, ref = frames[1][:w * h], frames[0][:w * h]
    pus = random_pus(rng, w, h, 1500)
    prm = ic.MeParams(lambda_sqrt=ic.lambda_sqrt(27), mv_constraint=constraint, sao=1, deblock=1, fme_level=fme_level)
    want = ic.oracle_pu_search(oracle, cur, ref, w, h, pus, prm)
    got = device_pu_search(lib, dev, cur, ref, w, h, pus, prm)
    bad = ic.me_results_differ(got, want, fme_level)
    assert len(bad) == 0, (len(bad), pus[bad[0]], want[bad[0]], got[bad[0]])
    assert (want["valid"] != 0).sum() > 1000 and (fme_level == 0 or (want["frac_valid"] != 0).sum() > 800)
    assert len({(int(a), int(b)) for a, b in want["mv"]}) > 100   # the searches really go places


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fast-pan-owf-qp37", "ultrafast-fast-pan-owf-qp30", "vertical-pan-owf", "noisy-qp27", "survey-416x240", "baseline-c4-2160p", "faster-owf-qp27", "faster-qp32"])
def test_device_reproduces_every_search_of_an_encode(oracle, name):
    import kvazaar_amd
    from kvazaar_amd.dev import Dev
    lib = kvazaar_amd.load_library()
    dev = Dev(lib)
    case = [c for c in ic.CASES if c[0] == name][0]
    _, w, h, n, qp, preset, dbk, sao, owf, src = case
    frames, rf, qps, pus, res, poc = ic.traced_encode(oracle, case, capacity=1200000)   # BASELINE config 4's own clip: ~40 s of oracle on one core
    total = 0
    for f in range(1, n):
        sel = np.flatnonzero(poc == f)
        if not len(sel):
            continue
        prm = ic.MeParams(lambda_sqrt=ic.lambda_sqrt(int(qps[f])), mv_constraint=int(owf > 0), sao=int(sao), deblock=int(dbk), fme_level=ic.PRESETS[preset]["fme_level"])
        got = device_pu_search(lib, dev, frames[f][:w * h], rf[f - 1][:w * h], w, h, pus[sel], prm)
        bad = ic.me_results_differ(got, res[sel], prm.fme_level)
        assert len(bad) == 0, (f, len(bad), len(sel), pus[sel][bad[0]], res[sel][bad[0]], got[bad[0]])
        total += len(sel)
    assert total > 300
