"""Shared plumbing of the batched CTU-pass tests: cost model struct, oracle / hostsim / HIP runners."""
import ctypes as C

import numpy as np

import flatapi
from flatapi import ptr


from kvazaar_amd.batch import CostModel, HipBatch, outputs  # noqa: F401,E402  (the product's ctypes view of the batch API)
from kvazaar_amd.batch import cost_model as _cost_model  # noqa: E402

# kvz_fast_coeff_get_weights(state) for QP 22 of the reference's default table (fast_coeff_cost.h:48-...), as packed by
# to_4xq88 (fast_coeff_cost.c:39-52); tests/test_ctu_pipeline.py checks it against the reference build.
COEFF_WEIGHTS_QP22 = 0x065403F0052C0004


def model_constants():
    """tests/golden/model_constants.json: the entropy table and the per-QP fast-coefficient-cost weights of the reference build"""
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_constants.json")))


def coeff_weights(qp):
    return int(model_constants()["coeff_weights"][str(qp)])


def run_oracle(oracle, model, width, height, yuv):
    o = outputs(width, height)
    ys, cs = width * height, width * height // 4
    y, u, v = yuv[:ys], yuv[ys:ys + cs], yuv[ys + cs:]
    f = oracle.lib.kvz_oracle_intra_frame
    f.restype = None
    f(C.byref(model), width, height, ptr(y), ptr(u), ptr(v), ptr(o["rec"]), ptr(o["rec"], offset=ys), ptr(o["rec"], offset=ys + cs),
      ptr(o["coeff"]), ptr(o["depth"]), ptr(o["mode"]), o["cost"].ctypes.data_as(C.POINTER(C.c_double)))
    return o


def run_hostsim(lib, model, width, height, yuv):
    o = outputs(width, height)
    f = lib.kvz_hostsim_intra_frame
    f.restype = None
    f(C.byref(model), width, height, ptr(yuv), ptr(o["rec"]), ptr(o["coeff"]), ptr(o["depth"]), ptr(o["mode"]),
      o["cost"].ctypes.data_as(C.POINTER(C.c_double)))
    return o


def _partition_outputs(o, width, height):
    o["part"], o["mode4"] = np.zeros((height // 8) * (width // 8), np.uint8), np.zeros((height // 4) * (width // 4), np.uint8)
    return o


def run_oracle_nxn(oracle, model, width, height, yuv):
    """kvz_oracle_intra_frame_nxn: the pass + the NxN flag per 8x8 CU ("part") and the luma mode per 4x4 unit ("mode4")"""
    o = _partition_outputs(outputs(width, height), width, height)
    ys, cs = width * height, width * height // 4
    f = oracle.lib.kvz_oracle_intra_frame_nxn
    f.restype = None
    f(C.byref(model), width, height, ptr(yuv[:ys]), ptr(yuv[ys:ys + cs]), ptr(yuv[ys + cs:]), ptr(o["rec"]), ptr(o["rec"], offset=ys), ptr(o["rec"], offset=ys + cs),
      ptr(o["coeff"]), ptr(o["depth"]), ptr(o["mode"]), o["cost"].ctypes.data_as(C.POINTER(C.c_double)), ptr(o["part"]), ptr(o["mode4"]))
    return o


def run_hostsim_nxn(lib, model, width, height, yuv):
    o = _partition_outputs(outputs(width, height), width, height)
    f = lib.kvz_hostsim_intra_frame_nxn
    f.restype = None
    f(C.byref(model), width, height, ptr(yuv), ptr(o["rec"]), ptr(o["coeff"]), ptr(o["depth"]), ptr(o["mode"]),
      o["cost"].ctypes.data_as(C.POINTER(C.c_double)), ptr(o["part"]), ptr(o["mode4"]))
    return o


def compare(a, b):
    """names of the outputs that differ (bit-exact comparison, doubles included)"""
    return [k for k in ("rec", "coeff", "depth", "mode", "cost", "part", "mode4") if k in a and k in b and a[k].tobytes() != b[k].tobytes()]


def yuv_frames(width, height, n, seed, kind):
    if kind == "adversarial":  # flat / noise / ramp / blocks (seed unused): the pictures that make band SAO, zero-coefficient CUs, big merges happen
        return list(adversarial_frames(width, height).values())[:n]
    from kvazaar_amd import synth
    return [np.concatenate([p.reshape(-1) for p in planes]) for planes in synth.frames(width, height, n, seed, kind)]


def adversarial_frames(width, height):
    """flat, extreme and pure-noise pictures: exercise the no-coefficient / early-termination / merge paths"""
    n = width * height * 3 // 2
    rng = np.random.default_rng(9)
    flat = np.full(n, 128, np.uint8)
    noise = rng.integers(0, 256, n, dtype=np.uint8)
    ramp = np.concatenate([(np.add.outer(np.arange(height), np.arange(width)) % 256).astype(np.uint8).reshape(-1),
                           np.full(n - width * height, 90, np.uint8)])
    blocks = np.concatenate([(((np.add.outer(np.arange(height) // 16, np.arange(width) // 16)) % 2) * 200 + 20).astype(np.uint8).reshape(-1),
                             rng.integers(100, 140, n - width * height, dtype=np.uint8)])
    return {"flat": flat, "noise": noise, "ramp": ramp, "blocks": blocks}


def hip_cost_model(lib, qp, weights=COEFF_WEIGHTS_QP22):
    return _cost_model(lib, qp, weights)
