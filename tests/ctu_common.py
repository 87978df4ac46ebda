"""Shared plumbing of the batched CTU-pass tests: cost model struct, oracle / hostsim / HIP runners."""
import ctypes as C

import numpy as np

import flatapi
from flatapi import ptr


class CostModel(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("lambda_sqrt", C.c_double), ("split_flag", (C.c_float * 2) * 3),
                ("part_size", C.c_float * 2), ("intra_mode", C.c_float * 2), ("chroma_mode", C.c_float * 2),
                ("cbf_luma", (C.c_float * 2) * 2), ("cbf_chroma", (C.c_float * 2) * 2), ("coeff_weights", C.c_uint64),
                ("qp", C.c_int32), ("adaptive", C.c_int32), ("coeff_cabac", C.c_int32), ("no_wpp", C.c_int32), ("ctx_init", C.c_uint8 * 160),
                ("entropy_fbits", C.c_float * 128)]

    def key(self):
        return bytes(self)


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


def outputs(width, height):
    nctu = ((width + 63) // 64) * ((height + 63) // 64)
    ncu = (width // 8) * (height // 8)
    return dict(rec=np.zeros(width * height * 3 // 2, np.uint8), coeff=np.zeros(nctu * 6144, np.int16),
                depth=np.zeros(ncu, np.uint8), mode=np.zeros(ncu, np.uint8), cost=np.zeros(nctu, np.float64))


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


def compare(a, b):
    """names of the outputs that differ (bit-exact comparison, doubles included)"""
    return [k for k in ("rec", "coeff", "depth", "mode", "cost") if a[k].tobytes() != b[k].tobytes()]


def yuv_frames(width, height, n, seed, kind):
    import synth
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


class HipBatch:
    """kvz_hip_batch_* through ctypes (include/kvz_hip_batch.h)"""

    def __init__(self, lib, width, height, n_frames):
        self.lib, self.w, self.h, self.n = lib, width, height, n_frames
        lib.kvz_hip_batch_create.restype = C.c_void_p
        lib.kvz_hip_batch_create.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.kvz_hip_batch_destroy.argtypes = [C.c_void_p]
        lib.kvz_hip_batch_upload.argtypes = [C.c_void_p, C.c_int, flatapi.u8p, flatapi.u8p, flatapi.u8p]
        lib.kvz_hip_batch_download.argtypes = [C.c_void_p, C.c_int, flatapi.u8p, flatapi.u8p, flatapi.u8p, flatapi.i16p, flatapi.u8p,
                                               flatapi.u8p, C.POINTER(C.c_double)]
        lib.kvz_hip_intra_frames.argtypes = [C.c_void_p, C.POINTER(CostModel)]
        lib.kvz_hip_intra_frames.restype = C.c_int
        lib.kvz_hip_batch_sync.argtypes = [C.c_void_p]
        lib.kvz_hip_batch_last_kernel_ms.argtypes = [C.c_void_p]
        lib.kvz_hip_batch_last_kernel_ms.restype = C.c_float
        self.handle = lib.kvz_hip_batch_create(width, height, n_frames)
        assert self.handle

    def upload(self, frame, yuv):
        ys, cs = self.w * self.h, self.w * self.h // 4
        self.lib.kvz_hip_batch_upload(self.handle, frame, ptr(yuv), ptr(yuv, offset=ys), ptr(yuv, offset=ys + cs))

    def run(self, model):
        n = self.lib.kvz_hip_intra_frames(self.handle, C.byref(model))
        self.lib.kvz_hip_batch_sync(self.handle)
        return n

    def deblock(self, qp, beta_offset_div2=0, tc_offset_div2=0):
        self.lib.kvz_hip_batch_deblock.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        self.lib.kvz_hip_batch_deblock.restype = None
        self.lib.kvz_hip_batch_deblock(self.handle, qp, beta_offset_div2, tc_offset_div2)
        self.lib.kvz_hip_batch_sync(self.handle)

    def checksums(self):
        out = np.zeros((self.n, 3), np.uint32)
        self.lib.kvz_hip_batch_checksums.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.kvz_hip_batch_checksums.restype = None
        self.lib.kvz_hip_batch_checksums(self.handle, out.ctypes.data)
        return out

    def kernel_ms(self):
        return self.lib.kvz_hip_batch_last_kernel_ms(self.handle)

    def download(self, frame):
        o = outputs(self.w, self.h)
        ys, cs = self.w * self.h, self.w * self.h // 4
        self.lib.kvz_hip_batch_download(self.handle, frame, ptr(o["rec"]), ptr(o["rec"], offset=ys), ptr(o["rec"], offset=ys + cs),
                                        ptr(o["coeff"]), ptr(o["depth"]), ptr(o["mode"]), o["cost"].ctypes.data_as(C.POINTER(C.c_double)))
        return o

    def close(self):
        if self.handle:
            self.lib.kvz_hip_batch_destroy(self.handle)
            self.handle = None


def hip_cost_model(lib, qp, weights=COEFF_WEIGHTS_QP22):
    m = CostModel()
    lib.kvz_hip_intra_cost_model_init.argtypes = [C.c_int, C.c_uint64, C.POINTER(CostModel)]
    lib.kvz_hip_intra_cost_model_init(qp, weights, C.byref(m))
    return m
