"""The flat strategy API three times over, for the parity tests:
  kvz_hip_*     product: libkvz_hip.so (include/kvz_hip.h), HIP kernels on the GPU -- binding table in kvazaar_amd/capi.py
  kvz_oracle_*  test oracle: oracle/libkvz_oracle.so (our C restatement)
  kvz_ref_*     reference build: oracle/_ref/libkvz_refshim.so (wrappers around kvazaar's own pointers)
This module adds the loaders of the two checkers to the product's binding table."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # the repository root, for scripts run from tests/
from kvazaar_amd.capi import (A, EpolParams, FlatLib, IPOL_COL_LEN, IPOL_IM_PLANE, QuantParams, ROOT, SIGNATURES, SaoParams,  # noqa: F401
                              i8p, i16p, i32p, ptr, u8p, u16p, u32p)


def oracle_path():
    return os.path.join(ROOT, "oracle", "libkvz_oracle.so")


def refshim_path():
    return os.path.join(ROOT, "oracle", "_ref", "libkvz_refshim.so")


def hip_path():
    return os.path.join(ROOT, "kvazaar_amd", "lib", "libkvz_hip.so")


def load_oracle():
    lib = FlatLib(oracle_path(), "kvz_oracle_")
    lib.lib.kvz_oracle_dct_matrix.restype = i16p
    lib.lib.kvz_oracle_dct_matrix.argtypes = [C.c_int]
    lib.lib.kvz_oracle_dst_matrix.restype = i16p
    lib.lib.kvz_oracle_scan_table.restype = u32p
    lib.lib.kvz_oracle_scan_table.argtypes = [C.c_int, C.c_int]
    return lib


_ref_cache = {}


def load_ref(cpuid):
    """Reference build with generic (cpuid=0) or best-available/AVX2 (cpuid=1) strategies bound.

    The function pointers are process globals (kvazaar.c:94), so selecting rebinds them for every holder."""
    if "lib" not in _ref_cache:
        lib = FlatLib(refshim_path(), "kvz_ref_", optional=("coeff_nxn_bins",))  # the reference has no record form: kvz_ref_encode_coeff_nxn_bytes checks the coded bytes
        lib.lib.kvz_ref_select.argtypes = [C.c_int]
        lib.lib.kvz_ref_dct_matrix.restype = i16p
        lib.lib.kvz_ref_dct_matrix.argtypes = [C.c_int]
        lib.lib.kvz_ref_dst_matrix.restype = i16p
        lib.lib.kvz_ref_scan_table.restype = u32p
        lib.lib.kvz_ref_scan_table.argtypes = [C.c_int, C.c_int]
        lib.lib.kvz_ref_fast_coeff_weights.restype = C.c_uint64
        lib.lib.kvz_ref_fast_coeff_weights.argtypes = [C.c_int]
        lib.lib.kvz_ref_quant_coeff.restype = i16p
        lib.lib.kvz_ref_quant_coeff.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.lib.kvz_ref_dequant_coeff.restype = i16p
        lib.lib.kvz_ref_dequant_coeff.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.lib.kvz_ref_set_scaling_list.restype = None
        lib.lib.kvz_ref_set_scaling_list.argtypes = [C.c_int, i16p, i32p]

        def set_scaling_list(lists):
            """the encoder control the reference's wrappers run on: a tests/scaling_lists.py ListSet, or None = --scaling-list off"""
            if lists is None:
                lib.lib.kvz_ref_set_scaling_list(0, None, None)
            else:
                mode, coeff, dc = lists.ref_args()
                lib.lib.kvz_ref_set_scaling_list(mode, ptr(coeff), ptr(dc))
        lib.set_scaling_list = set_scaling_list
        lib.lib.kvz_ref_entropy_fbits.restype = C.c_float
        lib.lib.kvz_ref_entropy_fbits.argtypes = [C.c_int]
        lib.lib.kvz_ref_optimized_sad.restype = C.c_uint32
        lib.lib.kvz_ref_optimized_sad.argtypes = [C.c_int, u8p, u8p, C.c_int32, C.c_uint32, C.c_uint32]
        _ref_cache["lib"] = lib
    lib = _ref_cache["lib"]
    assert lib.lib.kvz_ref_select(cpuid) == 1
    return lib
