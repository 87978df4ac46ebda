"""ctypes binding table of the flat strategy API of libkvz_hip.so (include/kvz_hip.h group 2: the reference's strategy
functions with the host structs replaced by the PODs of include/kvz_hip_types.h).

FlatLib binds <prefix><name> for every name in SIGNATURES; the product prefix is kvz_hip_ (`hip_api()`).  The test tree
binds its checkers (oracle, compiled reference) through the same table with their own prefixes (tests/flatapi.py), so that a
parity test is "call two libraries with identical arguments, demand identical bytes".
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repository root

u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
i16p = C.POINTER(C.c_int16)
u16p = C.POINTER(C.c_uint16)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)


class QuantParams(C.Structure):
    _fields_ = [("qp", C.c_int32), ("bitdepth", C.c_int32), ("slice_is_intra", C.c_int32), ("signhide", C.c_int32),
                ("scaling_list", C.c_int32), ("cu_is_intra", C.c_int32), ("quant_coeff", i16p), ("dequant_coeff", i16p)]


class SaoParams(C.Structure):
    _fields_ = [("type", C.c_int32), ("eo_class", C.c_int32), ("band_position", C.c_int32 * 2),
                ("offsets", C.c_int32 * 10), ("bitdepth", C.c_int32)]


class EpolParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("src_w", "src_h", "src_s", "blk_x", "blk_y", "blk_w", "blk_h",
                                         "pad_l", "pad_r", "pad_t", "pad_b", "pad_b_simd")]


_ipol_blocks = (None, [u8p, C.c_int16, C.c_int, C.c_int, u8p, i16p, C.c_int8, i16p, C.c_int8, C.c_int8])
_sample8 = (None, [u8p, C.c_int16, C.c_int, C.c_int, u8p, C.c_int16, C.c_int8, C.c_int8, i16p])
_sample16 = (None, [u8p, C.c_int16, C.c_int, C.c_int, i16p, C.c_int16, C.c_int8, C.c_int8, i16p])

SIGNATURES = {
    "reg_sad": (C.c_uint, [u8p, u8p, C.c_int, C.c_int, C.c_uint, C.c_uint]),
    "sad_nxn": (C.c_uint, [C.c_int, u8p, u8p]),
    "satd_nxn": (C.c_uint, [C.c_int, u8p, u8p]),
    "sad_nxn_dual": (None, [C.c_int, u8p, u8p, C.c_uint, u32p]),
    "satd_nxn_dual": (None, [C.c_int, u8p, u8p, C.c_uint, u32p]),
    "satd_any_size": (C.c_uint, [C.c_int, C.c_int, u8p, C.c_int, u8p, C.c_int]),
    "satd_any_size_quad": (None, [C.c_int, C.c_int, C.POINTER(u8p), C.c_int, u8p, C.c_int, C.c_uint, u32p, i8p]),
    "pixels_calc_ssd": (C.c_uint, [u8p, u8p, C.c_int, C.c_int, C.c_int]),
    "ver_sad": (C.c_uint32, [u8p, u8p, C.c_int32, C.c_int32, C.c_uint32]),
    "hor_sad": (C.c_uint32, [u8p, u8p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "pixel_var": (C.c_double, [u8p, C.c_uint32]),
    "bipred_average_plane": (None, [u8p, C.c_uint, u8p, i16p, u8p, i16p, C.c_uint, C.c_uint]),
    "image_calc_sad": (C.c_uint, [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "transform": (None, [C.c_int, C.c_int8, i16p, i16p]),
    "quant": (None, [C.POINTER(QuantParams), i16p, i16p, C.c_int32, C.c_int32, C.c_int8, C.c_int8, C.c_int8]),
    "dequant": (None, [C.POINTER(QuantParams), i16p, i16p, C.c_int32, C.c_int32, C.c_int8, C.c_int8]),
    "quantize_residual": (C.c_int, [C.POINTER(QuantParams), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    u8p, u8p, u8p, i16p, C.c_int]),
    "plane_checksum": (C.c_uint32, [u8p, C.c_int, C.c_int, C.c_int]),
    "coeff_nxn_bins": (C.c_int, [i16p, C.c_int, C.c_int, C.c_int, u32p, C.c_int]),
    "plane_md5": (None, [u8p, C.c_int, C.c_int, C.c_int, u8p]),
    "coeff_abs_sum": (C.c_uint32, [i16p, C.c_size_t]),
    "fast_coeff_cost": (C.c_double, [i16p, C.c_int32, C.c_uint64]),
    "find_last_scanpos": (None, [i16p, i16p, C.c_int8, C.c_int32, i16p, i32p, C.c_uint32, u16p, u32p, i32p, i32p,
                                 C.c_uint32, i32p, C.c_int32, C.c_int8]),
    "get_scaled_qp": (C.c_int32, [C.c_int8, C.c_int8, C.c_int8]),
    "angular_pred": (None, [C.c_int, C.c_int, u8p, u8p, u8p]),
    "intra_pred_planar": (None, [C.c_int, u8p, u8p, u8p]),
    "intra_pred_filtered_dc": (None, [C.c_int, u8p, u8p, u8p]),
    "sample_quarterpel_luma": _sample8,
    "sample_quarterpel_luma_hi": _sample16,
    "sample_octpel_chroma": _sample8,
    "sample_octpel_chroma_hi": _sample16,
    "filter_hpel_blocks_hor_ver_luma": _ipol_blocks,
    "filter_hpel_blocks_diag_luma": _ipol_blocks,
    "filter_qpel_blocks_hor_ver_luma": _ipol_blocks,
    "filter_qpel_blocks_diag_luma": _ipol_blocks,
    "get_extended_block": (C.c_int, [C.POINTER(EpolParams), u8p, u8p]),
    "sao_edge_ddistortion": (C.c_int, [C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]),
    "calc_sao_edge_dir": (None, [C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]),
    "sao_reconstruct_color": (None, [C.POINTER(SaoParams), u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "sao_band_ddistortion": (C.c_int, [C.c_int, u8p, u8p, C.c_int, C.c_int, C.c_int, i32p]),
}

IPOL_IM_PLANE = (64 + 7 + 1) * 64 + 1
IPOL_COL_LEN = 64 + 7 + 1


class FlatLib:
    """A loaded library exposing <prefix><name> for every name in SIGNATURES."""

    def __init__(self, path, prefix, optional=()):
        self.path, self.prefix = path, prefix
        self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            try:
                f = getattr(self.lib, prefix + name)
            except AttributeError:
                if name in optional:
                    continue
                raise
            f.restype, f.argtypes = res, args
            setattr(self, name, f)


def A(a, align=64):
    """64-byte aligned copy.  The reference's AVX2 kernels use aligned loads on block buffers, exactly as its
    callers allocate them (ALIGNED(32)/ALIGNED(64) stack arrays, e.g. search_intra.c:409-411, quant-generic.c:207)."""
    a = np.ascontiguousarray(a)
    buf = np.empty(a.nbytes + align, np.uint8)
    off = (-buf.ctypes.data) % align
    out = buf[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def ptr(a, ty=None, offset=0):
    """ctypes pointer into a numpy array (optionally at an element offset)."""
    assert a.flags["C_CONTIGUOUS"]
    ty = ty or {np.dtype("uint8"): u8p, np.dtype("int8"): i8p, np.dtype("int16"): i16p, np.dtype("uint16"): u16p,
                np.dtype("int32"): i32p, np.dtype("uint32"): u32p}[a.dtype]
    return C.cast(a.ctypes.data + offset * a.itemsize, ty)


def hip_api():
    """the product library behind the flat API (raises when it is not built: there is no CPU path)"""
    from .library import load_library
    from .build import LIB_PATH
    load_library()
    return FlatLib(LIB_PATH, "kvz_hip_")
