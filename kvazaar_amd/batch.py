"""ctypes view of include/kvz_hip_batch.h: the batched, device-resident all-intra CTU pass (kvz_hip_intra_frames) and what follows
it per picture (deblocking, picture hashes).  Used by bench.py, __graft_entry__.smoke(), tools/ and the GPU tests; no checker code
in here (the oracle runners are in tests/ctu_common.py)."""
import ctypes as C

import numpy as np

from .capi import i16p, ptr, u8p


class CostModel(C.Structure):
    """kvz_hip_intra_cost_model (include/kvz_hip_types.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("lambda_", C.c_double), ("lambda_sqrt", C.c_double), ("split_flag", (C.c_float * 2) * 3),
                ("part_size", C.c_float * 2), ("intra_mode", C.c_float * 2), ("chroma_mode", C.c_float * 2),
                ("cbf_luma", (C.c_float * 2) * 2), ("cbf_chroma", (C.c_float * 2) * 2), ("coeff_weights", C.c_uint64),
                ("qp", C.c_int32), ("adaptive", C.c_int32), ("coeff_cabac", C.c_int32), ("no_wpp", C.c_int32), ("search_32x32", C.c_int32), ("rdoq", C.c_int32), ("search_nxn", C.c_int32), ("ctx_init", C.c_uint8 * 160),
                ("entropy_fbits", C.c_float * 128)]

    def key(self):
        return bytes(self)


def default_coeff_weights(lib, qp):
    """kvz_fast_coeff_get_weights of kvazaar's built-in table (kvz_hip_default_coeff_weights)"""
    lib.kvz_hip_default_coeff_weights.restype = C.c_uint64
    lib.kvz_hip_default_coeff_weights.argtypes = [C.c_int]
    return int(lib.kvz_hip_default_coeff_weights(qp))


def cost_model(lib, qp, weights=None):
    """kvz_hip_intra_cost_model_init: the model of an I slice at `qp` as `kvazaar --preset ultrafast` sets it up"""
    m = CostModel()
    lib.kvz_hip_intra_cost_model_init.argtypes = [C.c_int, C.c_uint64, C.POINTER(CostModel)]
    lib.kvz_hip_intra_cost_model_init.restype = None
    lib.kvz_hip_intra_cost_model_init(qp, default_coeff_weights(lib, qp) if weights is None else weights, C.byref(m))
    return m


def outputs(width, height):
    """host arrays of one picture's results: reconstruction Y|U|V, coefficients, CU depth / intra mode per 8x8, CTU costs"""
    nctu = ((width + 63) // 64) * ((height + 63) // 64)
    ncu = (width // 8) * (height // 8)
    return dict(rec=np.zeros(width * height * 3 // 2, np.uint8), coeff=np.zeros(nctu * 6144, np.int16),
                depth=np.zeros(ncu, np.uint8), mode=np.zeros(ncu, np.uint8), cost=np.zeros(nctu, np.float64))


class BatchError(RuntimeError):
    """a pass of the batch was invalid (kvz_hip_batch_sync returned -1)"""


class HipBatch:
    """kvz_hip_batch_* through ctypes"""

    def __init__(self, lib, width, height, n_frames, device=-1):
        """device: kvz_hip_batch_create_on -- -1 = the calling thread's current device (the process default unless the thread chose another)"""
        self.lib, self.w, self.h, self.n = lib, width, height, n_frames
        vp, ci = C.c_void_p, C.c_int
        lib.kvz_hip_batch_create.restype = vp
        lib.kvz_hip_batch_create.argtypes = [ci, ci, ci]
        lib.kvz_hip_batch_destroy.argtypes = [vp]
        lib.kvz_hip_batch_destroy.restype = None
        lib.kvz_hip_batch_upload.argtypes = [vp, ci, u8p, u8p, u8p]
        lib.kvz_hip_batch_upload.restype = None
        lib.kvz_hip_batch_download.argtypes = [vp, ci, u8p, u8p, u8p, i16p, u8p, u8p, C.POINTER(C.c_double)]
        lib.kvz_hip_batch_download.restype = ci
        lib.kvz_hip_intra_frames.argtypes = [vp, C.POINTER(CostModel)]
        lib.kvz_hip_intra_frames.restype = ci
        lib.kvz_hip_batch_sync.argtypes = [vp]
        lib.kvz_hip_batch_sync.restype = ci
        lib.kvz_hip_batch_reset.argtypes = [vp]
        lib.kvz_hip_batch_reset.restype = ci
        lib.kvz_hip_batch_last_kernel_ms.argtypes = [vp]
        lib.kvz_hip_batch_last_kernel_ms.restype = C.c_float
        lib.kvz_hip_batch_ctus_per_frame.argtypes = [vp]
        lib.kvz_hip_batch_ctus_per_frame.restype = ci
        lib.kvz_hip_batch_deblock.argtypes = [vp, ci, ci, ci]
        lib.kvz_hip_batch_deblock.restype = None
        lib.kvz_hip_batch_checksums.argtypes = [vp, vp]
        lib.kvz_hip_batch_checksums.restype = ci
        lib.kvz_hip_batch_create_on.restype = vp
        lib.kvz_hip_batch_create_on.argtypes = [ci, ci, ci, ci]
        self.handle = lib.kvz_hip_batch_create_on(device, width, height, n_frames)
        if not self.handle:
            raise RuntimeError(f"kvz_hip_batch_create_on({device}, {width}, {height}, {n_frames}) failed")
        self.ctus_per_frame = lib.kvz_hip_batch_ctus_per_frame(self.handle)

    def upload(self, frame, yuv):
        ys, cs = self.w * self.h, self.w * self.h // 4
        self.lib.kvz_hip_batch_upload(self.handle, frame, ptr(yuv), ptr(yuv, offset=ys), ptr(yuv, offset=ys + cs))

    def upload_all_async(self, src_ptr):
        """kvz_hip_batch_upload_all_async: all n pictures (back to back, Y | U | V each) from host memory at address `src_ptr` -- pinned_bytes() -- on the batch's upload
        queue; starts when the batch's last pass has ended, the next launch waits for it"""
        self.lib.kvz_hip_batch_upload_all_async.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.kvz_hip_batch_upload_all_async.restype = None
        self.lib.kvz_hip_batch_upload_all_async(self.handle, src_ptr)

    def launch(self, model):
        """asynchronous on the batch's stream; returns the number of kernel launches"""
        return self.lib.kvz_hip_intra_frames(self.handle, C.byref(model))

    def order_after(self, other):
        """later work of this batch starts after everything queued on `other` so far (kvz_hip_batch_order_after)"""
        self.lib.kvz_hip_batch_order_after.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.kvz_hip_batch_order_after.restype = None
        self.lib.kvz_hip_batch_order_after(self.handle, other.handle)

    def set_device_share(self, num, den):
        """this batch's persistent pass takes num / den of the device's workgroup slots (kvz_hip_batch_set_device_share): batches of different geometry side by side"""
        self.lib.kvz_hip_batch_set_device_share.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self.lib.kvz_hip_batch_set_device_share.restype = None
        self.lib.kvz_hip_batch_set_device_share(self.handle, num, den)

    def sync(self):
        if self.lib.kvz_hip_batch_sync(self.handle) != 0:
            raise BatchError("kvz_hip_batch_sync: a CTU hand-off wait timed out; results invalid")

    def reset(self):
        """after a BatchError: drains the batch's stream and clears the (sticky) error word -- the batch can run again (kvz_hip_batch_reset)"""
        self.lib.kvz_hip_batch_reset(self.handle)

    def run(self, model):
        n = self.launch(model)
        if n < 0:
            raise BatchError("kvz_hip_intra_frames: the batch cannot run this model (see stderr)")
        self.sync()
        return n

    def deblock(self, qp, beta_offset_div2=0, tc_offset_div2=0, wait=True):
        self.lib.kvz_hip_batch_deblock(self.handle, qp, beta_offset_div2, tc_offset_div2)
        if wait:
            self.sync()

    def loop_filters(self, model, deblock=True, sao=True, beta_offset_div2=0, tc_offset_div2=0, wait=True):
        """kvz_hip_batch_loop_filters: deblocking + SAO decision + SAO reconstruction on the batch's stream"""
        f = self.lib.kvz_hip_batch_loop_filters
        f.argtypes = [C.c_void_p, C.POINTER(CostModel), C.c_int, C.c_int, C.c_int, C.c_int]
        f.restype = None
        f(self.handle, C.byref(model), int(deblock), beta_offset_div2, tc_offset_div2, int(sao))
        if wait:
            self.sync()

    def entropy_code(self, model, sao=False, capacity=None, not_last=None, then=None):
        """kvz_hip_batch_entropy_code: the slice data of every picture of the batch, coded on the device from the results of the last launch (and, with sao, of the
        last loop_filters(sao=True)).  -> (bytes of all substreams back to back, sizes as an array [frame][substream]).  then = (batch, model): that batch's pass is
        started once this coder's first stage is through (kvz_hip_batch_entropy_code_then)"""
        f = self.lib.kvz_hip_batch_entropy_code_then
        f.argtypes = [C.c_void_p, C.POINTER(CostModel), C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = C.c_long
        flags = None if not_last is None else np.ascontiguousarray(not_last, np.uint8)  # tiles: 1 = other tiles of the slice follow
        assert flags is None or flags.size == self.n
        rows = 1 if model.no_wpp else (self.h + 63) // 64
        capacity = capacity or entropy_capacity(self.n, self.w, self.h)
        if getattr(self, "_entropy_out", None) is None or self._entropy_out.nbytes < capacity:
            pinned_free(self.lib, getattr(self, "_entropy_ptr", None))
            self._entropy_ptr, self._entropy_out = pinned_bytes(self.lib, capacity)  # the slice data is downloaded by the call: a pinned destination, no staging copy
        sizes = np.zeros((self.n, rows), np.uint32)
        total = f(self.handle, C.byref(model), int(sao), flags.ctypes.data if flags is not None else None, self._entropy_out.ctypes.data, capacity, sizes.ctypes.data,
                  then[0].handle if then else None, C.addressof(then[1]) if then else None)
        if total < 0:
            # -1 the coder, -2 the next batch's launch (in both cases that pass is queued and wants a sync), -3 bad next model (nothing queued)
            raise BatchError(f"kvz_hip_batch_entropy_code failed ({total})")
        return self._entropy_out[:total], sizes

    def entropy_defer_download(self, on=True):
        """kvz_hip_batch_entropy_defer_download: entropy_code returns with the slice data's download queued; sync() before reading the bytes"""
        self.lib.kvz_hip_batch_entropy_defer_download.argtypes = [C.c_void_p, C.c_int]
        self.lib.kvz_hip_batch_entropy_defer_download.restype = None
        self.lib.kvz_hip_batch_entropy_defer_download(self.handle, int(bool(on)))

    def sao_params(self, frame):
        """-> (luma records, chroma records, merge flags) of one frame, one entry per LCU in raster order"""
        from .capi import SaoParams
        n = self.ctus_per_frame
        luma, chroma, merge = (SaoParams * n)(), (SaoParams * n)(), np.zeros(n, np.uint8)
        f = self.lib.kvz_hip_batch_sao_params
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = C.c_int
        if f(self.handle, frame, luma, chroma, merge.ctypes.data) != 0:
            raise BatchError("kvz_hip_batch_sao_params failed")
        return luma, chroma, merge

    def checksums(self):
        out = np.zeros((self.n, 3), np.uint32)
        if self.lib.kvz_hip_batch_checksums(self.handle, out.ctypes.data) != 0:
            raise BatchError("kvz_hip_batch_checksums: the batch's last pass was invalid")
        return out

    def md5(self):
        """-> uint8 array [frames][3][16]: kvz_image_md5 of every frame's current reconstruction (--hash md5)"""
        out = np.zeros((self.n, 3, 16), np.uint8)
        self.lib.kvz_hip_batch_md5.argtypes = [C.c_void_p, C.c_void_p]
        self.lib.kvz_hip_batch_md5.restype = C.c_int
        if self.lib.kvz_hip_batch_md5(self.handle, out.ctypes.data) != 0:
            raise BatchError("kvz_hip_batch_md5: the batch's last pass was invalid")
        return out

    def kernel_ms(self):
        return self.lib.kvz_hip_batch_last_kernel_ms(self.handle)

    def download(self, frame):
        o = outputs(self.w, self.h)
        ys, cs = self.w * self.h, self.w * self.h // 4
        rc = self.lib.kvz_hip_batch_download(self.handle, frame, ptr(o["rec"]), ptr(o["rec"], offset=ys), ptr(o["rec"], offset=ys + cs),
                                             ptr(o["coeff"]), ptr(o["depth"]), ptr(o["mode"]), o["cost"].ctypes.data_as(C.POINTER(C.c_double)))
        if rc != 0:
            raise BatchError("kvz_hip_batch_download: the batch's last pass was invalid")
        return o

    def download_partitions(self, frame):
        """after a pass with model.search_nxn: (NxN flag per 8x8 CU, luma mode per 4x4 unit)"""
        part, mode4 = np.zeros((self.h // 8) * (self.w // 8), np.uint8), np.zeros((self.h // 4) * (self.w // 4), np.uint8)
        self.lib.kvz_hip_batch_download_partitions.argtypes = [C.c_void_p, C.c_int, u8p, u8p]
        self.lib.kvz_hip_batch_download_partitions.restype = C.c_int
        if self.lib.kvz_hip_batch_download_partitions(self.handle, frame, ptr(part), ptr(mode4)) != 0:
            raise BatchError("kvz_hip_batch_download_partitions: invalid pass, or no pass with search_nxn has run")
        return part, mode4

    def close(self):
        if self.handle:
            self.lib.kvz_hip_batch_destroy(self.handle)
            self.handle = None
        pinned_free(self.lib, getattr(self, "_entropy_ptr", None))
        self._entropy_ptr = self._entropy_out = None


def entropy_capacity(n, w, h):
    """room for the slice data of n pictures: generous for small batches (noise at a low QP codes to more than the pictures' own bytes), half the pictures' bytes for big
    ones (pinned memory; the call fails with a message when a batch outgrows it and the caller can pass its own capacity)"""
    full = n * w * h * 2 + 65536
    return full if full <= (256 << 20) else n * w * h * 3 // 4 + 65536


def pinned_bytes(lib, nbytes):
    """kvz_hip_host_alloc'ed (hipHostMalloc) byte buffer -> (pointer, numpy view)"""
    lib.kvz_hip_host_alloc.restype = C.c_void_p
    lib.kvz_hip_host_alloc.argtypes = [C.c_size_t]
    p = lib.kvz_hip_host_alloc(nbytes)
    if not p:
        raise MemoryError(f"kvz_hip_host_alloc({nbytes})")
    return p, np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p))


def pinned_free(lib, p):
    if p:
        lib.kvz_hip_host_free.argtypes = [C.c_void_p]
        lib.kvz_hip_host_free(p)


class PinnedResults:
    """pinned host buffers for kvz_hip_batch_download_all_async: everything the host entropy coder needs of every frame of a batch"""

    def __init__(self, batch):
        lib, w, h, n = batch.lib, batch.w, batch.h, batch.n
        lib.kvz_hip_host_alloc.restype = C.c_void_p
        lib.kvz_hip_host_alloc.argtypes = [C.c_size_t]
        lib.kvz_hip_host_free.restype = None
        lib.kvz_hip_host_free.argtypes = [C.c_void_p]
        lib.kvz_hip_batch_download_all_async.restype = None
        lib.kvz_hip_batch_download_all_async.argtypes = [C.c_void_p] * 5
        self.lib, self.batch = lib, batch
        self.sizes = dict(rec=w * h * 3 // 2 * n, coeff=batch.ctus_per_frame * 6144 * 2 * n, depth=(w // 8) * (h // 8) * n, mode=(w // 8) * (h // 8) * n)
        self.ptrs = {k: lib.kvz_hip_host_alloc(v) for k, v in self.sizes.items()}
        self.bytes = sum(self.sizes.values())

    def download_async(self):
        self.lib.kvz_hip_batch_download_all_async(self.batch.handle, self.ptrs["rec"], self.ptrs["coeff"], self.ptrs["depth"], self.ptrs["mode"])

    def array(self, name, dtype=np.uint8):
        buf = (C.c_uint8 * self.sizes[name]).from_address(self.ptrs[name])
        return np.frombuffer(buf, dtype=dtype)

    def close(self):
        for p in self.ptrs.values():
            self.lib.kvz_hip_host_free(p)
        self.ptrs = {}
