"""Host side of the inter CTU pass (include/kvz_hip_dev.h kvz_hip_dev_inter_ctu_pass): the CU record layout, the per-picture parameters and a thin wrapper that
keeps the pictures of many independent sequences on the device.  The product path: no oracle, no CPU fallback -- the library call fails if the HIP kernels are missing."""
import ctypes as C
import os
import hashlib

import numpy as np

CU_DTYPE = np.dtype([("type", "u1"), ("depth", "u1"), ("mode", "u1"), ("tr_depth", "u1"), ("cbf", "<u2"), ("skipped", "u1"), ("merged", "u1"), ("merge_idx", "u1"),
                     ("mv_dir", "u1"), ("mv_ref", "u1", (2,)), ("mv_cand", "u1", (2,)), ("mv", "<i2", (2, 2))], align=True)  # kvz_hip_cu_info
assert CU_DTYPE.itemsize == 22


class InterParams(C.Structure):  # kvz_hip_inter_params
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in ("qp", "poc", "mv_constraint", "sao", "deblock", "fme_level", "pu_depth_inter_max", "no_wpp", "fast_residual_cost",
                                                "ref_width", "ref_height", "tile_x", "tile_y", "no_tmvp")]  # the last four: tiles (include/kvz_hip_dev.h), zero = the picture is the frame


    def __init__(self, **kw):
        super().__init__(**kw)
        if "struct_size" not in kw:
            self.struct_size = C.sizeof(InterParams)  # the version of the struct this binding was written against (include/kvz_hip_dev.h)


def veryfast_params(qp, poc, mv_constraint=True):
    """`--preset veryfast` (cfg.c:541-568) for a B picture of the low-delay GOP"""
    p = InterParams(qp=qp, poc=poc, mv_constraint=int(mv_constraint), sao=1, deblock=1, fme_level=2, pu_depth_inter_max=3, no_wpp=0, fast_residual_cost=28)
    import os
    for k in os.environ.get("KVZ_DEBUG_INTER_PARAMS", "").split(","):  # developer: timing experiments (e.g. no_tmvp=1,mv_constraint=0); the results no longer verify
        if "=" in k:
            setattr(p, k.split("=")[0], int(k.split("=")[1]))
    return p


def lowdelay_picture_qp(qp, frame, gop_len=4, gop_depth=3, intra_period=0, preset_given=True):
    """the picture QP kvazaar runs picture `frame` of `--gop lp-g<len>d<depth>t1 -q <qp>` at, without rate control: intra_qp_offset for the I picture
    (encoder.c:180-183), the GOP layer for the others (cfg.c:1455-1463) plus -- when a preset was given, whose "gop 8" leaves the random-access GOP's QP model in
    the entries kvz_config_process_lp_gop does not rewrite (cfg.c:485-760, gop.h:94-200) -- CLIP(0, 3, qp * scale + offset) of rate_control.c:1040-1056"""
    ra8_offset = (0.0, -6.25, -6.25, -7.0, -7.0, -6.25, -7.0, -7.0)
    ra8_scale = (0.0, 0.25, 0.25, 0.245, 0.245, 0.25, 0.245, 0.245)
    pos = frame % intra_period if intra_period else frame
    if pos == 0:
        l2 = 0
        while (1 << l2) < gop_len:
            l2 += 1
        q = qp + (max(-l2 + 1, -3) if gop_len > 1 else 0)
    else:
        k = (pos + gop_len - 1) % gop_len
        g = k + 1
        modulo = [0] * 8
        for d in range(gop_depth):
            modulo[gop_depth - 1 - d] = 1 << d
        modulo[0] = gop_len
        layer = 1
        while layer < gop_depth and g % modulo[layer - 1]:
            layer += 1
        dq = float(qp + layer)
        if preset_given and k < 8:
            dq += min(3.0, max(0.0, dq * ra8_scale[k] + ra8_offset[k]))
        q = int(dq + 0.5)
    return min(51, max(0, q))


def intra_picture_cu_info(width, height):
    """the CU info of an I picture as far as the next picture's search reads it (no motion: no temporal candidates, no co-located starting point)"""
    cu = np.zeros((height // 4, width // 4), CU_DTYPE)
    cu["type"] = 1
    cu["mv_ref"] = 255
    return cu


def cu_decision_bytes(cu):
    """the decisions of one picture as bytes, motion and flags only where they mean something (the digests of tests/golden/inter_recon.json are taken over these)"""
    inter = cu["type"] == 2
    coded = inter & (cu["merged"] == 0) & (cu["skipped"] == 0)
    parts = [cu["type"], cu["depth"], np.where(cu["type"] == 1, cu["mode"], 0), np.where(inter, cu["skipped"], 0), np.where(inter, cu["merged"], 0),
             np.where(inter & ((cu["merged"] | cu["skipped"]) > 0), cu["merge_idx"], 0), np.where(inter, cu["mv_dir"], 0)]
    for l in range(2):
        used = inter & ((cu["mv_dir"] >> l) & 1 > 0)
        parts += [np.where(used, cu["mv"][..., l, 0], 0).astype("<i2"), np.where(used, cu["mv"][..., l, 1], 0).astype("<i2"), np.where(coded & used, cu["mv_cand"][..., l], 0)]
    return b"".join(np.ascontiguousarray(p).tobytes() for p in parts)


def cu_digest(cu):
    return hashlib.sha256(cu_decision_bytes(cu)).hexdigest()[:24]


class InterPictures:
    """picture k of `n` independent sequences, resident on the device: sources, references (+ their CU info), and the pass's outputs"""

    def __init__(self, lib, width, height, n, with_levels=False):
        from .dev import Dev
        self.lib, self.dev, self.w, self.h, self.n = lib, Dev(lib), width, height, n
        self.fs, self.cells = width * height * 3 // 2, (width // 4) * (height // 4)
        lib.kvz_hip_dev_inter_ctu_pass.restype = C.c_int
        lib.kvz_hip_dev_inter_ctu_pass.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]
        lib.kvz_hip_dev_cu_dbk_from_info.restype = None
        lib.kvz_hip_dev_cu_dbk_from_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.kvz_hip_dev_loop_filters_inter.restype = C.c_int
        lib.kvz_hip_dev_loop_filters_inter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 3
        self.d_dbk = None
        lib.kvz_hip_dev_entropy_code_inter.restype = C.c_long
        lib.kvz_hip_dev_entropy_code_inter.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        e = self.dev.empty
        self.d_src, self.d_ref, self.d_rec = e(n * self.fs), e(n * self.fs), e(n * self.fs)
        self.d_ref_cu, self.d_cu = e(n * self.cells * CU_DTYPE.itemsize), e(n * self.cells * CU_DTYPE.itemsize)
        self.ctus = ((width + 63) // 64) * ((height + 63) // 64)
        self.d_coeff = e(n * self.ctus * 6144 * 2) if with_levels else None  # KVZ_HIP_CTU_COEFFS int16 per CTU: what the entropy coder reads
        self._entropy_out = None

    def upload(self, i, src, ref, ref_cu):
        up = self.lib.kvz_hip_dev_upload
        for base, a, size in ((self.d_src, src, self.fs), (self.d_ref, ref, self.fs), (self.d_ref_cu, ref_cu, self.cells * CU_DTYPE.itemsize)):
            a = np.ascontiguousarray(a)
            assert a.nbytes == size
            up(base + i * size, a.ctypes.data, size)

    def run(self, params):
        rc = self.lib.kvz_hip_dev_inter_ctu_pass(self.d_src, self.d_ref, self.d_ref_cu, self.d_rec, self.d_cu, self.d_coeff, self.w, self.h, self.n, C.addressof(params))
        if rc != 0:
            raise RuntimeError(f"kvz_hip_dev_inter_ctu_pass returned {rc}")

    def upload_source(self, i, src):
        a = np.ascontiguousarray(src)
        assert a.nbytes == self.fs
        self.lib.kvz_hip_dev_upload(self.d_src + i * self.fs, a.ctypes.data, self.fs)

    def new_source_set(self):
        """another resident set of n source pictures (the next picture of every sequence); returns the handle use_source_set() takes"""
        self._source_sets = getattr(self, "_source_sets", [self.d_src])
        self._source_sets.append(self.dev.empty(self.n * self.fs))
        return len(self._source_sets) - 1

    def use_source_set(self, k):
        self._source_sets = getattr(self, "_source_sets", [self.d_src])
        self.d_src = self._source_sets[k]

    def loop_filters(self, params, slice_is_b=True):
        """deblocking and SAO of the pictures the pass just produced, in place (kvz_hip_dev_loop_filters_inter): d_rec becomes what the next picture predicts from"""
        if self.d_dbk is None:
            self.d_dbk = self.dev.empty(self.n * self.cells * 20)  # kvz_hip_cu_dbk
        self.lib.kvz_hip_dev_cu_dbk_from_info(self.d_cu, self.n * self.cells, self.d_dbk)
        rc = self.lib.kvz_hip_dev_loop_filters_inter(self.d_src, self.d_rec, self.w, self.h, self.n, self.d_dbk, params.qp, int(slice_is_b), params.deblock, 0, 0, params.sao,
                                                     params.no_wpp, None, None, None)
        if rc != 0:
            raise RuntimeError(f"kvz_hip_dev_loop_filters_inter returned {rc}")

    def entropy_code(self, params):
        """kvz_hip_dev_entropy_code_inter: the slice data of the pictures the pass just produced (after loop_filters when params.sao: their SAO decisions are coded).
        -> (bytes of all substreams back to back, sizes [sequence][substream])"""
        if self.d_coeff is None:
            raise RuntimeError("InterPictures(..., with_levels=True) keeps the levels the entropy coder needs")
        rows = 1 if params.no_wpp else (self.h + 63) // 64
        from .batch import entropy_capacity
        capacity = entropy_capacity(self.n, self.w, self.h)
        if self._entropy_out is None:
            from .batch import pinned_bytes
            self._entropy_ptr, self._entropy_out = pinned_bytes(self.lib, capacity)  # pinned: the call downloads the slice data straight into it
        sizes = np.zeros((self.n, rows), np.uint32)
        total = self.lib.kvz_hip_dev_entropy_code_inter(self.d_cu, self.d_ref_cu, self.d_coeff, self.w, self.h, self.n, C.addressof(params), self._entropy_out.ctypes.data,
                                                        capacity, sizes.ctypes.data)
        if total < 0:
            raise RuntimeError("kvz_hip_dev_entropy_code_inter failed")
        return self._entropy_out[:total], sizes

    def advance(self):
        """the pictures just encoded (after their loop filters) and their CU records become the references of the next picture"""
        self.d_ref, self.d_rec = self.d_rec, self.d_ref
        self.d_ref_cu, self.d_cu = self.d_cu, self.d_ref_cu

    def sync(self):
        self.lib.kvz_hip_dev_sync()

    def download(self, i):
        rec, cu = np.empty(self.fs, np.uint8), np.empty((self.h // 4, self.w // 4), CU_DTYPE)
        self.lib.kvz_hip_dev_download(rec.ctypes.data, self.d_rec + i * self.fs, self.fs)
        self.lib.kvz_hip_dev_download(cu.ctypes.data, self.d_cu + i * self.cells * CU_DTYPE.itemsize, cu.nbytes)
        return rec, cu

    def close(self):
        self.dev.free(self.d_ref, self.d_rec, self.d_ref_cu, self.d_cu, *getattr(self, "_source_sets", [self.d_src]))
        if self.d_dbk is not None:
            self.dev.free(self.d_dbk)
        if self.d_coeff is not None:
            self.dev.free(self.d_coeff)
        if getattr(self, "_entropy_ptr", None):
            from .batch import pinned_free
            pinned_free(self.lib, self._entropy_ptr)
            self._entropy_ptr = self._entropy_out = None


class TiledInterSequences:
    """BASELINE config 4 sharded by tile: picture k of `n` independent sequences of a width x height frame cut into kvazaar's uniform --tiles grid.  This rank's
    tiles are resident as pictures of their own (source, reconstruction, CU records: the tile is an independent sub-picture for prediction, neighbours, contexts and loop
    filters); the REFERENCE frames and their CU records are whole frames on every rank -- motion vectors leave the tile --, rebuilt after every picture by THE collective of the
    configuration: one all-gather of the ranks' filtered tiles (RCCL over xGMI when world > 1) and a paste, and the same for the CU records.
    Buffers are torch tensors (the collective is torch.distributed's); the passes run through the C ABI on the library's stream of this thread."""

    def __init__(self, lib, width, height, cols, rows, n, rank=0, world=1, dist=None, device="cuda"):
        import torch
        from . import sharding
        self.lib, self.w, self.h, self.n, self.rank, self.world, self.dist, self.torch = lib, width, height, n, rank, world, dist, torch
        self.plan = sharding.exchange_plan(width, height, cols, rows, world)
        self.tiles = self.plan["tiles"]
        self.mine = sharding.tiles_of_rank(len(self.tiles), rank, world)
        self.per_rank = self.plan["slots_per_rank"]
        self.fs, self.cells = width * height * 3 // 2, (width // 4) * (height // 4)
        self.rec_item = CU_DTYPE.itemsize
        dev = torch.device(device)
        e = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self.ref, self.ref_cu = e(n * self.fs), e(n * self.cells * self.rec_item)
        self.slot_px = max(t[2] * t[3] * 3 // 2 for t in self.tiles)
        self.slot_cu = max((t[2] // 4) * (t[3] // 4) for t in self.tiles) * self.rec_item
        # send / receive buffers of the two all-gathers: [slot of the rank][sequence][bytes of the largest tile]
        self.send_px, self.send_cu = torch.zeros(self.per_rank * n * self.slot_px, dtype=torch.uint8, device=dev), torch.zeros(self.per_rank * n * self.slot_cu, dtype=torch.uint8, device=dev)
        self.recv_px = e(world * self.per_rank * n * self.slot_px) if world > 1 else self.send_px
        self.recv_cu = e(world * self.per_rank * n * self.slot_cu) if world > 1 else self.send_cu
        # this rank's tiles grouped by size: one launch of the pass (and of the loop filters) per group, picture j * n + i = tile j of the group, sequence i
        self.groups = {}
        for ti in self.mine:
            self.groups.setdefault((self.tiles[ti][2], self.tiles[ti][3]), []).append(ti)
        self.src, self.rec, self.cu, self.dbk, self.xy = {}, {}, {}, {}, {}
        for (tw, th), members in self.groups.items():
            m = len(members) * n
            self.src[(tw, th)], self.rec[(tw, th)] = e(m * tw * th * 3 // 2), e(m * tw * th * 3 // 2)
            self.cu[(tw, th)], self.dbk[(tw, th)] = e(m * (tw // 4) * (th // 4) * self.rec_item), e(m * (tw // 4) * (th // 4) * 20)
            xy = np.array([[self.tiles[ti][0], self.tiles[ti][1]] for ti in members for _ in range(n)], np.int32)
            self.xy[(tw, th)] = torch.from_numpy(xy).to(dev)
        lib.kvz_hip_dev_inter_ctu_pass_tiles.restype = C.c_int
        lib.kvz_hip_dev_inter_ctu_pass_tiles.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p, C.c_void_p, C.c_int]
        lib.kvz_hip_dev_cu_dbk_from_info.restype = None
        lib.kvz_hip_dev_cu_dbk_from_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.kvz_hip_dev_loop_filters_inter.restype = C.c_int
        lib.kvz_hip_dev_loop_filters_inter.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 7 + [C.c_void_p] * 3
        self.ctus = sum(((self.tiles[ti][2] + 63) // 64) * ((self.tiles[ti][3] + 63) // 64) for ti in self.mine)  # of this rank, per sequence
        self.pass_ms = 0.0

    def group_params(self, base):
        return InterParams(qp=base.qp, poc=base.poc, mv_constraint=0, sao=base.sao, deblock=base.deblock, fme_level=base.fme_level, pu_depth_inter_max=base.pu_depth_inter_max,
                           no_wpp=1, fast_residual_cost=base.fast_residual_cost, ref_width=self.w, ref_height=self.h, tile_x=0, tile_y=0, no_tmvp=1)  # tiles: --no-wpp, no TMVP (cfg.c:920-975)

    def upload_sources(self, frames_of_sequence):
        """frames_of_sequence(i) -> the planar frame of sequence i for the picture about to be encoded; every tile of this rank gets its part"""
        from . import sharding
        torch = self.torch
        for (tw, th), members in self.groups.items():
            ts = tw * th * 3 // 2
            host = np.empty((len(members), self.n, ts), np.uint8)
            for j, ti in enumerate(members):
                cache = {}
                for i in range(self.n):
                    f = frames_of_sequence(i)
                    k = id(f)
                    if k not in cache:
                        cache[k] = sharding.crop_tile(f, self.w, self.h, self.tiles[ti])
                    host[j, i] = cache[k]
            self.src[(tw, th)].copy_(torch.from_numpy(host).reshape(-1))

    def set_reference(self, frames, cu):
        """whole reference frames [n, fs] and their CU records [n, h/4, w/4] (host arrays): the start of a chain"""
        torch = self.torch
        self.ref.copy_(torch.from_numpy(np.ascontiguousarray(frames).reshape(-1)))
        self.ref_cu.copy_(torch.from_numpy(np.ascontiguousarray(cu).view(np.uint8).reshape(-1)))

    def save_reference(self):
        """keep a device copy of the reference frames and their CU records as they are now (restore_reference puts them back: a bench step that codes the same picture again)"""
        self._saved = (self.ref.clone(), self.ref_cu.clone())

    def restore_reference(self):
        self.ref.copy_(self._saved[0])
        self.ref_cu.copy_(self._saved[1])

    def run_picture(self, base):
        """one B picture of every sequence: the CTU pass of this rank's tiles, their loop filters, then the exchange that turns the result into the next picture's reference"""
        torch = self.torch
        torch.cuda.synchronize()
        self.lib.kvz_hip_dev_inter_kernel_ms.restype = C.c_float
        prm = self.group_params(base)

        def group_pass(g, members, parts=1):
            tw, th = g
            m = len(members) * self.n
            self.lib.kvz_hip_dev_inter_set_share(parts)  # per calling thread: this pass's share of the workgroup slots (the pool's threads stay alive)
            rc = self.lib.kvz_hip_dev_inter_ctu_pass_tiles(self.src[g].data_ptr(), self.ref.data_ptr(), self.ref_cu.data_ptr(), self.rec[g].data_ptr(), self.cu[g].data_ptr(), None, tw, th,
                                                           m, C.addressof(prm), self.xy[g].data_ptr(), self.n)
            if rc != 0:
                raise RuntimeError(f"kvz_hip_dev_inter_ctu_pass_tiles returned {rc}")
            ms = float(self.lib.kvz_hip_dev_inter_kernel_ms())
            self.lib.kvz_hip_dev_cu_dbk_from_info(self.cu[g].data_ptr(), m * (tw // 4) * (th // 4), self.dbk[g].data_ptr())
            rc = self.lib.kvz_hip_dev_loop_filters_inter(self.src[g].data_ptr(), self.rec[g].data_ptr(), tw, th, m, self.dbk[g].data_ptr(), prm.qp, 1, prm.deblock, 0, 0, prm.sao, 1, None, None, None)
            if rc != 0:
                raise RuntimeError(f"kvz_hip_dev_loop_filters_inter returned {rc}")
            self.lib.kvz_hip_dev_sync()  # this thread's stream of the library
            return ms
        groups = list(self.groups.items())
        if len(groups) > 1:
            # kvazaar's uniform grid gives this rank tiles of two sizes: their passes are two persistent launches, and the first one launched fills the device while
            # each holds fewer serial tile chains than the device has workgroup slots.  So: side by side, each on its share of the slots (kvz_hip_dev_inter_set_share), from
            # two host threads -- a thread has its own stream and scratch in the library, and the blocking calls release the GIL.
            if getattr(self, "_pool", None) is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=len(groups))
            self.pass_ms = max(f.result() for f in [self._pool.submit(group_pass, g, mem, len(groups)) for g, mem in groups])
        else:
            self.pass_ms = sum(group_pass(g, mem) for g, mem in groups)
        self.exchange()

    def slots_per_cu(self):
        """workgroups of the inter pass's kernel that fit a CU (kvz_hip_dev_inter_slots_per_cu: its occupancy; 12 with the round-4 build)"""
        self.lib.kvz_hip_dev_inter_slots_per_cu.restype = C.c_int
        return int(self.lib.kvz_hip_dev_inter_slots_per_cu())

    def exchange(self):
        """all-gather of the tiles' pictures and CU records, pasted into every rank's reference frames"""
        torch = self.torch
        n = self.n
        slot_of = {ti: k for k, ti in enumerate(self.mine)}
        for (tw, th), members in self.groups.items():
            ts, tc = tw * th * 3 // 2, (tw // 4) * (th // 4) * self.rec_item
            for j, ti in enumerate(members):
                self.send_px.view(self.per_rank, n, self.slot_px)[slot_of[ti], :, :ts] = self.rec[(tw, th)].view(len(members), n, ts)[j]
                self.send_cu.view(self.per_rank, n, self.slot_cu)[slot_of[ti], :, :tc] = self.cu[(tw, th)].view(len(members), n, tc)[j]
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.recv_px, self.send_px)
            self.dist.all_gather_into_tensor(self.recv_cu, self.send_cu)
        from . import sharding
        ys, cs = self.w * self.h, (self.w // 2) * (self.h // 2)
        fr = self.ref.view(n, self.fs)
        Y, U, V = fr[:, :ys].view(n, self.h, self.w), fr[:, ys:ys + cs].view(n, self.h // 2, self.w // 2), fr[:, ys + cs:].view(n, self.h // 2, self.w // 2)
        CU = self.ref_cu.view(n, self.h // 4, self.w // 4, self.rec_item)
        px, cu = self.recv_px.view(self.world, self.per_rank, n, self.slot_px), self.recv_cu.view(self.world, self.per_rank, n, self.slot_cu)
        for r in range(self.world):
            for k, ti in enumerate(sharding.tiles_of_rank(len(self.tiles), r, self.world)):
                x, y, tw, th = self.tiles[ti]
                c = (tw // 2) * (th // 2)
                t = px[r, k]
                Y[:, y:y + th, x:x + tw] = t[:, :tw * th].view(n, th, tw)
                U[:, y // 2:(y + th) // 2, x // 2:(x + tw) // 2] = t[:, tw * th:tw * th + c].view(n, th // 2, tw // 2)
                V[:, y // 2:(y + th) // 2, x // 2:(x + tw) // 2] = t[:, tw * th + c:tw * th + 2 * c].view(n, th // 2, tw // 2)
                CU[:, y // 4:(y + th) // 4, x // 4:(x + tw) // 4] = cu[r, k][:, :(tw // 4) * (th // 4) * self.rec_item].view(n, th // 4, tw // 4, self.rec_item)
        torch.cuda.synchronize()

    def download_reference(self, i):
        """the assembled frame and CU records of sequence i after the last exchange"""
        frame = self.ref.view(self.n, self.fs)[i].cpu().numpy().copy()
        cu = self.ref_cu.view(self.n, self.cells * self.rec_item)[i].cpu().numpy().copy().view(CU_DTYPE).reshape(self.h // 4, self.w // 4)
        return frame, cu
