"""Multi-GPU plumbing of the batched pass: how frames are dealt to ranks and how step times are combined.

All-intra frames (and tiles) are independent sub-problems (encoderstate.c:944-979, 1599-1620), so ranks never exchange
pixels: the only collectives are a barrier around the timed region and a MAX-reduce of the elapsed time, exactly what
the bench contract asks for.  Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import time


def frames_for_rank(n_frames, rank, world):
    """Contiguous, balanced shard [lo, hi) of a job of n_frames for `rank` (strong-scaling split of a given clip)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def tile_grid(width, height, cols, rows):
    """Tile rectangles (x, y, w, h) in raster order for kvazaar's `--tiles COLSxROWS` with uniform spacing: boundaries in
    CTU units are i * size_in_ctus // count (encoder.c:383-404, H.265 (6-3)/(6-4)).  Tiles are independent sub-pictures:
    intra prediction, deblocking and SAO never cross their edges (encoder_state-bitstream.c:545,549), so a tile goes
    through the CTU pass as a picture of its own size."""
    wl, hl = (width + 63) // 64, (height + 63) // 64
    if not (1 <= cols <= wl and 1 <= rows <= hl):
        raise ValueError(f"{cols}x{rows} tiles do not fit a {wl}x{hl}-CTU picture")
    col_bd = [i * wl // cols for i in range(cols + 1)]
    row_bd = [i * hl // rows for i in range(rows + 1)]
    tiles = []
    for r in range(rows):
        for c in range(cols):
            x, y = col_bd[c] * 64, row_bd[r] * 64
            tiles.append((x, y, min(width, col_bd[c + 1] * 64) - x, min(height, row_bd[r + 1] * 64) - y))
    return tiles


def crop_tile(yuv, width, height, tile):
    """The planar 4:2:0 sub-picture of one tile (numpy uint8 array in, Y|U|V planar array out)."""
    import numpy as np
    x, y, w, h = tile
    ys, cs = width * height, (width // 2) * (height // 2)
    Y = yuv[:ys].reshape(height, width)[y:y + h, x:x + w]
    U = yuv[ys:ys + cs].reshape(height // 2, width // 2)[y // 2:(y + h) // 2, x // 2:(x + w) // 2]
    V = yuv[ys + cs:ys + 2 * cs].reshape(height // 2, width // 2)[y // 2:(y + h) // 2, x // 2:(x + w) // 2]
    return np.concatenate([Y.reshape(-1), U.reshape(-1), V.reshape(-1)])


def paste_tile(dst, width, height, tile, sub):
    """Inverse of crop_tile: writes a tile's planar picture `sub` into the planar frame `dst` in place."""
    x, y, w, h = tile
    ys, cs = width * height, (width // 2) * (height // 2)
    dst[:ys].reshape(height, width)[y:y + h, x:x + w] = sub[:w * h].reshape(h, w)
    c = (w // 2) * (h // 2)
    dst[ys:ys + cs].reshape(height // 2, width // 2)[y // 2:(y + h) // 2, x // 2:(x + w) // 2] = sub[w * h:w * h + c].reshape(h // 2, w // 2)
    dst[ys + cs:ys + 2 * cs].reshape(height // 2, width // 2)[y // 2:(y + h) // 2, x // 2:(x + w) // 2] = sub[w * h + c:].reshape(h // 2, w // 2)


def barrier(dist, device_sync=None):
    if device_sync:
        device_sync()
    if dist is not None and dist.is_initialized():
        dist.barrier()
    if device_sync:
        device_sync()


def timed_steps(step_fn, steps, dist=None, device_sync=None, tensor_device="cpu"):
    """Runs step_fn() `steps` times bracketed by sync + barrier on both sides; returns the MAX over ranks of the
    elapsed seconds (every rank gets the same number)."""
    import torch
    barrier(dist, device_sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    barrier(dist, device_sync)
    dt = time.perf_counter() - t0
    if dist is not None and dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64, device=tensor_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# ---- the one data-path collective of the sharded INTER configuration -----------------------------------------------
# With tiles dealt to ranks, motion vectors of the next picture may point into any tile of the reference picture
# (search_inter.c:149-178: only --mv-constraint keeps them inside), so after a picture's loop filters every rank needs every tile's final
# reconstruction: one all-gather per picture.  Tiles of a uniform grid differ in size by at most one CTU row / column, so each rank
# contributes a fixed-size slot per tile (the largest tile's planar 4:2:0 bytes) and pastes the received slots into its full reference
# frame.  Backend "nccl" = RCCL over xGMI on the GPUs (uint8 tensors on the device), "gloo" in the CPU tests.

def tile_slot_bytes(tiles):
    """bytes of one exchange slot: the largest tile's Y|U|V planar picture"""
    return max(t[2] * t[3] * 3 // 2 for t in tiles)


def tiles_of_rank(n_tiles, rank, world):
    lo, hi = frames_for_rank(n_tiles, rank, world)
    return list(range(lo, hi))


def exchange_plan(width, height, cols, rows, world):
    """static description of the exchange for a picture: tiles, slot size, slots per rank (ranks with fewer tiles send padding), and the
    bytes a rank receives per picture -- what the bench line reports next to the measured time"""
    tiles = tile_grid(width, height, cols, rows)
    per_rank = max(len(tiles_of_rank(len(tiles), r, world)) for r in range(world))
    slot = tile_slot_bytes(tiles)
    return {"tiles": tiles, "slot_bytes": slot, "slots_per_rank": per_rank, "frame_bytes": width * height * 3 // 2,
            "recv_bytes_per_rank": (world - 1) * per_rank * slot}


class ReferenceExchange:
    """The per-picture exchange with everything that does not change between pictures set up ONCE: the send / receive buffers (a rank's tile pictures are
    produced straight into its send slots: send_slot(k)), the paste table, and -- on the GPU -- ONE paste launch of the library (kvz_hip_dev_paste_tiles) on
    torch's current stream instead of 3 x tiles strided torch copies.  exchange() = all_gather_into_tensor + paste."""

    def __init__(self, dist, plan, rank, world, width, height, device, lib=None):
        import numpy as np
        import torch
        self.dist, self.plan, self.rank, self.world, self.w, self.h, self.lib = dist, plan, rank, world, width, height, lib
        tiles, self.slot, self.per_rank = plan["tiles"], plan["slot_bytes"], plan["slots_per_rank"]
        self.mine = tiles_of_rank(len(tiles), rank, world)
        self.send = torch.zeros(self.per_rank * self.slot, dtype=torch.uint8, device=device)
        self.recv = torch.empty(world * self.per_rank * self.slot, dtype=torch.uint8, device=device)
        self.frame = torch.empty(width * height * 3 // 2, dtype=torch.uint8, device=device)
        table = []
        for r in range(world):
            for k, ti in enumerate(tiles_of_rank(len(tiles), r, world)):
                table.append(tuple(tiles[ti]) + (r * self.per_rank + k,))
        self.table = table
        self.table_np = np.ascontiguousarray(np.array(table, np.int32).reshape(-1))
        self.on_gpu = self.send.is_cuda and lib is not None
        if self.on_gpu:
            import ctypes as C
            lib.kvz_hip_dev_paste_tiles.restype = C.c_int
            lib.kvz_hip_dev_paste_tiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_void_p]

    def send_slot(self, k):
        """the k-th of this rank's tile pictures (a view into the send buffer: write the tile's planar bytes here)"""
        ti = self.mine[k]
        n = self.plan["tiles"][ti][2] * self.plan["tiles"][ti][3] * 3 // 2
        return self.send[k * self.slot:k * self.slot + n]

    def exchange(self):
        import torch
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.recv, self.send)
            src = self.recv
        else:
            src = self.send  # one rank holds every tile: nothing moves
        if self.on_gpu:
            rc = self.lib.kvz_hip_dev_paste_tiles(self.frame.data_ptr(), self.w, self.h, src.data_ptr(), self.slot, self.table_np.ctypes.data, len(self.table),
                                                  torch.cuda.current_stream().cuda_stream)
            if rc != 0:
                raise RuntimeError("kvz_hip_dev_paste_tiles failed")
            return self.frame
        ys, cs = self.w * self.h, (self.w // 2) * (self.h // 2)
        Y, U, V = self.frame[:ys].view(self.h, self.w), self.frame[ys:ys + cs].view(self.h // 2, self.w // 2), self.frame[ys + cs:ys + 2 * cs].view(self.h // 2, self.w // 2)
        for (x, y, w, h, slot_i) in self.table:
            base, c = slot_i * self.slot, (w // 2) * (h // 2)
            Y[y:y + h, x:x + w] = src[base:base + w * h].view(h, w)
            U[y // 2:(y + h) // 2, x // 2:(x + w) // 2] = src[base + w * h:base + w * h + c].view(h // 2, w // 2)
            V[y // 2:(y + h) // 2, x // 2:(x + w) // 2] = src[base + w * h + c:base + w * h + 2 * c].view(h // 2, w // 2)
        return self.frame


def allgather_cu_records(dist, plan, rank, world, local, width, height, dtype):
    """The CU records of a picture whose tiles were searched on different ranks, assembled on every rank (what the next picture's motion search reads of its reference
    picture: the co-located CU of search_inter.c:1286-1339; kvz_hip_inter_params::ref_width / tile_x).  local: {tile index: (h / 4, w / 4) array of `dtype` records} for this
    rank's tiles.  One all_gather of fixed-size slots (a record per 4x4 unit: 1/8 of the picture's bytes) and a paste, host side."""
    import numpy as np
    import torch
    tiles, per_rank = plan["tiles"], plan["slots_per_rank"]
    item = np.dtype(dtype).itemsize
    slot = max((t[2] // 4) * (t[3] // 4) for t in tiles) * item
    send = torch.zeros(per_rank * slot, dtype=torch.uint8)
    for k, ti in enumerate(tiles_of_rank(len(tiles), rank, world)):
        b = np.ascontiguousarray(local[ti]).view(np.uint8).reshape(-1)
        send[k * slot:k * slot + b.size] = torch.from_numpy(b.copy())
    if world > 1:
        recv = torch.empty(world * per_rank * slot, dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send)
    else:
        recv = send
    out = np.zeros((height // 4, width // 4), dtype)
    raw = recv.numpy()
    for r in range(world):
        for k, ti in enumerate(tiles_of_rank(len(tiles), r, world)):
            x, y, w, h = tiles[ti]
            base = (r * per_rank + k) * slot
            out[y // 4:(y + h) // 4, x // 4:(x + w) // 4] = raw[base:base + (w // 4) * (h // 4) * item].view(dtype).reshape(h // 4, w // 4)
    return out


def allgather_reference_frame(dist, plan, rank, world, local_tiles, width, height, out_frame=None, state=None, lib=None):
    """local_tiles: {tile index: 1-D uint8 torch tensor holding that tile's planar picture} for this rank's tiles (any device).
    Returns the full planar reference frame (1-D uint8 tensor on the same device) assembled from every rank's tiles.  One-shot convenience around
    ReferenceExchange (pass `state` to reuse its buffers between pictures; producers that write into state.send_slot() skip the packing copy)."""
    import torch
    device = next(iter(local_tiles.values())).device if local_tiles else torch.device("cpu")
    if state is None:
        state = ReferenceExchange(dist, plan, rank, world, width, height, device, lib)
    for k, ti in enumerate(state.mine):
        t = local_tiles[ti]
        state.send_slot(k).copy_(t)
    frame = state.exchange()
    if out_frame is not None:
        out_frame.copy_(frame)
        return out_frame
    return frame
