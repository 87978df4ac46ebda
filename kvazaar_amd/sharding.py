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
