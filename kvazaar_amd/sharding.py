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
