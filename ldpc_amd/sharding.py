"""Multi-GPU sharding of a Monte-Carlo run: frames are independent units, so each
rank decodes its own contiguous range of global frame indices and the only
exchange is one all-reduce of the error counters per sweep point -- the
reference's `sum` over parfor results (test_mex_B_float.m:24-30, :73-78).
Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
import numpy as np


def frame_range(total_frames, rank, world):
    """[lo, hi) of global frame indices owned by `rank`: balanced, contiguous, and
    independent of how many ranks there are for the union."""
    base, extra = divmod(int(total_frames), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def frame_seed(base_seed, global_frame):
    """Counter-style seed of one frame, so results do not depend on the sharding."""
    return (int(base_seed) * 0x9E3779B97F4A7C15 + int(global_frame) * 0xD1B54A32D192ED03) % (1 << 63)


def reduce_counters(counters, device=None):
    """Sum of a small integer vector over all ranks: [failures, undetected, frames,
    iteration histogram ...].  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.asarray(counters, dtype=np.int64), device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def reduce_max(value, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def run_sharded(mc, total_frames, channel, param, llr_dtype, max_iter, seed=1, device=None, **kw):
    """One sweep point over all ranks: rank r runs the device Monte-Carlo loop
    (ldpc_amd.MonteCarlo.run) on its own range of global frame indices -- the frames are
    keyed by that index, so the union is the same run whatever the world size -- and the
    counters are summed by the path's only collective.  Returns the summed dict."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    lo, hi = frame_range(total_frames, rank, world)
    part = mc.run(hi - lo, channel, param, llr_dtype, max_iter, seed=seed, first_frame=lo, **kw)
    keys = list(part)
    total = reduce_counters([part[k] for k in keys], device=device)
    return {k: int(v) for k, v in zip(keys, total)}
