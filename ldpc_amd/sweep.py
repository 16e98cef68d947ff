"""Channel-parameter sweeps: the loop of the reference's experiment drivers around the
Monte-Carlo harness, with the frames of every point sharded over the ranks.

What it mirrors (harness semantics only; MATLAB itself is out of scope):
  * test_mex_B_float.m:9-36 -- for each bit_err_prob: `parallel` batches of `serial`
    trials under parfor, result(1)/result(2) summed over the batches (:73-78),
    `save('full_env.mat')` after every point (:35) so a long run can be picked up again;
  * parallel.py:385-396 -- Pool.map over the channel points, each returning
    (parameter, error count, undetected count).
Here one point = one call of `sharding.run_sharded` (every rank runs the device loop on
its own global frame range; one all-reduce of the counters), the per-point record is a
JSON line appended by rank 0 (the checkpoint), and a restart skips the points already
in the file.  Frames are keyed by (seed, point index, global frame index): the sweep's
counters do not depend on the number of ranks or on where it was interrupted.
"""
import json
import math
import os
import time

from .sharding import run_sharded, reduce_max


def ebn0_to_sigma(ebn0_db, rate):
    """BPSK over AWGN, unit symbol energy: sigma^2 = 1 / (2 R 10^(Eb/N0 / 10))."""
    return math.sqrt(1.0 / (2.0 * rate * 10.0 ** (ebn0_db / 10.0)))


def point_seed(seed, index):
    """Seed of sweep point `index` (its frames are then numbered from 0)."""
    return (int(seed) * 0x9E3779B97F4A7C15 + (int(index) + 1) * 0xBF58476D1CE4E5B9) % (1 << 63)


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def load_points(path):
    """Records of a sweep file, keyed by point index (later lines win)."""
    done = {}
    if path and os.path.exists(path):
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                try:
                    r = json.loads(line)
                except ValueError:
                    continue            # a line cut off by an interrupted run
                if "index" in r and r.get("complete"):
                    done[int(r["index"])] = r
    return done


def _bcast_done(done_idx, device):
    """Rank 0 read the file; every rank must skip the same points."""
    import torch
    import torch.distributed as dist
    rank, world = _rank_world()
    if world == 1:
        return done_idx
    obj = [sorted(done_idx) if rank == 0 else None]
    dist.broadcast_object_list(obj, src=0, device=device)
    return set(obj[0])


def sweep(mc, params, frames_per_point, channel, llr_dtype, max_iter, seed=1, out_path=None,
          resume=True, min_failures=None, chunk=None, device=None, label=None, mc_params=None, **kw):
    """Runs every point of `params` (what is recorded: p, sigma or Eb/N0) and returns the
    list of per-point records.

    mc_params      the channel parameter handed to the device loop per point (default: params);
                   e.g. params = Eb/N0 in dB, mc_params = the matching sigmas.
    min_failures   if set, a point is run in chunks of `chunk` frames (default
                   frames_per_point / 10) and stops once failures + undetected reach it --
                   the usual early stop of an FER run; every rank sees the same summed
                   counters, so all ranks stop together.
    out_path       JSON-lines checkpoint (rank 0 appends one record per finished point);
                   with resume=True the points already there are not run again.
    Record: index, param, frames, failures (errors[0]: dec() did not reach the syndrome),
    undetected (errors[1]: reached it with a word != x), fer, undetected_rate, ber,
    avg_iterations, seconds (max over ranks), world.
    """
    rank, world = _rank_world()
    params = list(params)
    mc_params = list(params if mc_params is None else mc_params)
    if len(mc_params) != len(params):
        raise ValueError("mc_params must have one entry per point")
    done = load_points(out_path) if (resume and rank == 0) else {}
    done_idx = _bcast_done(set(done), device)
    n = getattr(getattr(mc, "code", None), "n", None)
    records = []
    for idx, (par, mpar) in enumerate(zip(params, mc_params)):
        if idx in done_idx:
            records.append(done.get(idx, {"index": idx, "param": par, "complete": True, "skipped": True}))
            continue
        total = dict()
        t0 = time.perf_counter()
        step = int(frames_per_point) if not min_failures else int(chunk or max(1, frames_per_point // 10))
        first = 0
        while first < frames_per_point:
            cnt = min(step, frames_per_point - first)
            part = run_sharded(_Offset(mc, first), cnt, channel, mpar, llr_dtype, max_iter,
                               seed=point_seed(seed, idx), device=device, **kw)
            for k, v in part.items():
                total[k] = total.get(k, 0) + v
            first += cnt
            if min_failures and total.get("not_converged", 0) + total.get("undetected", 0) >= min_failures:
                break
        secs = reduce_max(time.perf_counter() - t0, device=device)
        frames = max(1, total.get("frames", first))
        rec = {"index": idx, "param": par, "mc_param": mpar, "label": label, "frames": total.get("frames", first),
               "failures": total.get("not_converged", 0), "undetected": total.get("undetected", 0),
               "fer": total.get("not_converged", 0) / frames,
               "undetected_rate": total.get("undetected", 0) / frames,
               "ber": (total["bit_errors"] / (frames * n)) if (n and "bit_errors" in total) else None,
               "avg_iterations": (total["iterations"] / frames) if "iterations" in total else None,
               "seconds": secs, "world": world, "max_iter": max_iter, "seed": seed, "complete": True,
               "counters": total}
        records.append(rec)
        if rank == 0 and out_path:
            with open(out_path, "a") as f:
                f.write(json.dumps(rec) + "\n")
                f.flush()
                os.fsync(f.fileno())
    return records


class _Offset:
    """run_sharded numbers a point's frames from 0; a later chunk of the same point starts
    at `base` so that chunking does not change which frames a point consists of."""

    def __init__(self, mc, base):
        self.mc, self.base = mc, base

    def run(self, nframes, channel, param, llr_dtype, max_iter, seed=1, first_frame=0, **kw):
        return self.mc.run(nframes, channel, param, llr_dtype, max_iter, seed=seed,
                           first_frame=self.base + first_frame, **kw)
