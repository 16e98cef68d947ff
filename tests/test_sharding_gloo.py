"""CPU, world_size 2 over gloo: the N>1 path of the Monte-Carlo sharding (frame ranges,
sharding-independent frame seeds, the counter all-reduce)."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ldpc_amd.sharding import frame_range, frame_seed, reduce_counters, reduce_max
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = frame_range(total, rank, world)
    # stand-in for the decode of frames lo..hi: per-frame outcome derived from the frame seed
    fails = undet = 0
    hist = np.zeros(11, np.int64)
    for f in range(lo, hi):
        r = np.random.default_rng(frame_seed(0x1DBC0001, f)).integers(0, 100)
        fails += r < 10
        undet += r == 10
        hist[r % 11] += 1
    c = reduce_counters(np.concatenate([[fails, undet, hi - lo], hist]))
    t = reduce_max(1.0 + rank)
    dist.barrier()
    if rank == 0:
        np.save(out, np.concatenate([c, [t]]))
    dist.destroy_process_group()


def test_two_rank_reduce_equals_single_rank(tmp_path):
    from ldpc_amd.sharding import frame_range, frame_seed
    total = 1001
    out = str(tmp_path / "c.npy")
    mp.spawn(_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    got = np.load(out)
    fails = undet = 0
    hist = np.zeros(11, np.int64)
    for f in range(total):
        r = np.random.default_rng(frame_seed(0x1DBC0001, f)).integers(0, 100)
        fails += r < 10
        undet += r == 10
        hist[r % 11] += 1
    assert got[:3].tolist() == [fails, undet, total]
    assert np.array_equal(got[3:14], hist)
    assert got[14] == 2.0
    # ranges tile [0, total) for any world size
    for w in (1, 2, 3, 8):
        r = [frame_range(total, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == total and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


class _FakeMonteCarlo:
    """Stands in for ldpc_amd.MonteCarlo on the CPU: per-frame outcomes derived from the
    numpy restatement of the device generator (frames keyed by their global index)."""

    def run(self, nframes, channel, param, llr_dtype, max_iter, seed=1, first_frame=0):
        sys.path.insert(0, ROOT)
        from oracle import mc_frontend as ref
        c = dict(frames=0, not_converged=0, undetected=0, bit_errors=0)
        for f in range(first_frame, first_frame + nframes):
            err, _ = ref.bsc_exact_pattern(64, param, seed, f)
            w = int(np.flatnonzero(err)[0])
            c["frames"] += 1
            c["not_converged"] += w < 8
            c["undetected"] += w == 9
            c["bit_errors"] += w
        return c


def _mc_worker(rank, world, port, total, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ldpc_amd.sharding import run_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = run_sharded(_FakeMonteCarlo(), total, 0, 0.1, 2, 10, seed=5)
    if rank == 0:
        np.save(out, np.array([c[k] for k in ("frames", "not_converged", "undetected", "bit_errors")]))
    dist.destroy_process_group()


def test_run_sharded_two_ranks_equals_one(tmp_path):
    total = 333
    out = str(tmp_path / "m.npy")
    mp.spawn(_mc_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    one = _FakeMonteCarlo().run(total, 0, 0.1, 2, 10, seed=5)
    assert np.load(out).tolist() == [one[k] for k in ("frames", "not_converged", "undetected", "bit_errors")]
    assert one["frames"] == total and 0 < one["not_converged"] < total


def _sweep_worker(rank, world, port, out, frames, min_failures):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from ldpc_amd.sweep import sweep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sweep(_FakeMonteCarlo(), [0.1, 0.2, 0.3], frames, 0, 2, 10, seed=9, out_path=out, min_failures=min_failures,
          chunk=50)
    dist.destroy_process_group()


def test_sweep_two_ranks_equals_one_and_resumes(tmp_path):
    """The sweep driver (test_mex_B_float.m:20-36 / parallel.py:385-396 semantics): per-point
    counters do not depend on the world size or on chunking; the JSON-lines checkpoint lets a
    restart skip finished points (the reference's save('full_env.mat') per point)."""
    import json
    from ldpc_amd.sweep import load_points, sweep
    two = str(tmp_path / "two.jsonl")
    mp.spawn(_sweep_worker, args=(2, _free_port(), two, 230, None), nprocs=2, join=True)
    one = sweep(_FakeMonteCarlo(), [0.1, 0.2, 0.3], 230, 0, 2, 10, seed=9)
    got = load_points(two)
    assert sorted(got) == [0, 1, 2]
    for r in one:
        g = got[r["index"]]
        assert g["world"] == 2 and r["world"] == 1
        assert g["counters"] == r["counters"] and g["frames"] == 230
        assert g["fer"] == r["fer"] == r["failures"] / 230
    # different points draw different frames
    assert one[0]["counters"] != one[1]["counters"] or one[1]["counters"] != one[2]["counters"]
    # a second two-rank run on the finished file runs nothing (rank 0 reads it, every rank skips alike)
    mp.spawn(_sweep_worker, args=(2, _free_port(), two, 230, None), nprocs=2, join=True)
    assert len(open(two).read().splitlines()) == 3
    # resume: drop the last record (and leave a line cut off mid-write); only that point is re-run
    lines = open(two).read().splitlines()
    part = str(tmp_path / "part.jsonl")
    with open(part, "w") as f:
        f.write("\n".join(lines[:2]) + "\n" + lines[2][:40])

    class Counting(_FakeMonteCarlo):
        calls = 0

        def run(self, *a, **k):
            Counting.calls += 1
            return super().run(*a, **k)
    with open(part, "a") as f:
        f.write("\n")
    recs = sweep(Counting(), [0.1, 0.2, 0.3], 230, 0, 2, 10, seed=9, out_path=part)
    assert Counting.calls == 1 and recs[2]["counters"] == one[2]["counters"]
    assert [json.loads(l)["index"] for l in open(part).read().splitlines() if l.startswith('{"index"') and l.endswith("}")] == [0, 1, 2]
    # early stop: chunks of 50 frames until >= 12 failures, same stop on 1 and 2 ranks
    stop2 = str(tmp_path / "stop2.jsonl")
    mp.spawn(_sweep_worker, args=(2, _free_port(), stop2, 1000, 12), nprocs=2, join=True)
    stop1 = sweep(_FakeMonteCarlo(), [0.1, 0.2, 0.3], 1000, 0, 2, 10, seed=9, min_failures=12, chunk=50)
    g2 = load_points(stop2)
    for r in stop1:
        assert r["frames"] % 50 == 0 and r["failures"] + r["undetected"] >= 12 or r["frames"] == 1000
        assert g2[r["index"]]["counters"] == r["counters"]
    assert any(r["frames"] < 1000 for r in stop1)
