"""bench.py --gpus N must start N ranks itself (the driver runs `python bench.py --gpus 8` with
no launcher for the scaling curve): the launcher path is driven here with 2 ranks on the CPU
(`--dry`: gloo group, barrier, the two all-reduces, no decoding) -- the counterpart of the
reference's parfor + sum aggregation (test_mex_B_float.m:24-30,73-78)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_2_launches_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--dry", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_in_reduce"] == 2 and out["steps"] == 3 and out["dry"] is True


def test_single_rank_dry_needs_no_launcher():
    r = _run(["--dry", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1


def test_world_size_must_match_gpus():
    """Under an external launcher (the driver's torch.distributed.run) --gpus must equal the
    number of ranks started; a mismatch is an error, not a silently smaller job."""
    r = _run(["--gpus", "4", "--dry"], {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert r.returncode != 0 and "--gpus 4" in r.stderr
