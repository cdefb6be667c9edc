"""bench.py's launch contract on the CPU (no GPU is touched: MA_BENCH_DRY_RUN=1 stops after the rendezvous)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = dict(os.environ, MA_BENCH_DRY_RUN="1", **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if k not in env:
            e.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_flag_starts_that_many_ranks_without_an_outer_torchrun():
    """`python bench.py --gpus 2` (no torchrun environment): the process starts two ranks itself before anything touches a GPU,
    they rendezvous over 127.0.0.1 (gloo here) and rank 0 reports n_gpus = 2 (the flag was parsed and ignored)."""
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-500:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["gpus_flag"] == 2


def test_single_rank_needs_no_launcher():
    r = _run(["--gpus", "1"])
    assert r.returncode == 0
    assert json.loads(r.stdout.splitlines()[-1])["n_gpus"] == 1


def test_world_size_must_equal_the_gpus_flag():
    """inside a torchrun environment of another size the bench refuses: a 1-rank number is never printed as an N-GPU one"""
    r = _run(["--gpus", "8"], WORLD_SIZE="1", RANK="0")
    assert r.returncode != 0 and "refusing" in r.stderr
    r = _run(["--gpus", "1"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 0
