"""bench.py --gpus N launches its ranks itself; without a GPU every rank fails and the launcher must say so with a
non-zero exit code (never a hang, never a made-up line)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_reports_failed_ranks_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_sharded.py::test_bench_launches_its_own_ranks")
    env = dict(os.environ, SQMC_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--equil", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "ranks failed" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_launcher_takes_the_other_ranks_down_when_one_dies():
    """a rank that dies after the start leaves its peers waiting in a collective: the launcher ends them and reports, promptly"""
    import time
    env = dict(os.environ, SQMC_BENCH_TEST_ONE_RANK_DIES="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.time() - t0 < 60
    assert "ranks failed" in r.stderr and "(1, 3)" in r.stderr
