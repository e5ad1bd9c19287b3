"""`python bench.py --gpus N` from a bare shell starts its own N ranks (one process per GPU, torch.distributed.run on
127.0.0.1) BEFORE anything touches a GPU in the parent; in this GPU-less container every rank gets as far as its first GPU
call and says so, and the parent passes the failure on."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_its_own_ranks(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU box runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    text = r.stdout + r.stderr
    assert r.returncode != 0
    assert "rank 0 of 2: no GPU visible" in text and "rank 1 of 2: no GPU visible" in text, text[-2000:]
