"""CPU tier: `python bench.py --gpus N` must start its own N ranks when no launcher set WORLD_SIZE (the driver's N = 1 command
shape with a larger N), instead of asserting on WORLD_SIZE.  Without a GPU every rank then stops at the "needs MI355X GPUs"
check -- which is what this test looks for: the launch happened, and the failure is the loud no-GPU one."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU tier runs the real thing (test_bench_two_ranks_one_gpu)")
def test_bench_self_launches_ranks_without_a_launcher():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2", "10", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "WORLD_SIZE=" not in out.stderr, out.stderr[-1500:]
    assert out.stderr.count("bench.py needs MI355X GPUs") >= 2, out.stderr[-1500:]   # both ranks got that far


def test_bench_rejects_a_mismatched_launcher():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2", "10"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode != 0 and "--gpus 2 but the launcher set WORLD_SIZE=3" in out.stderr
