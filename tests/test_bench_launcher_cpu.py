"""`python bench.py --gpus N` must be startable exactly like `--gpus 1` (Missing #2): without WORLD_SIZE in the
environment the process becomes the launcher — it starts its N ranks as fresh children of torch.distributed.run (before it has
touched a GPU, never re-executing itself), forwards rank 0's JSON line and returns the children's exit code.  Rehearsed on the CPU
with --dry-run (control plane only: the ranks join gloo and count each other)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("n", [2, 3])
def test_bench_spawns_its_own_ranks(n):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # ONE line on stdout, whatever the ranks chatter about
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["ranks_seen"] == n and out["dry_run"] is True


def test_bench_refuses_a_mismatched_launch():
    """inside a launcher of 2 ranks, `--gpus 4` is a configuration error, not something to paper over"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
