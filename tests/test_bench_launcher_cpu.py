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


def _alive(pids):
    out = []
    for pid in pids:
        try:
            with open("/proc/%d/stat" % pid) as fh:
                if fh.read().split(") ")[1][0] != "Z":  # a zombie is dead
                    out.append(pid)
        except OSError:
            pass
    return out


def _descendants(pid):
    """pids of every process below `pid` (by /proc's ppid links)"""
    kids = {}
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                with open("/proc/%s/stat" % d) as fh:
                    ppid = int(fh.read().split(") ")[1].split()[1])
                kids.setdefault(ppid, []).append(int(d))
            except (OSError, IndexError, ValueError):
                pass
    out, todo = [], [pid]
    while todo:
        for k in kids.get(todo.pop(), []):
            out.append(k)
            todo.append(k)
    return out


def test_ending_the_launcher_ends_the_ranks():
    """`timeout ... python bench.py --gpus 2` used to kill the launcher only — torch.distributed.run and the ranks (stuck in a
    synchronisation, holding the GPU) lived on.  The ranks now get a process group of their own and whatever ends the launcher takes it down:
    SIGTERM to the launcher while both ranks sleep -> within seconds none of its descendants is alive."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ORC_BENCH_DRY_RUN_SLEEP"] = "120"
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        deadline = time.time() + 60
        ranks = []
        while time.time() < deadline and len(ranks) < 3:  # torch.distributed.run + two ranks
            time.sleep(0.5)
            ranks = _descendants(p.pid)
        assert len(ranks) >= 3, ranks
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=30)
        deadline = time.time() + 20
        while time.time() < deadline and _alive(ranks):
            time.sleep(0.5)
        assert not _alive(ranks), "still alive after the launcher ended: %s" % _alive(ranks)
    finally:
        if p.poll() is None:
            p.kill()
        for pid in _alive(_descendants(p.pid)):
            os.kill(pid, signal.SIGKILL)


def test_a_stuck_rank_reports_and_exits_nonzero():
    """for N > 1 the watchdog is on by default (bench.WATCHDOG_DEFAULT_S; here 3 s): a rank without progress prints its Python
    stacks, its hardware-queue note and the library's stream states, and EXITS with code 3 — the launcher returns non-zero, no silent time-out."""
    import bench
    assert bench.WATCHDOG_DEFAULT_S >= 120
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ORC_BENCH_DRY_RUN_SLEEP="60", ORC_BENCH_WATCHDOG="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    # (the two ranks write to one pipe: their lines may interleave, so count the pieces rather than match whole lines)
    assert r.stderr.count("no progress for 3 s after 'start'") == 2 and "[bench watchdog r" in r.stderr, r.stderr[-3000:]
    assert "most recent call first" in r.stderr  # the stacks


def test_strong_scaling_cut_is_the_one_mesh():
    """bench.py --scaling strong: rank r of N owns layers [r nz / N, (r + 1) nz / N) of the SAME nx x ny x nz channel (parallel.slab_arrays with
    nz / N layers per rank): the owned cells of the N slabs tile the whole mesh in ORC's order, with the whole mesh's geometry (to rounding) — so the initial
    fields (a function of the centroids) and every matrix row are those of the N = 1 run."""
    import numpy as np
    from orc_amd.mesh import hex_channel
    from orc_amd.parallel import slab_arrays
    import bench
    nx, ny, nz, world = 6, 5, 8, 4
    whole = hex_channel(nx, ny, nz)
    cc = np.asarray(whole["cell_centroid"])
    f_whole = bench.initial_fields(cc)
    seen = []
    for rank in range(world):
        a, halo, gids = slab_arrays(nx, ny, nz // world, rank, world)
        n_own = halo["n_owned"]
        assert halo["n_global"] == whole.n_cells
        seen.append(gids[:n_own])
        assert np.allclose(np.asarray(a["cell_centroid"]), cc[gids], rtol=0, atol=1e-18)
        # (a slab is generated as a channel of its own depth and shifted: its geometry equals the whole mesh's to rounding, not to the bit)
        assert np.allclose(np.asarray(a["cell_volume"]), np.asarray(whole["cell_volume"])[gids], rtol=1e-12, atol=0)
        for x, y in zip(bench.initial_fields(np.asarray(a["cell_centroid"]), ids=gids), f_whole):  # (what make_slab_solver(global_noise=True) sets)
            assert np.allclose(x, y[gids], rtol=1e-12, atol=0)
    assert np.array_equal(np.concatenate(seen), np.arange(whole.n_cells))
