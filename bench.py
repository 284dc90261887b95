#!/usr/bin/env python3
"""bench.py — SIMPLE iterations/s on the ~10M-cell synthetic hex channel (BASELINE.json configs[3]) on MI355X.

One "step" = one full SIMPLE iteration of solver::solve_steady (solver.rs:60-222): momentum assembly, three
momentum solves, pressure-correction assembly + solve, correction — all device-resident in liborc_amd.so.
N = 1: 400 x 160 x 160 = 10 240 000 hex cells, UMIST TVD momentum, Rhie-Chow, SecondOrder,
reference Multigrid arm (pairwise-aggregation AMG with BiCGSTAB smoother) + Jacobi preconditioning.
N > 1: weak scaling, every rank owns a 400 x 160 x 160 slab of a 400 x 160 x (160 N) channel.
Every warm-up and timed step restores the same device-side snapshot (state after two spin-up iterations) and runs one
full SIMPLE iteration, so all steps do identical work (the reference algorithm itself diverges on this mesh).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (CSR SpMV inside BiCGSTAB, HBM bound,
measured live with HIP events on the library stream) and `cpu_baseline` (the CPU oracle = restatement of
ORC's Rust path, 1 core, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PMC_FILE = "r05_spmv_pmc.json"  # written by scripts/pmc_summary.py from the rocprofv3 --pmc passes of scripts/gpu_pmc.sh
PMC_FILE_CONFIG5 = "r05_config5_spmv_pmc.json"  # the same for --workload config5 (scripts/gpu_pmc.sh ... --workload config5)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy ceiling)
REF_CELLS = 400 * 160 * 160


def initial_fields(cc, dp_dx=5.0, h=1e-3, mu=1e-3, lx=0.002, seed=0x4F5243, x0=0.0, lx_total=None, ids=None):
    """Analytical Poiseuille profile (tests.rs:26-29) with a 1e-6 relative splitmix64 perturbation (nothing is
    exactly zero or exactly equal between neighbours) and the matching linear pressure drop.  A rougher start
    (percent-level cell-to-cell noise = large mass imbalance per cell) makes the reference algorithm itself
    diverge on fine 3-D meshes with its default relaxation factors (checked with the CPU oracle)."""
    from orc_amd.mesh import splitmix64_uniform
    n = len(cc)
    y = cc[:, 1]
    lx_total = lx_total or lx
    # ids (global cell ids of a rank's cells): the perturbation of the WHOLE mesh's cell, so that a strong-scaling run starts from the N = 1 field
    u = 1.0 / (2.0 * mu) * dp_dx * (y * y - h * y) * (1.0 + 1e-6 * splitmix64_uniform(n, seed, ids))
    v = 1e-12 * splitmix64_uniform(n, seed + 1, ids)
    w = 1e-12 * splitmix64_uniform(n, seed + 2, ids)
    p = -dp_dx * lx_total * (1.0 - (cc[:, 0] + x0) / lx_total) * (1.0 + 1e-6 * splitmix64_uniform(n, seed + 3, ids))
    return u, v, w, p


def cpu_baseline(settings_kw, sample=(100, 40, 40), iters=6, mixed_ref_cells=None):
    """The CPU oracle (oracle/: single-threaded C restatement of ORC's Rust path) timed on a bounded sample of
    the same workload: same generator, BCs, settings and initial-field recipe at 1/64 of the cells (about 12 s).
    mixed_ref_cells: the config-5 workload — a 60 x 30 x 30-block mixed channel, scaled to one GPU's slab of that many cells."""
    from oracle import pyoracle as po
    from orc_amd.mesh import hex_channel, set_channel_bcs, set_mixed_channel_bcs
    if mixed_ref_cells:
        from orc_amd import parallel
        sample = (60, 30, 30)
        _a, _h, _g, a = parallel.mixed_slab_arrays(sample[0], sample[1], sample[2], 0, 1)
        set_mixed_channel_bcs(a)
    else:
        a = set_channel_bcs(hex_channel(*sample))
    nx, ny, nz = sample
    om = po.Mesh.from_arrays(a)
    u, v, w, p = initial_fields(np.asarray(a["cell_centroid"]))
    kw = dict(settings_kw)
    kw["frozen_diagonals"] = 0  # the reference's own mode
    kw["breakdown_guard"] = 0
    note = ""
    if int(kw["solver_type"]) >= 16:  # multicolour-GS extensions do not exist in ORC: time its own arm of the same family
        ref_solver = 2 if int(kw["solver_type"]) == 18 else 3
        note = "; the GS extension has no reference counterpart (linear_algebra.rs:219-246 panics): the oracle runs ORC's %s arm" % ("Multigrid" if ref_solver == 2 else "BiCGSTAB")
        kw["solver_type"] = ref_solver
    s = po.default_settings(**kw)
    t0 = time.perf_counter()
    st, _ = po.solve_steady(om, u, v, w, p, s, 1000.0, 1e-3, iters)
    dt = time.perf_counter() - t0
    n = a.n_cells
    ref_cells = mixed_ref_cells or REF_CELLS
    it_per_s_sample = iters / dt
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    import shutil
    return {
        "value": it_per_s_sample * n / ref_cells,
        "unit": "SIMPLE iterations/s (10.24M-cell equivalent)" if not mixed_ref_cells else "SIMPLE iterations/s (one GPU's mixed slab of %d cells)" % ref_cells,
        "cores": 1,
        "host_cpu": "%s (%d logical cores on the box; ORC's path is single-threaded)" % (cpu_model, os.cpu_count() or 0),
        "rust_toolchain": "present" if shutil.which("cargo") else "absent (ORC itself cannot be built: the port is timed)",
        "kind": "port",
        "sample": "%dx%dx%d %s (%d cells = 1/%d of the workload), %d SIMPLE iterations in %.2f s, status %d; scaled by cells"
                  % (nx, ny, nz, "blocks of the mixed channel" if mixed_ref_cells else "hex channel", n, ref_cells // n, iters, dt, st) + note
                  + "; the sample times iterations 1-%d from the seeded field (the GPU line times iteration spin-up + 1 of its run; an iteration's cost grows slowly with the state)" % iters,
    }


def spawn_ranks(n):
    """python bench.py --gpus N without a launcher: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process (one rank per GPU over RCCL, or over
    the host-staged transport on one GPU when ORC_BENCH_HOST_TRANSPORT=1), stream its output through and return its exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    # a rank that makes no progress for this many seconds prints its stacks and the library's stream states and EXITS non-zero (Watchdog
    # below): a first multi-GPU run that stalls must not end as a silent time-out of whoever started it
    env.setdefault("ORC_BENCH_WATCHDOG", str(WATCHDOG_DEFAULT_S))
    # (r04 set GPU_MAX_HW_QUEUES=2 here for ranks sharing one card; the library now keeps such ranks' streams in one priority class —
    # runtime.cpp, stream_create — which is the cause that setting worked around: DESIGN.md §7)
    import signal
    # the ranks get a session (= process group) of their own: whatever ends this launcher — a time-out's SIGTERM, Ctrl-C, an exception —
    # takes torch.distributed.run AND the ranks with it; nothing is left holding the GPU
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1, start_new_session=True)

    def end_group(sig):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass

    def forward(signum, _frame):
        end_group(signal.SIGTERM)
        raise SystemExit(128 + signum)

    old = {sg: signal.signal(sg, forward) for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    line, rc = None, 1
    try:
        for out in proc.stdout:  # rank 0 prints exactly one JSON line; anything else a rank writes to stdout goes to our stderr
            if out.startswith('{"metric"'):
                line = out
            else:
                sys.stderr.write(out)
        rc = proc.wait()
    finally:
        if proc.poll() is None:  # we are leaving early: end the whole group, politely, then for good
            end_group(signal.SIGTERM)
            try:
                proc.wait(timeout=10)
            except subprocess.TimeoutExpired:
                end_group(signal.SIGKILL)
                proc.wait()
        else:
            end_group(signal.SIGKILL)  # the launcher is gone; a rank that outlived it (it should not) goes too
        for sg, h in old.items():
            signal.signal(sg, h)
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks ended without a result line\n")
        rc = 1
    return rc


WATCHDOG_DEFAULT_S = 240  # N > 1: seconds without progress (mesh set-up, a spin-up iteration, a step) after which a rank reports and exits


class Watchdog:
    """A timer thread that is re-armed at every sign of progress (`kick`).  When it fires the rank writes to stderr: the Python stack of every
    thread (the main one sits in a ctypes call of the library), how many hardware queues the process holds (KFD sysfs, when readable), and which
    of the library's streams still hold work (orc_debug_stream_report: hipStreamQuery, never blocks) — then EXITS with code 3.  No restart, no
    re-exec: a process that has touched the GPU ends, its launcher sees a failed rank and ends the others."""

    def __init__(self, seconds, rank):
        self.seconds, self.rank, self.timer, self.where = seconds, rank, None, "start"

    def kick(self, where):
        import threading
        self.where = where
        if self.timer is not None:
            self.timer.cancel()
        if self.seconds > 0:
            self.timer = threading.Timer(self.seconds, self.bark)
            self.timer.daemon = True
            self.timer.start()

    def stop(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def bark(self):
        import faulthandler
        err = sys.stderr
        err.write("[bench watchdog r%d] no progress for %d s after '%s'\n" % (self.rank, self.seconds, self.where))
        faulthandler.dump_traceback(file=err, all_threads=True)
        err.write("[bench watchdog r%d] %s\n" % (self.rank, hardware_queue_note()))
        # the stream report calls into the HIP runtime, which may itself wait for a lock the stuck call holds (seen: hipStreamQuery beside a
        # hipStreamSynchronize that never returns): ask from a helper thread and give it five seconds — the exit below does not depend on it
        import threading

        def report():
            try:
                import ctypes
                lib = sys.modules["orc_amd._lib"].lib()
                buf = ctypes.create_string_buffer(8192)
                busy = lib.orc_debug_stream_report(buf, len(buf))
                err.write("[bench watchdog r%d] library streams (%d busy):\n%s" % (self.rank, busy, buf.value.decode()))
            except Exception as e:  # the library may not be loaded yet
                err.write("[bench watchdog r%d] no stream report (%r)\n" % (self.rank, e))

        th = threading.Thread(target=report, daemon=True)
        th.start()
        th.join(5.0)
        if th.is_alive():
            err.write("[bench watchdog r%d] the stream report itself did not return within 5 s (the HIP runtime is waiting too)\n" % self.rank)
        err.flush()
        os._exit(3)


def hardware_queue_note():
    """user-mode queues of this process as the kernel driver lists them (compute + SDMA), for the watchdog and the N > 1 result line"""
    d = "/sys/class/kfd/kfd/proc/%d/queues" % os.getpid()
    try:
        qs = os.listdir(d)
        kinds = {}
        for q in qs:
            try:
                t = open(os.path.join(d, q, "type")).read().strip()
            except OSError:
                t = "?"
            kinds[t] = kinds.get(t, 0) + 1
        return "KFD queues of pid %d: %d %s (GPU_MAX_HW_QUEUES=%s)" % (os.getpid(), len(qs), kinds, os.environ.get("GPU_MAX_HW_QUEUES", "unset: 4 per priority class"))
    except OSError as e:
        return "KFD queue list not readable (%s)" % e.__class__.__name__


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hex", choices=["hex", "config5"], help="hex: the synthetic hex channel (BASELINE configs[3], the headline); "
                    "config5: BASELINE configs[4], the mixed tet / hex / poly channel — every rank generates and owns nz block layers "
                    "(default 252 x 100 x 72 blocks = 5.14 M cells per GPU, 41 M on 8 GPUs)")
    ap.add_argument("--nx", type=int, default=None)
    ap.add_argument("--ny", type=int, default=None)
    ap.add_argument("--nz", type=int, default=None)
    ap.add_argument("--solver", default="multigrid", choices=["multigrid", "bicgstab", "jacobi", "multigrid_gs", "bicgstab_gs"])
    ap.add_argument("--momentum", default="umist", choices=["ud", "cd1", "quick", "umist"])
    ap.add_argument("--inner", type=int, default=50, help="matrix_solver.iterations (lib.rs:80)")
    ap.add_argument("--momentum-relaxation", type=float, default=0.1)
    ap.add_argument("--pressure-relaxation", type=float, default=0.001)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="N > 1: weak (default, what the driver's scaling run times) = every rank owns "
                    "an nx x ny x nz slab of an N times deeper channel; strong = the SAME nx x ny x nz mesh cut into N z-slabs (BASELINE configs[3]'s "
                    "'1 and 2x MI355X' on one ~10 M-cell mesh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--levels-csv", default=None, help="write the per-level product table (profiles/rNN_levels.csv)")
    ap.add_argument("--spin-up", type=int, default=2, help="untimed SIMPLE iterations before the snapshot every timed step restores "
                    "(the timed iteration is number spin-up + 1 of the run; its cost grows slowly with the state)")
    ap.add_argument("--dry-run", action="store_true", help="launcher rehearsal (CPU test of `--gpus N` without a launcher): the ranks join the "
                    "gloo control plane, agree on a number and rank 0 prints a line with value null; no GPU work")
    args = ap.parse_args()
    dflt = (400, 160, 160) if args.workload == "hex" else (252, 100, 72)
    args.nx, args.ny, args.nz = (v if v is not None else d for v, d in zip((args.nx, args.ny, args.nz), dflt))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` started like the N = 1 case: this process becomes the launcher.  Decided before anything here has
        # touched the GPU (no HIP call, no torch import); the ranks are fresh CHILD processes of torch.distributed.run, nothing is
        # re-executed in place.  Rank 0's JSON line is forwarded, the children's exit code is ours.
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d inside a torch.distributed.run of %d ranks: the two must agree" % (args.gpus, world))
    # N > 1: on by default (a launcher that is not ours — the driver's torch.distributed.run — gets it too); ORC_BENCH_WATCHDOG=0 switches it off
    wd = Watchdog(int(os.environ.get("ORC_BENCH_WATCHDOG", WATCHDOG_DEFAULT_S if world > 1 else 0)), rank)
    wd.kick("start")

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")  # control plane only; the data path uses RCCL inside liborc_amd
    if args.dry_run:
        time.sleep(float(os.environ.get("ORC_BENCH_DRY_RUN_SLEEP", "0")))  # (tests: a rank that hangs — for the watchdog and for the launcher's clean-up)
        seen = world
        if dist is not None:
            t = torch.tensor([1.0], dtype=torch.float64)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "SIMPLE iterations/s (dry run of the launcher: no GPU work)", "value": None, "n_gpus": world, "ranks_seen": seen,
                              "dry_run": True}), flush=True)
        return

    import orc_amd
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import MomentumDiscretization as MD
    from orc_amd.settings import NumericalSettings, SolutionMethod as SM
    from orc_amd.solver import Solver

    host_transport = os.environ.get("ORC_BENCH_HOST_TRANSPORT") == "1"  # rehearsal on a 1-GPU box: ranks share cuda:0
    # a launcher that narrows every rank's view to its own card (HIP_VISIBLE_DEVICES per rank) leaves one visible device: ordinal 0
    n_visible = max(1, int(orc_amd._lib.lib().orc_device_count()))
    orc_amd.init(0 if host_transport else local_rank % n_visible)
    if world > 1:
        from orc_amd import parallel
        if host_transport:
            parallel.init_host_transport(dist, rank, world)
        else:
            parallel.init_comm(dist, rank, world)

    solver_map = {"multigrid": SM.Multigrid, "bicgstab": SM.BiCGSTAB, "jacobi": SM.Jacobi, "multigrid_gs": SM.Multigrid_GS,
                  "bicgstab_gs": SM.BiCGSTAB_GS}
    mom_map = {"ud": MD.UD, "cd1": MD.CD1, "quick": MD.TVD_QUICK, "umist": MD.TVD_UMIST}
    # momentum/pressure relaxation 0.5/0.01 -> 0.1/0.001: with its default factors the reference algorithm itself diverges
    # on fine true-3-D meshes (CPU oracle and device alike); the factors only scale the correction
    # step (solver.rs:1186,1221-1223), so the work per SIMPLE iteration is unchanged.
    settings_kw = dict(momentum=mom_map[args.momentum], solver_type=solver_map[args.solver], iterations=args.inner,
                       momentum_relaxation=args.momentum_relaxation, pressure_relaxation=args.pressure_relaxation)
    settings = NumericalSettings.default(**settings_kw)

    nx, ny, nz = args.nx, args.ny, args.nz
    strong = args.scaling == "strong" and world > 1
    if strong:
        # the same nx x ny x nz mesh on every N: rank r owns layers [r nz / N, (r + 1) nz / N) (+ ghost layers); the generators need whole (config 5:
        # even) layer counts per rank
        if nz % world or (args.workload == "config5" and (nz // world) % 2):
            raise SystemExit("bench.py --scaling strong: nz = %d does not cut into %d %sslabs" % (nz, world, "even " if args.workload == "config5" else ""))
        nz = nz // world
    t_setup = time.perf_counter()
    mixed_facts = None
    if args.workload == "config5":
        from orc_amd import parallel
        solver, mesh, n_cells_total, nnz, mixed_facts = parallel.make_mixed_slab_solver(nx, ny, nz, rank, world, settings, initial_fields, dist)
    elif world == 1:
        a = set_channel_bcs(hex_channel(nx, ny, nz))
        mesh = Mesh(a)
        u, v, w, p = initial_fields(np.asarray(a["cell_centroid"]))
        solver = Solver(mesh, settings, 1000.0, 1e-3)
        solver.set_fields(u, v, w, p)
        n_cells_total = mesh.n_cells
        nnz = mesh.nnz
    else:
        from orc_amd import parallel
        solver, mesh, n_cells_total, nnz = parallel.make_slab_solver(nx, ny, nz, rank, world, settings, initial_fields, global_noise=strong)
    t_setup = time.perf_counter() - t_setup
    wd.kick("mesh and solver set-up")

    def barrier_sync():
        orc_amd._lib.check(orc_amd._lib.lib().orc_synchronize())
        if dist is not None:
            dist.barrier()
        # the kernels run on liborc_amd's own stream (synchronised above); torch is only loaded for N > 1
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()

    # The reference algorithm (fixed-count, unguarded inner solves; no implicit momentum under-relaxation) does not
    # converge on a mesh this fine: the CPU oracle and the device diverge alike, and the work of an
    # iteration DOES depend on the field state (coarse AMG operators get denser as the fields degrade).  So that every
    # timed step does identical work, the benchmark times ONE iteration of the trajectory repeatedly: two untimed
    # spin-up iterations from the seeded fields, a device-side snapshot of the state (u, v, w, p, momentum diagonals),
    # then every warm-up and timed step = restore the snapshot (device-to-device, 0.57 GB, inside the timed region)
    # + one full SIMPLE iteration (iteration 3 of the run).  Nothing is skipped or cached between steps: assembly,
    # the four hierarchy set-ups and all solves run in full each time.
    SPIN_UP = max(args.spin_up, 0)
    trajectory = []  # report of every untimed iteration: mean u, v, w, velocity- and pressure-correction norms (solver.rs:206-216)
    for _ in range(SPIN_UP):
        st_, rep_ = solver.iterate(1, report=True, raise_on_error=False)
        if st_ != 0:
            raise SystemExit("bench.py: spin-up iteration failed with status %d" % st_)
        trajectory.append([float(rep_[0][k]) for k in (0, 1, 2, 6, 7)])
        wd.kick("spin-up iteration %d" % (_ + 1))
    solver.snapshot()
    from orc_amd.linear_algebra import breakdown_guard_events
    guard_before = breakdown_guard_events()
    step_ms = []

    def run(k, record):
        last, status = None, 0
        for _ in range(k):
            ts = time.perf_counter()
            solver.restore()
            st_, rep_ = solver.iterate(1, report=True, raise_on_error=False)
            last = rep_[0]
            if record:
                step_ms.append((time.perf_counter() - ts) * 1e3)  # iterate() ends with a host sync
            wd.kick("%s step %d" % ("timed" if record else "warm-up", _ + 1))
            if st_ != 0:
                status = st_
                break
        return status, last

    if args.warmup > 0:
        run(args.warmup, False)
    coll_before = 0
    if dist is not None:
        import ctypes
        orc_amd._lib.lib().orc_debug_collectives.restype = ctypes.c_longlong
        coll_before = int(orc_amd._lib.lib().orc_debug_collectives(0))
    barrier_sync()
    t0 = time.perf_counter()
    st, last_rep = run(args.steps, True)
    dt_local = time.perf_counter() - t0  # this rank's own view (the line's ms_per_step is the slowest rank's, barriers included)
    barrier_sync()
    dt = time.perf_counter() - t0
    rep = np.array([last_rep]) if last_rep is not None else None
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    wd.kick("timed region")
    multi = None
    if dist is not None:
        # what a first multi-GPU run must be able to say about itself (VERDICT r04 #6b): did N ranks run on N cards, how evenly, through how many
        # collectives, with how large a halo — gathered over the control plane, reported by rank 0
        import ctypes
        L = orc_amd._lib.lib()
        L.orc_debug_collectives.restype = ctypes.c_longlong
        buf = ctypes.create_string_buffer(256)
        dev_info = buf.value.decode() if L.orc_device_info(buf, len(buf)) == 0 else "?"
        fb, tb = ctypes.c_int64(0), ctypes.c_int64(0)
        orc_amd._lib.check(L.orc_device_memory(ctypes.byref(fb), ctypes.byref(tb)))
        halo = getattr(mesh, "halo", None) or {}
        sp_ = [int(x) for x in halo.get("send_ptr", [0])]
        mine = {"rank": rank, "device": dev_info, "hbm_used_gb": round((tb.value - fb.value) / 1e9, 1), "ms_per_step": round(dt_local / args.steps * 1e3, 2),
                "owned_cells": int(getattr(mesh, "n_owned", mesh.n_cells)), "ghost_cells": int(mesh.n_cells - getattr(mesh, "n_owned", mesh.n_cells)),
                "peers": [int(q) for q in halo.get("peers", [])],
                "halo_bytes_per_field_and_exchange": [8 * (b - a_) for a_, b in zip(sp_[:-1], sp_[1:])],
                "collectives_in_timed_region": int(L.orc_debug_collectives(0)) - coll_before, "hardware_queues": hardware_queue_note()}
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        ms_all = [r_["ms_per_step"] for r_ in allr]
        multi = {"ranks": allr, "distinct_devices": len({r_["device"] for r_ in allr}),
                 "ms_per_step_min_max_over_ranks": [min(ms_all), max(ms_all)],
                 "collectives_per_iteration": allr[0]["collectives_in_timed_region"] / float(max(args.steps, 1)),
                 "collectives_definition": "halo exchanges + all-reduces (status agreements included) the library issued per SIMPLE iteration on rank 0 "
                                           "(orc_debug_collectives), timed steps only",
                 "watchdog_s": wd.seconds}
    guard_events_timed = breakdown_guard_events() - guard_before  # solves of the warm-up + timed steps the breakdown guard froze
    # device memory in use by the timed run (before the measurement legs below build a hierarchy of their own)
    import ctypes
    free_b, total_b = ctypes.c_int64(0), ctypes.c_int64(0)
    orc_amd._lib.check(orc_amd._lib.lib().orc_device_memory(ctypes.byref(free_b), ctypes.byref(total_b)))
    hbm_used_gb = (total_b.value - free_b.value) / 1e9

    # dominant kernel: CSR(SELL-64) SpMV inside BiCGSTAB, timed live with HIP events on the library stream
    solver.restore()  # the matrices of the timed iteration's state
    solver.assemble_momentum_only()
    spmv_ms, _ = solver.bench_spmv(args.spmv_reps)  # the plain product y = A x (no scalings, no epilogue): secondary figure
    n_local = getattr(mesh, "n_owned", mesh.n_cells)  # rows of this rank's matrices
    nnz_local = mesh.nnz
    spmv_bytes = 12.0 * nnz_local + 20.0 * n_local  # f64 value + i32 column per nnz; row_ptr, x, y per row
    # the products INSIDE the BiCGSTAB loop, as the solver launches them on level 0 (two Jacobi scalings, reduction epilogues)
    inloop = solver.bench_inloop_products(args.spmv_reps)
    inloop_ms = 0.5 * (inloop[0] + inloop[1])  # one of each per BiCGSTAB iteration (linear_algebra.rs:256, :260)
    achieved = spmv_bytes / (inloop_ms * 1e-3) / 1e9
    triple_ms = 0.5 * (inloop[2] + inloop[3]) if inloop[2] > 0 else None
    bicg_ms = solver.bench_bicgstab_iteration(10)
    # per level of the momentum system's Multigrid hierarchy: rows, nnz, padded entries, product time, roofline fraction
    levels = []
    if args.solver in ("multigrid", "multigrid_gs") and world == 1:
        solver.restore()
        for lvl, (rows_l, nnz_l, padded_l, ms_l) in enumerate(solver.bench_amg_levels(20)):
            # ALGORITHMIC bytes per product: value + 4-byte column per stored entry, row length + x + y per row.  Every level's
            # smoothing solve materialises its Jacobi-scaled values (orc_bench_amg_levels does the same), so no product reads a scaling
            # vector (r03 added 8 n on levels 2-3 by mistake); the launches really stream 2-byte columns / window positions,
            # so `frac_of_peak` is algorithmic throughput, not traffic — profiles/r04_pmc_products.csv has the measured bytes per level.
            bytes_l = 12.0 * nnz_l + 20.0 * rows_l
            levels.append({"level": lvl, "rows": rows_l, "nnz": nnz_l, "padded": padded_l, "us_per_product": ms_l * 1e3,
                           "algorithmic_bytes": bytes_l, "GBs": bytes_l / (ms_l * 1e-3) / 1e9,
                           "frac_of_peak": bytes_l / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBS})
        if args.levels_csv and rank == 0:
            with open(args.levels_csv, "w") as fh:
                fh.write("level,rows,nnz,padded_entries,padding_ratio,us_per_product,algorithmic_bytes_12nnz_20n,algorithmic_GB_per_s,algorithmic_frac_of_8TBs\n")
                for L in levels:
                    fh.write("%d,%d,%d,%d,%.4f,%.2f,%.0f,%.1f,%.4f\n" % (L["level"], L["rows"], L["nnz"], L["padded"], L["padded"] / max(L["nnz"], 1),
                                                                        L["us_per_product"], L["algorithmic_bytes"], L["GBs"], L["frac_of_peak"]))
    gs = None
    if args.solver in ("bicgstab_gs", "multigrid_gs") and world == 1:
        # BASELINE configs[2]: the time-dominant kernel is the multicolour Gauss-Seidel sweep (preconditioner application): one sweep =
        # n_colors launches of gs_color_sorted_k and moves one SpMV's bytes + the right-hand side
        gs_ms, gs_colors = solver.bench_gs_sweep(args.spmv_reps)
        gs_bytes = spmv_bytes + 8.0 * n_local
        ms1, ms3, _ = solver.bench_gs_sweep0(args.spmv_reps)  # [r04] what the slot-space solver launches: from zero, no fill, u/v/w per launch
        gs = {"kernel": "gsx_sweep0_k<3> (u, v, w per colour launch; gsx_sweep0_k<1> for the p' system)", "launches_per_sweep": gs_colors,
              "three_systems": {"avg_sweep_ms": ms3, "algorithmic_bytes_per_sweep": 3.0 * gs_bytes, "achieved": 3.0 * gs_bytes / (ms3 * 1e-3) / 1e9,
                                "frac": 3.0 * gs_bytes / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS},
              "one_system": {"avg_sweep_ms": ms1, "algorithmic_bytes_per_sweep": gs_bytes, "achieved": gs_bytes / (ms1 * 1e-3) / 1e9,
                             "frac": gs_bytes / (ms1 * 1e-3) / 1e9 / HBM_PEAK_GBS},
              "r03_kernel_gs_color_sorted_k": {"avg_sweep_ms": gs_ms, "frac": gs_bytes / (gs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
              "avg_sweep_ms": ms3, "avg_launch_ms": ms3 / max(gs_colors, 1),
              "algorithmic_bytes_per_sweep": 3.0 * gs_bytes, "achieved": 3.0 * gs_bytes / (ms3 * 1e-3) / 1e9, "frac": 3.0 * gs_bytes / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
              "note": "one sweep = 12 nnz + 20 n (SURVEY 8d) + 8 n (the right-hand side); at ~1 M cells the matrix (62 MB) lives in the 256 MiB "
                      "Infinity Cache and a launch covers 1/n_colors of the rows: these launches are latency-, not HBM-bound"}
    if gs is not None and args.solver == "multigrid_gs":
        # the smoother's sweeps start from the current iterate, not from zero: gs_color_sorted_k on every level is what this solver launches
        gs.update({"kernel": "gs_color_sorted_k (level 0: the smoother's general sweep; gsx_sweep0_k belongs to the bicgstab_gs solver)",
                   "avg_sweep_ms": gs_ms, "avg_launch_ms": gs_ms / max(gs_colors, 1), "algorithmic_bytes_per_sweep": gs_bytes,
                   "achieved": gs_bytes / (gs_ms * 1e-3) / 1e9, "frac": gs_bytes / (gs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    nz_total = args.nz if strong else nz * world  # layers of the whole mesh
    key = (args.nx, args.ny, args.nz, args.momentum, args.solver)
    workload_name = ("BASELINE configs[3]" if key == (400, 160, 160, "umist", "multigrid")
                     else "BASELINE configs[3] as its text reads (AMG V-cycle with a GS smoother: an extension, ORC's smoother is BiCGSTAB)" if key == (400, 160, 160, "umist", "multigrid_gs")
                     else "BASELINE configs[2]" if key == (512, 2016, 1, "quick", "bicgstab_gs") else "custom")
    mixed = args.workload == "config5"
    if mixed:
        workload_name = "BASELINE configs[4]" if (key[:3] == (252, 100, 72) and args.solver == "multigrid") else "custom (BASELINE configs[4] family)"
    if strong:
        workload_name += " on %dx MI355X (strong scaling: ONE %dx%dx%d mesh cut into %d z-slabs)" % (world, nx, ny, nz_total, world)
    bicg_bytes = 2.0 * spmv_bytes + 104.0 * n_local
    # Memory-side traffic per launch from the PMC counters (TCC_EA0_RDREQ by request size + TCC_EA0_WRREQ, separate rocprofv3
    # --pmc passes: scripts/gpu_pmc.sh; equal to FETCH_SIZE x 2 + WRITE_SIZE of MI355X_MICROARCH.md §HBM for this
    # kernel): bench.py cannot collect counters itself, so it quotes the committed profile of the same kernel on the same
    # matrix when one exists, else null.
    variant = getattr(solver, "inloop_variant", "false, true, false")  # "<narrow columns>, <scaled on the fly>, <non-temporal matrix loads>" as launched
    KERNEL = "spmv_uniform_k<EpiStoreSum, false, true, %s> / <EpiTs, false, true, %s>" % (variant, variant)
    traffic, traffic_source = None, None
    try:
        pmc_file = PMC_FILE_CONFIG5 if mixed else PMC_FILE
        pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
        # the counters must be those of THIS kernel on THIS matrix: anything else is not quoted
        if pmc["n"] == n_local and pmc["nnz"] == nnz_local and pmc["kernel"] == KERNEL:
            traffic = pmc["hbm_bytes_per_launch"]
            traffic_source = "profiles/%s (%s)" % (pmc_file, pmc.get("method", "rocprofv3 --pmc"))
    except Exception:
        traffic = None

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        units = n_cells_total / float(REF_CELLS)  # 10.24M-cell SIMPLE iterations per step (N at weak scaling)
        if mixed:
            units = 1.0 if strong else float(world)  # one slab of nz block layers per rank: slab-iterations per step (strong: the one mesh)
        out = {
            "metric": ("SIMPLE iterations/s (10.24M-cell hex channel equivalents; = iterations/s at N=1)" if not mixed else
                       ("SIMPLE iterations/s of ONE %dx%dx%d-block mixed tet/hex/poly mesh cut into N slabs" % (nx, ny, nz_total)) if strong else
                       "SIMPLE iterations/s (mixed tet/hex/poly slab equivalents: N x iterations/s, one %dx%dx%d-block slab per GPU)" % (nx, ny, nz)),
            "value": units * args.steps / dt,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "scaling_definition": ("strong: the same %dx%dx%d mesh on every N, cut along z into N slabs of %d layers (one ghost layer per inner side%s); value = "
                                   "SIMPLE iterations per second of that one mesh%s" % (nx, ny, nz_total, nz, ", two for the mixed mesh" if mixed else "",
                                                                                        " in 10.24M-cell equivalents" if not mixed else "")) if strong else
                                  ("weak: every rank owns a %dx%dx%d slab of a %dx%dx(%d*N) channel cut along z; value = N x (10.24M-cell SIMPLE "
                                   "iterations per second), i.e. cells processed per second / 10.24M" % (nx, ny, nz, nx, ny, nz)) if not mixed else
                                  ("weak: every rank generates and owns %d block layers of a %dx%dx(%d*N)-block mixed channel cut along z (cells by centroid; "
                                   "two generated ghost layers per inner side, no process holds the whole mesh); value = N x SIMPLE iterations per second"
                                   % (nz, nx, ny, nz)),
            "step_definition": "restore the device-side snapshot taken after %d spin-up iterations (0.57 GB device-to-device, inside the "
                               "timed region) + one full SIMPLE iteration; every step does identical work" % SPIN_UP,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "status": int(st),
            "step_ms": [round(x, 2) for x in step_ms],
            "config": {
                "workload": "%s: synthetic %s %dx%dx%d per GPU (%d cells total), TVD-%s momentum, "
                            "Rhie-Chow + SecondOrder, solver=%s (%d inner iterations) + Jacobi preconditioner, relaxation u %.3g / p %.3g, "
                            "full SIMPLE iteration" % (workload_name, "mixed tet / pyramid / prism / hex / polyhedral channel, blocks" if mixed else "hex channel",
                                                       nx, ny, nz, n_cells_total, args.momentum.upper(), args.solver, args.inner,
                                                       args.momentum_relaxation, args.pressure_relaxation),
                "mixed_mesh": mixed_facts,
                "cells_total": int(n_cells_total),
                "parallelism": ("cell slabs x%d, halo exchange + all-reduce over %s" % (world, "the host-staged debug transport (gloo; ranks share one GPU: a "
                                "rehearsal, not a measurement)" if host_transport else "RCCL (ncclSend/ncclRecv + ncclAllReduce over xGMI)")) if world > 1 else "single GPU",
                "transport": ("host" if host_transport else "rccl") if world > 1 else None,
                "momentum_solves": "u, v, w in lock-step on their shared pattern (one column stream, interleaved vectors)" if (args.solver in ("multigrid", "bicgstab", "bicgstab_gs") and os.environ.get("ORC_TRIPLE_MOMENTUM", "1") != "0" and (world == 1 or args.solver != "bicgstab_gs")) else "one system per solve",
                "setup_s": round(t_setup, 2),
                "hbm_used_gb": round(hbm_used_gb, 1),
            },
            "roofline": {
                "kernel": KERNEL,
                "kernel_role": "SELL-64 CSR SpMV INSIDE the BiCGSTAB loop on level 0 (a_u of the momentum system, values scaled by the arm's Jacobi "
                               "preconditioner and the smoother's nested one — materialised once per smoothing solve — with the reduction epilogues: "
                               "nu = A p + sum(nu), t = A s + t.s, t.t), one system per launch — what the p' solve and every one-system solve launch; "
                               "HIP events on the library stream around %d launches each" % args.spmv_reps,
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "avg_launch_ms": inloop_ms,
                "avg_launch_ms_by_epilogue": {"EpiStoreSum": inloop[0], "EpiTs": inloop[1]},
                "algorithmic_bytes_per_launch": spmv_bytes,
                "algorithmic_bytes_definition": "SURVEY 8(d): 12 nnz + 20 n per system (f64 value + i32 column per entry; row length, x, y per row); the "
                                                "epilogue's operand (8 n, EpiTs) the in-loop kernel also reads is NOT counted; with the narrow column image "
                                                "(template argument 4 = true) the launch streams 2-byte columns, so `frac` is algorithmic throughput",
                "three_systems_per_launch": None if triple_ms is None else {
                    "kernel": "spmv3_uniform_k<EpiStoreSum3, 4, true, %s> / <EpiTs3, 4, true, %s>" % (variant, variant),
                    "avg_launch_ms": triple_ms,
                    "avg_launch_ms_by_epilogue": {"EpiStoreSum3": inloop[2], "EpiTs3": inloop[3]},
                    "algorithmic_bytes_per_launch": 3.0 * spmv_bytes,
                    "achieved": 3.0 * spmv_bytes / (triple_ms * 1e-3) / 1e9,
                    "frac": 3.0 * spmv_bytes / (triple_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "u, v, w momentum products in one launch: 3 SpMV units of SURVEY 8(d); the launch itself moves 28 nnz + 60 n bytes (one "
                            "column stream for three value streams), so `frac` is algorithmic throughput, not traffic",
                },
                "plain_product": {"kernel": "spmv_uniform_k<EpiStore, false, true, %s, false>" % variant.split(",")[0], "avg_launch_ms": spmv_ms,
                                  "frac": spmv_bytes / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "bicgstab_iteration_ms": bicg_ms,
                "bicgstab_iteration_GBs": bicg_bytes / (bicg_ms * 1e-3) / 1e9,
                "gauss_seidel_sweep": gs,
            },
            "amg_levels": levels,
            "multi_gpu": multi,
            "report_last": [float(x) for x in rep[-1]] if rep is not None and len(rep) else None,
            "report_trajectory": {"columns": ["u_mean", "v_mean", "w_mean", "velocity_correction_norm", "pressure_correction_norm"],
                                  "spin_up": trajectory,
                                  "timed": [float(rep[-1][k]) for k in (0, 1, 2, 6, 7)] if rep is not None and len(rep) else None},
            "breakdown_guard_events_in_timed_region": int(guard_events_timed),
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(settings_kw, iters=3 if mixed else 6, mixed_ref_cells=(n_cells_total // world) if mixed else None)
        print(json.dumps(out), flush=True)
    wd.stop()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
