#!/usr/bin/env python3
"""The device at the BENCHMARK'S OWN SIZE in the reference's own mode.

bench.py's workload — 400 x 160 x 160 hex channel, TVD-UMIST, Rhie-Chow, SecondOrder, Multigrid arm with 50 BiCGSTAB iterations per
level + Jacobi preconditioner, relaxation 0.1 / 0.001, bench.initial_fields — run with frozen_diagonals = 0 (in-place Rhie-Chow
diagonals, discretization.rs:182-197), breakdown_guard = 0 (linear_algebra.rs:255-268 has none) and reduction_order = 1 (every dot
product in nalgebra's dotx association): the mode in which the device reproduces the oracle bit for bit at test sizes
(tests/test_gpu_reference_order.py, test_gpu_bench_family.py).  Its per-iteration report (mean velocities, mean Peclet number,
velocity- and pressure-correction norms: solver.rs:206-216) is compared, double by double, with what the oracle — one CPU core, 44
minutes — wrote for the same run into profiles/r03_oracle_trajectory_400x160x160_inplace.json.  Identical bits in all six doubles of
an iteration mean that 4 x 10.24 M field values went through ~2 800 products and ~11 000 dot products per iteration on code paths
that only exist at this size (window fallbacks, 2 048-workgroup grids, non-temporal streams) and came out the same.

    python scripts/reference_mode_fullsize.py --iterations 3 --out profiles/r04_reference_mode_400x160x160.json
"""
import argparse
import json
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=400); ap.add_argument("--ny", type=int, default=160); ap.add_argument("--nz", type=int, default=160)
ap.add_argument("--iterations", type=int, default=3)
ap.add_argument("--workload", default="hex", choices=["hex", "config5"], help="config5: the mixed tet / hex / poly slab of bench.py --workload config5 "
                "(--nx 252 --ny 100 --nz 72 --oracle profiles/r04_oracle_trajectory_config5_252x100x72_inplace.json)")
ap.add_argument("--oracle", default=os.path.join(ROOT, "profiles", "r03_oracle_trajectory_400x160x160_inplace.json"))
ap.add_argument("--out", default=None)
args = ap.parse_args()

import bench
import orc_amd
from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
from orc_amd.settings import NumericalSettings
from orc_amd.solver import Solver

orc_amd.init(0)
ref = json.load(open(args.oracle))
assert ref["shape"] == [args.nx, args.ny, args.nz], "the oracle file is for another mesh"
kw = {k: v for k, v in ref["settings"].items()}
assert kw["frozen_diagonals"] == 0 and kw["breakdown_guard"] == 0
if args.workload == "config5":
    from orc_amd import parallel
    from orc_amd.mesh import set_mixed_channel_bcs
    assert ref.get("workload") == "config5", "the oracle file is for another workload"
    _a, _h, _g, a = parallel.mixed_slab_arrays(args.nx, args.ny, args.nz, 0, 1)
    set_mixed_channel_bcs(a)
else:
    a = set_channel_bcs(hex_channel(args.nx, args.ny, args.nz))
mesh = Mesh(a)
f0 = bench.initial_fields(np.asarray(a["cell_centroid"]))
del a
s = Solver(mesh, NumericalSettings.default(reduction_order=1, **kw), 1000.0, 1e-3)
s.set_fields(*f0)
bits = lambda x: struct.unpack("<Q", struct.pack("<d", float(x)))[0]
rows, all_same = [], True
for it in range(args.iterations):
    t0 = time.perf_counter()
    st, rep = s.iterate(1, report=True, raise_on_error=False)
    dt = time.perf_counter() - t0
    dev = [float(rep[0][k]) for k in (0, 1, 2, 3, 6, 7)]  # means, mean Peclet, velocity- and pressure-correction norms
    orc = ref["report"][it] if it < len(ref["report"]) else None
    row = {"iteration": it + 1, "status": int(st), "seconds": round(dt, 2), "device": dev, "device_hex": ["%016x" % bits(x) for x in dev]}
    if orc is not None:
        same = [bits(d) == bits(o) for d, o in zip(dev, orc)]
        row.update(oracle=orc, identical_bits=same, rel_diff=[abs(d - o) / max(abs(o), 1e-300) for d, o in zip(dev, orc)])
        all_same = all_same and all(same)
    rows.append(row)
    print(json.dumps(row), flush=True)
    if st != 0:
        break
out = {"what": "device in the reference's own mode (in-place diagonals, no guard, nalgebra reduction order) against the oracle's committed trajectory",
       "shape": [args.nx, args.ny, args.nz], "settings": kw, "oracle_file": os.path.relpath(args.oracle, ROOT),
       "report_columns": ref["report_columns"], "iterations": rows, "all_identical": bool(all_same)}
print("ALL IDENTICAL" if all_same else "DIFFERENCES (see identical_bits per iteration)")
if args.out:
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
