#!/usr/bin/env python3
"""The CPU oracle (oracle/: restatement of ORC's solve_steady, solver.rs:60-222) on the BENCH workload family — same
generator, BCs, settings and initial fields as bench.py — at any size, one process, one core: per-iteration report
(mean velocities, velocity- and pressure-correction norms) as JSON.  At the full 400x160x160 this is about ten minutes of
CPU per SIMPLE iteration and ~40 GB of host memory; its output is what bench.py's `report_trajectory` is compared with
(profiles/r03_oracle_trajectory_*.json).  Test infrastructure: nothing in the product path imports this."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import pyoracle as po
from orc_amd.mesh import hex_channel, set_channel_bcs

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=400); ap.add_argument("--ny", type=int, default=160); ap.add_argument("--nz", type=int, default=160)
ap.add_argument("--iterations", type=int, default=3)
ap.add_argument("--momentum-relaxation", type=float, default=0.1); ap.add_argument("--pressure-relaxation", type=float, default=0.001)
ap.add_argument("--frozen", type=int, default=0, help="0 = the reference's in-place diagonals (default), 1 = frozen (the product's mode)")
ap.add_argument("--out", default=None)
args = ap.parse_args()
a = set_channel_bcs(hex_channel(args.nx, args.ny, args.nz))
om = po.Mesh.from_arrays(a)
u, v, w, p = bench.initial_fields(np.asarray(a["cell_centroid"]))
del a
kw = dict(momentum=5, solver_type=2, iterations=50, momentum_relaxation=args.momentum_relaxation, pressure_relaxation=args.pressure_relaxation,
          frozen_diagonals=args.frozen, breakdown_guard=0)
t0 = time.perf_counter()
st, rep = po.solve_steady(om, u, v, w, p, po.default_settings(**kw), 1000.0, 1e-3, args.iterations, report=True)
dt = time.perf_counter() - t0
out = {"shape": [args.nx, args.ny, args.nz], "cells": args.nx * args.ny * args.nz, "settings": {k: (float(x) if isinstance(x, float) else int(x)) for k, x in kw.items()},
       "status": int(st), "status_string": po.status_string(st), "seconds": dt, "iterations": args.iterations,
       "report_columns": ["u_mean", "v_mean", "w_mean", "peclet_avg", "velocity_correction_norm", "pressure_correction_norm"],
       "report": rep.tolist(), "field_max_abs": [float(np.nanmax(np.abs(x))) for x in (u, v, w, p)],
       "field_nan": [int(np.isnan(x).sum()) for x in (u, v, w, p)]}
print(json.dumps(out), flush=True)
if args.out:
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
