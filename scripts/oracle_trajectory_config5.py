#!/usr/bin/env python3
"""The CPU oracle on BASELINE configs[4]'s per-GPU slab — the mixed tet / pyramid / prism / hex / polyhedral channel of bench.py --workload config5
(252 x 100 x 72 blocks = 5.14 M cells by default), same BCs, settings and initial fields — in the reference's own mode: per-iteration report as JSON,
for scripts/reference_mode_fullsize.py --workload config5 to compare the device with, double by double.  One core, ~4 minutes per iteration at the
full size.  Test infrastructure: nothing in the product path imports this."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import pyoracle as po
from orc_amd import parallel
from orc_amd.mesh import set_mixed_channel_bcs

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=252); ap.add_argument("--ny", type=int, default=100); ap.add_argument("--nz", type=int, default=72)
ap.add_argument("--iterations", type=int, default=3)
ap.add_argument("--out", default=None)
args = ap.parse_args()
t0 = time.perf_counter()
_a, _h, _g, a = parallel.mixed_slab_arrays(args.nx, args.ny, args.nz, 0, 1)
set_mixed_channel_bcs(a)
print("mesh: %d cells in %.1f s" % (a.n_cells, time.perf_counter() - t0), flush=True)
om = po.Mesh.from_arrays(a)
u, v, w, p = bench.initial_fields(np.asarray(a["cell_centroid"]))
n = a.n_cells
del a, _a
kw = dict(momentum=5, solver_type=2, iterations=50, momentum_relaxation=0.1, pressure_relaxation=0.001, frozen_diagonals=0, breakdown_guard=0)
t0 = time.perf_counter()
st, rep = po.solve_steady(om, u, v, w, p, po.default_settings(**kw), 1000.0, 1e-3, args.iterations, report=True)
dt = time.perf_counter() - t0
out = {"workload": "config5", "shape": [args.nx, args.ny, args.nz], "cells": int(n), "settings": {k: (float(x) if isinstance(x, float) else int(x)) for k, x in kw.items()},
       "status": int(st), "status_string": po.status_string(st), "seconds": dt, "iterations": args.iterations,
       "report_columns": ["u_mean", "v_mean", "w_mean", "peclet_avg", "velocity_correction_norm", "pressure_correction_norm"],
       "report": rep.tolist(), "field_max_abs": [float(np.nanmax(np.abs(x))) for x in (u, v, w, p)], "field_nan": [int(np.isnan(x).sum()) for x in (u, v, w, p)]}
print(json.dumps(out), flush=True)
if args.out:
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
