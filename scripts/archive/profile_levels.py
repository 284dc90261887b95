#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the per-level products (scripts/gpu_pmc_levels.sh): builds the bench configuration,
two spin-up iterations, then `reps` products per level of a_u's hierarchy under the given kernel variant."""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import orc_amd
from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
from orc_amd.settings import NumericalSettings
from orc_amd.solver import Solver

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=400); ap.add_argument("--ny", type=int, default=160); ap.add_argument("--nz", type=int, default=160)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--reps", type=int, default=4)
args = ap.parse_args()
orc_amd.init(0)
a = set_channel_bcs(hex_channel(args.nx, args.ny, args.nz))
mesh = Mesh(a)
u, v, w, p = bench.initial_fields(np.asarray(a["cell_centroid"]))
s = Solver(mesh, NumericalSettings.default(momentum=5, momentum_relaxation=0.1, pressure_relaxation=0.001, solver_type=3), 1000.0, 1e-3)
s.set_fields(u, v, w, p)
s.iterate(1)   # BiCGSTAB solver: a cheap iteration that leaves assembled momentum matrices behind
L = orc_amd._lib.lib()
L.orc_debug_set_spmv_variant(ctypes.c_int(args.variant))
for lvl, (rows, nnz, padded, ms) in enumerate(s.bench_amg_levels(args.reps)):
    print("level %d rows %d nnz %d padded %d %.1f us" % (lvl, rows, nnz, padded, ms * 1e3), flush=True)
