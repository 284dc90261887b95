#!/usr/bin/env python3
"""Per-level product timings of the momentum system's Multigrid hierarchy under the product-kernel variants
(orc_debug_set_spmv_variant): where the coarse levels lose their bandwidth."""
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
ap.add_argument("--variants", default="0,1,2,4,5")
args = ap.parse_args()
orc_amd.init(0)
a = set_channel_bcs(hex_channel(args.nx, args.ny, args.nz))
mesh = Mesh(a)
u, v, w, p = bench.initial_fields(np.asarray(a["cell_centroid"]))
s = Solver(mesh, NumericalSettings.default(momentum=5, momentum_relaxation=0.1, pressure_relaxation=0.001), 1000.0, 1e-3)
s.set_fields(u, v, w, p)
s.iterate(2)
L = orc_amd._lib.lib()
names = {0: "production", 1: "padded", 2: "padded-predicated", 3: "packed", 4: "packed-nogather", 5: "padded-nogather", 8: "r01-kernel", 9: "r01-kernel-ragged", 6: "padded-guarded", 7: "predicated-guarded",
         20: "lds-window", 21: "lds-no-window-load", 22: "lds-window-load-only", 10: "pipelined", 11: "pipelined-padded", 12: "pipelined-predicated"}
for var in [int(x) for x in args.variants.split(",")]:
    L.orc_debug_set_spmv_variant(ctypes.c_int(var))
    for lvl, (rows, nnz, padded, ms) in enumerate(s.bench_amg_levels(20)):
        b = 12.0 * nnz + 20.0 * rows + 8.0 * rows * (2 if lvl == 0 else 1)
        print("variant %-18s level %d rows %9d nnz %9d padded %9d  %7.1f us  %6.0f GB/s  %.3f of peak" % (names.get(var, var), lvl, rows, nnz, padded, ms * 1e3, b / ms / 1e6, b / ms / 1e6 / 8000), flush=True)
L.orc_debug_set_spmv_variant(ctypes.c_int(0))
