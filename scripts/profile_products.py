#!/usr/bin/env python3
"""Counter-profiling target for the products of the timed workload (BASELINE configs[3], 400x160x160): one momentum assembly,
then exactly the launches bench.py times — the level-0 products as the BiCGSTAB loop launches them (one system and three
systems per launch: orc_bench_inloop_products), the plain product, and one product per level of a_u's Multigrid hierarchy
(orc_bench_amg_levels: levels 0-1 spmv_uniform_k, levels 2-3 spmv_xwin_k) — `--reps` launches each after one warm launch.
Run under `rocprofv3 --kernel-trace --stats` and, separately, one `--pmc` pass per counter group (scripts/gpu_pmc.sh).
`--workload config5`: the same for BASELINE configs[4]'s per-GPU slab (252 x 100 x 72 blocks of the mixed tet / hex / poly channel, 5.14 M cells)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orc_amd  # noqa: E402
from bench import initial_fields  # noqa: E402
from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs  # noqa: E402
from orc_amd.settings import NumericalSettings  # noqa: E402
from orc_amd.solver import Solver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=400); ap.add_argument("--ny", type=int, default=160); ap.add_argument("--nz", type=int, default=160)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--workload", default="hex", choices=["hex", "config5"])
args = ap.parse_args()
orc_amd.init(0)
settings = NumericalSettings.default(momentum=5, momentum_relaxation=0.1, pressure_relaxation=0.001)
if args.workload == "config5":
    from orc_amd import parallel
    dflt = (252, 100, 72)
    nx, ny, nz = (v if v != d0 else d for v, d0, d in zip((args.nx, args.ny, args.nz), (400, 160, 160), dflt))
    s, m, _n, _nnz, _facts = parallel.make_mixed_slab_solver(nx, ny, nz, 0, 1, settings, initial_fields)
else:
    a = set_channel_bcs(hex_channel(args.nx, args.ny, args.nz))
    m = Mesh(a)
    s = Solver(m, settings, 1000.0, 1e-3)
    s.set_fields(*initial_fields(np.asarray(a["cell_centroid"])))
s.assemble_momentum_only()
inloop = s.bench_inloop_products(args.reps)
plain, _ = s.bench_spmv(args.reps)
levels = s.bench_amg_levels(args.reps)
n, nnz = m.n_cells, m.nnz
B = 12.0 * nnz + 20.0 * n
print("cells %d nnz %d reps %d | in-loop one system %.1f / %.1f us (%.3f of 8 TB/s) | three systems %.1f / %.1f us (%.3f) | plain %.1f us (%.3f)"
      % (n, nnz, args.reps, inloop[0] * 1e3, inloop[1] * 1e3, B / (0.5 * (inloop[0] + inloop[1])) / 1e6 / 8000.0, inloop[2] * 1e3, inloop[3] * 1e3,
         3 * B / (0.5 * (inloop[2] + inloop[3]) + 1e-30) / 1e6 / 8000.0, plain * 1e3, B / plain / 1e6 / 8000.0))
for lvl, (rows, nz, padded, ms) in enumerate(levels):
    print("level %d rows %d nnz %d padded %d  %.1f us" % (lvl, rows, nz, padded, ms * 1e3))
