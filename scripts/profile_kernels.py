#!/usr/bin/env python3
"""Kernel-level profiling target: C4 mesh, one momentum assembly (real a_u), then the SpMV and the BiCGSTAB iteration
body repeated a few times.  Run under `rocprofv3 --kernel-trace --stats` and, separately, `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import orc_amd  # noqa: E402
from bench import initial_fields  # noqa: E402
from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs  # noqa: E402
from orc_amd.settings import NumericalSettings  # noqa: E402
from orc_amd.solver import Solver  # noqa: E402

nx, ny, nz = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (400, 160, 160)))
orc_amd.init(0)
a = set_channel_bcs(hex_channel(nx, ny, nz))
m = Mesh(a)
s = Solver(m, NumericalSettings.default(momentum=5, momentum_relaxation=0.1, pressure_relaxation=0.001), 1000.0, 1e-3)
s.set_fields(*initial_fields(np.asarray(a["cell_centroid"])))
s.assemble_momentum()
s.assemble_pressure()
ms, _ = s.bench_spmv(20)
bi = s.bench_bicgstab_iteration(5)
n, nnz = m.n_cells, m.nnz
B = 12.0 * nnz + 20.0 * n
print("cells %d nnz %d | spmv %.4f ms -> %.1f GB/s (%.3f of 8 TB/s) | bicgstab iteration %.4f ms -> %.1f GB/s"
      % (n, nnz, ms, B / ms / 1e6, B / ms / 1e6 / 8000.0, bi, (2 * B + 104.0 * n) / bi / 1e6))
