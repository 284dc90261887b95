#!/usr/bin/env python3
"""How alike are the greedy pairings of the u, v and w momentum systems of one SIMPLE iteration (level 0 and level 1)?
Decides whether a sibling system's pairing is a useful starting point for the fixed-point iteration."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import orc_amd
from orc_amd.linear_algebra import amg_coarsen
from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
from orc_amd.settings import NumericalSettings
from orc_amd.solver import Solver

nx, ny, nz = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (400, 40, 40)))
orc_amd.init(0)
a = set_channel_bcs(hex_channel(nx, ny, nz))
m = Mesh(a)
s = Solver(m, NumericalSettings.default(momentum=5, momentum_relaxation=0.1, pressure_relaxation=0.001), 1000.0, 1e-3)
s.set_fields(*bench.initial_fields(np.asarray(a["cell_centroid"])))
s.iterate(2)
au, av, aw = s.assemble_momentum()[:3]
res = []
for name, vals in (("u", au), ("v", av), ("w", aw)):
    A = m.csr(vals); A.sort_indices()
    p0, A1, _ = amg_coarsen(A)
    p1, _, _ = amg_coarsen(A1)
    res.append((name, p0, p1))
for i in range(3):
    for j in range(i + 1, 3):
        print("%s vs %s: level 0 %.2f %% of the rows have the same partner, level 1 %.2f %%" %
              (res[i][0], res[j][0], 100.0 * np.mean(res[i][1] == res[j][1]), 100.0 * np.mean(res[i][2] == res[j][2])), flush=True)
