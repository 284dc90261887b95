#!/usr/bin/env python3
"""How compressible are the column indices of the SELL-64 images the products stream?  For every (slice, depth) the range
of the 64 lanes' columns: a range < 256 fits a per-(slice, depth) base + one byte per entry, < 65536 two bytes.
Level 0 = the momentum matrix of the hex channel, level 1 = its first coarse operator (orc_amg_coarsen)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import orc_amd
from orc_amd.linear_algebra import amg_coarsen
from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
from orc_amd.settings import NumericalSettings
from orc_amd.solver import Solver

nx, ny, nz = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (200, 80, 80)))
orc_amd.init(0)
a = set_channel_bcs(hex_channel(nx, ny, nz))
m = Mesh(a)
s = Solver(m, NumericalSettings.default(momentum=5, momentum_relaxation=0.1, pressure_relaxation=0.001), 1000.0, 1e-3)
s.set_fields(*bench.initial_fields(np.asarray(a["cell_centroid"])))
au = s.assemble_momentum()[0]
A = m.csr(au)
A.sort_indices()


def stats(A, name):
    n = A.shape[0]
    ip, ci = A.indptr, A.indices
    lens = np.diff(ip)
    ns = (n + 63) // 64
    tot = esc8 = esc16 = 0
    for sl in range(ns):
        lo, hi = sl * 64, min(sl * 64 + 64, n)
        w = lens[lo:hi].max()
        for k in range(w):
            rows = np.arange(lo, hi)[lens[lo:hi] > k]
            c = ci[ip[rows] + k]
            r = c.max() - c.min()
            tot += len(rows)
            if r >= 256: esc8 += len(rows)
            if r >= 65536: esc16 += len(rows)
    print("%s: n %d nnz %d  entries in (slice, depth) groups with range >= 256: %.1f %%, >= 65536: %.1f %%  -> %.2f B/entry with byte deltas + escapes" %
          (name, n, A.nnz, 100.0 * esc8 / tot, 100.0 * esc16 / tot, (1.0 * (tot - esc8) + 4.0 * esc8) / tot), flush=True)


stats(A, "level 0")
_, A1, _ = amg_coarsen(A)
A1 = A1.tocsr(); A1.sort_indices()
stats(A1, "level 1")
