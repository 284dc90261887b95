"""debug: which solve of the 3x3 cube / UMIST / Multigrid iteration differs from the oracle in reference order"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scipy.sparse as sp
import helpers as H
import orc_amd
from oracle import pyoracle as po
from orc_amd.mesh import Mesh, MeshArrays
from orc_amd.settings import NumericalSettings
from orc_amd.solver import Solver
from orc_amd.linear_algebra import iterative_solve, set_breakdown_guard, set_reduction_order

orc_amd.init(0)
om = po.Mesh.read(os.path.join(ROOT, "tests/golden/meshes/3x3_cube.msh"))
H.cube_bcs_mixed(om)
a = MeshArrays(om.arrays())
dm = Mesh(a)
kw = dict(momentum=5, solver_type=2, frozen_diagonals=1, breakdown_guard=0)
u, v, w, p = H.seeded_fields(a, seed=5, scale_u=4e-4)
s = Solver(dm, NumericalSettings.default(reduction_order=1, **kw), 1000.0, 1e-3)
s.set_fields(u, v, w, p)
au, av, aw, bu, bv, bw, pe = s.assemble_momentum()
rp, ci = dm.matrix_pattern()
set_reduction_order(1); set_breakdown_guard(False)
for name, vals, b, x0 in (("u", au, bu, u), ("v", av, bv, v), ("w", aw, bw, w)):
    A = sp.csr_matrix((vals, ci, rp), shape=(len(b), len(b)))
    print(name, "matrix finite:", np.isfinite(vals).all(), "b finite:", np.isfinite(b).all(), "min/max offdiag", vals.min(), vals.max())
    for method, mname in ((3, "bicgstab"), (2, "multigrid")):
        for it in (5, 20, 30, 40, 50):
            x, xo = x0.copy(), x0.copy()
            st = iterative_solve(A, b, x, it, method, 0.5, 1e-3, 1, raise_on_error=False)
            sto = po.iterative_solve(po.Csr.from_scipy(A), b, xo, it, method, 0.5, 1e-3, 1)
            nn, nno = int(np.isnan(x).sum()), int(np.isnan(xo).sum())
            same = np.array_equal(x.view(np.uint64), xo.view(np.uint64))
            print("  %s %s it=%d st=%d/%d nan=%d/%d same=%s maxdiff=%.3e" % (name, mname, it, st, sto, nn, nno, same, np.nanmax(np.abs(x - xo)) if nn < len(x) and nno < len(x) else float('nan')))
