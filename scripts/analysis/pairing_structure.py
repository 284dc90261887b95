"""CPU analysis (oracle side, no GPU): what does ORC's greedy pairing (linear_algebra.rs:12-63) look like on a bench-family momentum matrix,
and how far is the unconstrained arg-min (the device's starting state, agg_init_k) from it?  Prints, for a few grid lines, the partner offsets
of the arg-min state and of the fixed point, the share of rows that differ, and the lengths of the runs of consecutive differing rows."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from oracle import pyoracle as po
from orc_amd.mesh import hex_channel, set_channel_bcs
from bench import initial_fields

nx, ny, nz = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (100, 16, 8)))
a = set_channel_bcs(hex_channel(nx, ny, nz, lx=0.002 * nx / 400, ly=0.001 * ny / 160))
om = po.Mesh.from_arrays(a)
u, v, w, p = initial_fields(np.asarray(a["cell_centroid"]), lx=0.002 * nx / 400)
s = po.default_settings(momentum=5, solver_type=2, iterations=50, momentum_relaxation=0.1, pressure_relaxation=0.001, frozen_diagonals=0, breakdown_guard=0)
st, _ = po.solve_steady(om, u, v, w, p, s, 1000.0, 1e-3, 2)
print("spin-up status", st, "u mean", u.mean())
a_di, bu, bv, bw = po.build_momentum_diffusion_matrix(om, 1e-3)
au, av, aw = (po.initialize_momentum_matrix(om) for _ in range(3))
po.build_momentum_advection_matrices(au, av, aw, a_di, om, u, v, w, p, s, 1000.0)
m = au.to_scipy().tocsr()
n = m.shape[0]
dinv = 1.0 / m.diagonal()
m = (m.multiply(dinv[:, None])).tocsr()  # the arm sees the Jacobi-scaled system
indptr, indices, data = m.indptr, m.indices, m.data

def greedy(constrained=True):
    choice = np.full(n, -1, np.int64)
    taken = np.zeros(n, bool)
    for i in range(n):
        best, bj = np.finfo(float).max, -1
        for q in range(indptr[i], indptr[i + 1]):
            j = indices[q]
            if j == i or (constrained and taken[j]):
                continue
            if data[q] < best:
                best, bj = data[q], j
        choice[i] = bj
        if constrained and bj >= 0:
            taken[bj] = True
    return choice

arg = greedy(False)
fix = greedy(True)
diff = arg != fix
print("rows %d, differ %d (%.1f %%)" % (n, diff.sum(), 100.0 * diff.mean()))
off_arg, off_fix = arg - np.arange(n), fix - np.arange(n)
for name, off in (("arg-min", off_arg), ("fixed point", off_fix)):
    vals, cnt = np.unique(off, return_counts=True)
    print(name, "partner offsets:", dict(zip(vals.tolist(), cnt.tolist())))
# runs of consecutive differing rows (in index order)
runs = []
r = 0
for d in diff:
    if d: r += 1
    elif r: runs.append(r); r = 0
if r: runs.append(r)
runs = np.array(runs) if runs else np.zeros(1, int)
print("runs of consecutive differing rows: %d runs, mean %.1f, max %d" % (len(runs), runs.mean(), runs.max()))
for line in (0, 1, ny // 2, ny * nz // 2):
    lo = line * nx
    print("line %d arg-min :" % line, "".join("+" if o == 1 else "-" if o == -1 else "^" if o == nx else "v" if o == -nx else "?" for o in off_arg[lo:lo + nx]))
    print("line %d fixed   :" % line, "".join("+" if o == 1 else "-" if o == -1 else "^" if o == nx else "v" if o == -nx else "?" for o in off_fix[lo:lo + nx]))
