#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/, the restatement of the reference).

"Restatement-derived" golden vectors: the reference holds no output fixtures for this path and cannot
be built here, so these files pin the oracle (and, through the GPU tests, the HIP path) against silent changes.
Inputs are the reference's own mesh fixtures (tests/golden/meshes/) with the BC set-ups of its drivers.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CASES = {
    "3x3_cube": ("3x3_cube", H.cube_bcs, dict(momentum=po.TVD_UMIST)),
    "3x3_cube_mixed": ("3x3_cube", H.cube_bcs_mixed, dict(momentum=po.CD1)),
    "channel_flow": ("channel_flow", H.channel_bcs, dict(momentum=po.CD1)),
}


def main():
    for name, (mesh, bcs, kw) in CASES.items():
        om = po.Mesh.read(os.path.join(OUT, "meshes", mesh + ".msh"))
        bcs(om)
        a = om.arrays()
        u, v, w, p = H.seeded_fields(a, seed=11, w_zero=(mesh == "channel_flow"))
        rho, mu = 1000.0, 1e-3
        out = dict(u0=u, v0=v, w0=w, p0=p)
        for frozen in (0, 1):
            s = po.default_settings(frozen_diagonals=frozen, **kw)
            A_di, bu_di, bv_di, bw_di = po.build_momentum_diffusion_matrix(om, mu)
            mats = [po.initialize_momentum_matrix(om) for _ in range(3)]
            tag = "frozen" if frozen else "faithful"
            if frozen:
                out["a_di"] = A_di.arrays()[2].copy()
                out["b_di"] = np.stack([bu_di, bv_di, bw_di])
                out["row_ptr"], out["col"] = A_di.arrays()[0].copy(), A_di.arrays()[1].copy()
            # two consecutive assemblies (iteration 1 reads diag = 1, iteration 2 the assembled diagonals: Q2/Q3)
            for it in (1, 2):
                bu, bv, bw, pe = po.build_momentum_advection_matrices(mats[0], mats[1], mats[2], A_di, om, u, v, w, p, s, rho)
                out["a_uvw_%s_it%d" % (tag, it)] = np.stack([m.arrays()[2].copy() for m in mats])
                out["b_uvw_%s_it%d" % (tag, it)] = np.stack([bu, bv, bw])
                out["peclet_%s_it%d" % (tag, it)] = np.array(pe)
            A_p, b_p = po.build_pressure_correction_matrices(om, u, v, w, p, mats[0], mats[1], mats[2], s, rho)
            out["a_p_%s" % tag] = A_p.arrays()[2].copy()
            out["b_p_%s" % tag] = b_p
        # Jacobi-solver SIMPLE iterations are reduction-free: fields after 5 iterations are a bitwise pin
        s = po.default_settings(frozen_diagonals=1, solver_type=po.JACOBI, **kw)
        f = [x.copy() for x in (u, v, w, p)]
        st, _ = po.solve_steady(om, *f, s, rho, mu, 5)
        out["status_jacobi_frozen_5it"] = np.array(st)  # the mixed-BC cube diverges in the Jacobi arm: the status is the pin
        if st == 0:
            out["fields_jacobi_frozen_5it"] = np.stack(f)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        print(name, {k: np.asarray(v).shape for k, v in out.items() if k.startswith(("a_uvw", "fields"))})
    # the reference's unit-test system (linear_algebra.rs:313-337) solved by the oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import unit_test_system
    a, b, sol = unit_test_system()
    x = np.zeros(len(b))
    A = po.Csr.from_scipy(a)
    po.iterative_solve(A, b, x, 50, po.JACOBI, 0.5, 1e-3 / len(b) ** 3, po.PRECOND_JACOBI)
    xj = x.copy()
    po.iterative_solve(A, b, x, 50, po.BICGSTAB, 0.5, 1e-3 / len(b) ** 3, po.PRECOND_JACOBI)
    np.savez_compressed(os.path.join(OUT, "unit_test_system.npz"), x_after_jacobi=xj, x_after_bicgstab=x, b=b, solution=sol)


if __name__ == "__main__":
    main()
