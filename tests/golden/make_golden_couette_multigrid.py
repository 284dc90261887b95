#!/usr/bin/env python3
"""Generates tests/golden/couette_multigrid_converged.npz: the converged u/v/w/p of couette_flow_128x64x1.msh under the
reference's DEFAULT solver stack (Multigrid + Jacobi preconditioner, 50 smoother iterations, CD1 momentum, Rhie-Chow,
SecondOrder) computed by the CPU oracle in the reference's own in-place mode (frozen_diagonals = 0), BCs of
tests.rs:60-76 with a moving top wall.  "Restatement-derived" (SURVEY 8c): the reference cannot be built here.
About 3 minutes of CPU.  Run from the repo root:  python tests/golden/make_golden_couette_multigrid.py
"""
import gzip
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from conftest import splitmix64_uniform  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

ITERATIONS = 1500


def start_fields(cc):
    n = len(cc)
    ua = H.analytical_poiseuille(cc[:, 1], dp_dx=10.0, u_top=5e-4)
    u0 = ua * (1 + 0.02 * splitmix64_uniform(n, 1))
    v0 = 1e-7 * splitmix64_uniform(n, 2)
    w0 = 1e-12 * splitmix64_uniform(n, 3)
    p0 = -0.02 * (1 - cc[:, 0] / 0.002) * (1 + 0.01 * splitmix64_uniform(n, 4))
    return u0, v0, w0, p0


def main():
    gz = os.path.join(ROOT, "tests", "golden", "meshes", "couette_flow_128x64x1.msh.gz")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "couette.msh")
        with gzip.open(gz, "rb") as fi, open(path, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        om = po.Mesh.read(path)
    H.channel_bcs(om, top_wall_velocity=5e-4, dp_dx=10.0)
    cc = np.asarray(om.arrays()["cell_centroid"])
    f = [x.copy() for x in start_fields(cc)]
    s = po.default_settings(momentum=po.CD1, solver_type=po.MULTIGRID, iterations=50, frozen_diagonals=0)
    st, rep = po.solve_steady(om, *f, s, 1000.0, 1e-3, ITERATIONS, report=True)
    assert st == 0
    print("velocity-correction norm after %d iterations: %.3e" % (ITERATIONS, rep[-1][4]))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "couette_multigrid_converged.npz"), u=f[0], v=f[1], w=f[2], p=f[3],
                        iterations=ITERATIONS, velocity_correction=rep[-1][4])


if __name__ == "__main__":
    main()
