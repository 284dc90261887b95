"""HIP path vs the committed golden vectors (tests/golden/*.npz) — no oracle call needed on the GPU box for these."""
import os

import numpy as np
import pytest

import helpers as H
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

CASES = {"3x3_cube": ("3x3_cube", H.cube_bcs, dict(momentum=5)), "3x3_cube_mixed": ("3x3_cube", H.cube_bcs_mixed, dict(momentum=1)),
         "channel_flow": ("channel_flow", H.channel_bcs, dict(momentum=1))}


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_reproduces_golden(gpu, oracle, mesh_path, name):
    from orc_amd import discretization as D
    from orc_amd.mesh import Mesh, MeshArrays
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    mesh, bcs, kw = CASES[name]
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    om = oracle.Mesh.read(mesh_path(mesh))  # the oracle is used as the .msh reader only
    bcs(om)
    dm = Mesh(MeshArrays(om.arrays()))
    u, v, w, p = (np.ascontiguousarray(g[k]) for k in ("u0", "v0", "w0", "p0"))
    s = NumericalSettings.default(frozen_diagonals=1, **kw)
    rp, ci = dm.matrix_pattern()
    assert np.array_equal(rp, g["row_ptr"]) and np.array_equal(ci, g["col"])
    a_di, bu, bv, bw = D.build_momentum_diffusion_matrix(dm, 1e-3)
    assert np.array_equal(a_di, g["a_di"]) and np.array_equal(np.stack([bu, bv, bw]), g["b_di"])
    mats = [D.initialize_momentum_matrix(dm) for _ in range(3)]
    for it in (1, 2):
        b = D.build_momentum_advection_matrices(dm, mats[0], mats[1], mats[2], a_di, u, v, w, p, s, 1000.0)
        assert np.array_equal(np.stack(mats), g["a_uvw_frozen_it%d" % it], equal_nan=True)
        assert np.array_equal(np.stack(b[:3]), g["b_uvw_frozen_it%d" % it], equal_nan=True)
    a_p, b_p = D.build_pressure_correction_matrices(dm, u, v, w, p, mats[0], mats[1], mats[2], s, 1000.0)
    assert np.array_equal(a_p, g["a_p_frozen"], equal_nan=True) and np.array_equal(b_p, g["b_p_frozen"], equal_nan=True)
    f = [x.copy() for x in (u, v, w, p)]
    st = solve_steady(dm, *f, NumericalSettings.default(frozen_diagonals=1, solver_type=1, **kw), 1000.0, 1e-3, 5, raise_on_error=False)
    assert st == int(g["status_jacobi_frozen_5it"])
    if st == 0:
        assert np.array_equal(np.stack(f), g["fields_jacobi_frozen_5it"])
