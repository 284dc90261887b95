"""the least-squares gradient arms (solver.rs:803-869, 903-947), the gradient pass solve_steady runs after
its loop (solver.rs:227-242) and initialize_velocity_field (solver.rs:511-696) on the device against the oracle."""
import numpy as np
import pytest

import helpers as H
import meshgen

pytestmark = pytest.mark.gpu

LSQ, GG_NODE = 2, 1
JACOBI, MULTIGRID, BICGSTAB = 1, 2, 3


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint64), b[~nb].view(np.uint64))


def fixture(oracle, mesh_path, name):
    from orc_amd.mesh import Mesh, MeshArrays
    om = oracle.Mesh.read(mesh_path(name))
    if name == "3x3_cube":
        H.cube_bcs_mixed(om)
    elif name == "3D_1x3":
        H.line_bcs(om)
    else:
        H.channel_bcs(om)
    a = MeshArrays(om.arrays())
    return om, Mesh(a), a


def oracle_lsq_gradients(oracle, om, u, v, w, p):
    import ctypes as C
    n = om.n_cells
    gp, gu = np.zeros((n, 3)), np.zeros((n, 3, 3))
    L = oracle.lib()
    vec, ten = oracle._Vec3(), (oracle._Vec3 * 3)()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for c in range(n):
        assert L.or_calculate_pressure_gradient(om.ptr, dp(p), C.c_int64(c), C.c_int(LSQ), C.c_int(1), C.byref(vec)) == 0
        gp[c] = (vec.x, vec.y, vec.z)
        assert L.or_calculate_velocity_gradient(om.ptr, dp(u), dp(v), dp(w), C.c_int64(c), C.c_int(LSQ), ten) == 0
        gu[c] = [(r.x, r.y, r.z) for r in ten]
    return gp, gu


@pytest.mark.parametrize("name", ["3x3_cube", "channel_flow", "couette_flow_8x8x1"])
def test_least_squares_gradients_bit_exact(gpu, oracle, mesh_path, name):
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import calculate_gradients
    om, dm, a = fixture(oracle, mesh_path, name)
    u, v, w, p = H.seeded_fields(a, seed=3)
    gp, gu = calculate_gradients(dm, u, v, w, p, NumericalSettings.default(gradient_reconstruction=LSQ))
    gpo, guo = oracle_lsq_gradients(oracle, om, u, v, w, p)
    assert same_bits(gp, gpo) and same_bits(gu, guo)


def test_least_squares_gradient_is_exact_for_linear_fields_in_the_interior(gpu, oracle, mesh_path):
    """A linear field's least-squares gradient is exact where every face is interior (boundary rows carry the face VALUE
    instead of a difference in the reference, so cells that touch a boundary are not): centre cell of the 3x3x3 cube."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import calculate_gradients
    om, dm, a = fixture(oracle, mesh_path, "3x3_cube")
    cc = np.asarray(a["cell_centroid"])
    g = np.array([0.3, -1.1, 0.7])
    p = cc @ g + 0.25
    u, v, w = cc @ np.array([1.0, 2.0, 3.0]), cc @ np.array([-0.5, 0.1, 0.2]), cc @ np.array([0.0, 0.4, -0.9])
    gp, gu = calculate_gradients(dm, u, v, w, p, NumericalSettings.default(gradient_reconstruction=LSQ))
    nf = np.diff(a["cell_face_ptr"])
    interior = [c for c in range(len(cc)) if all(a["face_c1"][f] >= 0 for f in a["cell_faces"][a["cell_face_ptr"][c]:a["cell_face_ptr"][c + 1]])]
    assert len(interior) == 1 and nf[interior[0]] == 6
    c = interior[0]
    assert np.allclose(gp[c], g, rtol=1e-12, atol=1e-12)
    assert np.allclose(gu[c], [[1.0, 2.0, 3.0], [-0.5, 0.1, 0.2], [0.0, 0.4, -0.9]], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("solver", [JACOBI, BICGSTAB])
def test_solve_steady_with_least_squares_gradients_bit_exact(gpu, oracle, mesh_path, solver):
    """Whole SIMPLE iterations with gradient_reconstruction = LeastSquares feeding Rhie-Chow, SecondOrder and the TVD
    ratio (flagged "causing unphysical oscillations" in the reference, lib.rs:157 — parity, not physics)."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    om, dm, a = fixture(oracle, mesh_path, "channel_flow")
    kw = dict(momentum=5, solver_type=solver, iterations=20, gradient_reconstruction=LSQ, frozen_diagonals=1, breakdown_guard=0)
    u, v, w, p = H.seeded_fields(a, seed=6, scale_u=4e-4)
    fo = [x.copy() for x in (u, v, w, p)]
    sto, _ = oracle.solve_steady(om, *fo, oracle.default_settings(**kw), 1000.0, 1e-3, 3)
    std = solve_steady(dm, u, v, w, p, NumericalSettings.default(reduction_order=1, **kw), 1000.0, 1e-3, 3, raise_on_error=False)
    assert std == sto == 0
    for x, y in zip((u, v, w, p), fo):
        assert same_bits(x, y)


def test_post_loop_gradient_pass_panics_like_the_reference(gpu, oracle, mesh_path):
    """solver.rs:227-242 evaluates both gradients once more after the loop: a singular least-squares matrix (a 2-D-like
    cell whose rows span only two directions) is the reference's try_inverse().unwrap() panic, also with zero iterations."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    om, dm, a = fixture(oracle, mesh_path, "3D_1x3")
    n = dm.n_cells
    f = [np.zeros(n) for _ in range(4)]
    fo = [np.zeros(n) for _ in range(4)]
    kw = dict(gradient_reconstruction=LSQ, frozen_diagonals=1)
    sto, _ = oracle.solve_steady(om, *fo, oracle.default_settings(**kw), 1000.0, 1e-3, 0)
    std = solve_steady(dm, *f, NumericalSettings.default(**kw), 1000.0, 1e-3, 0, raise_on_error=False)
    assert std == sto


@pytest.mark.parametrize("name", ["3x3_cube", "channel_flow"])
def test_initialize_velocity_field_bit_exact(gpu, oracle, mesh_path, name):
    """Potential-flow initialisation with a velocity inlet and a pressure outlet: psi after ten BiCGSTAB iterations and the
    least-squares velocities (zero columns dropped on the one-cell-thick mesh), every bit."""
    from orc_amd.mesh import Mesh, MeshArrays
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import VELOCITY_ONLY, check_boundary_conditions, initialize_flow_new, initialize_velocity_field
    om = oracle.Mesh.read(mesh_path(name))
    if name == "3x3_cube":
        H.cube_bcs_mixed(om)
    else:
        H.channel_bcs(om)
        om.set_zone("INLET", H.BC_VINLET, 0.0, (1e-3, 0.0, 0.0))
    dm = Mesh(MeshArrays(om.arrays()))
    sto, uo, vo, wo, psio = oracle.initialize_velocity_field(om)
    assert sto == 0
    u, v, w, psi = initialize_velocity_field(dm, NumericalSettings.default(reduction_order=1, breakdown_guard=0))
    assert same_bits(psi, psio)
    for x, y in ((u, uo), (v, vo), (w, wo)):
        assert same_bits(x, y)
    assert np.abs(u).max() > 0
    if check_boundary_conditions(dm) == VELOCITY_ONLY:
        un, vn, wn, pn = initialize_flow_new(dm, 1e-3, 1000.0, 10)  # the same arm through initialize_flow_new (product defaults)
        ud = initialize_velocity_field(dm)[0]
        assert not pn.any() and same_bits(un, ud) and np.isfinite(un).all()
