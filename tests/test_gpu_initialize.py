"""solver::initialize_pressure_field / initialize_flow / initialize_flow_new / check_boundary_conditions
(solver.rs:246-509, 710-772) on the device against the CPU oracle."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def setup(oracle, mesh_path, name, bcs=H.channel_bcs, **bc):
    from orc_amd.mesh import Mesh, MeshArrays
    om = oracle.Mesh.read(mesh_path(name))
    bcs(om, **bc)
    a = MeshArrays(om.arrays())
    return om, Mesh(a), a


def hex_setup(oracle, nx, ny, nz):
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    a = set_channel_bcs(hex_channel(nx, ny, nz))
    return oracle.Mesh.from_arrays(a), Mesh(a), a


@pytest.mark.parametrize("name, bcs, kw", [
    ("couette_flow_8x8x1", H.channel_bcs, dict(top_wall_velocity=5e-4)),
    ("channel_flow", H.channel_bcs, {}),
    ("3x3_cube", H.cube_bcs, {}),
    ("3x3_cube", H.cube_bcs_mixed, {}),
    ("couette_flow_128x64x1", H.channel_bcs, dict(top_wall_velocity=5e-4)),
])
def test_initialize_pressure_field_bit_exact(gpu, oracle, mesh_path, name, bcs, kw):
    """Laplace assembly is cell-local with the reference's summation order and the Jacobi arm has no reduction in its
    data path: ten sweeps reproduce the oracle bit for bit."""
    from orc_amd.solver import initialize_pressure_field
    om, dm, _ = setup(oracle, mesh_path, name, bcs, **kw)
    st, po = oracle.initialize_pressure_field(om)
    assert st == 0
    p = initialize_pressure_field(dm)
    assert np.array_equal(p, po)
    assert np.abs(po).max() > 0


def test_initialize_pressure_field_true_3d(gpu, oracle):
    from orc_amd.solver import initialize_pressure_field
    om, dm, _ = hex_setup(oracle, 12, 8, 6)
    st, po = oracle.initialize_pressure_field(om)
    assert st == 0 and np.array_equal(initialize_pressure_field(dm), po)


def test_check_boundary_conditions(gpu, oracle, mesh_path):
    from orc_amd._lib import OrcError
    from orc_amd.solver import HYBRID, PRESSURE_ONLY, VELOCITY_ONLY, check_boundary_conditions, initialize_flow, initialize_flow_new
    _, dm, a = setup(oracle, mesh_path, "couette_flow_8x8x1")
    assert check_boundary_conditions(dm) == PRESSURE_ONLY
    a.set_zone("WALL", H.BC_WALL, 0.0, (5e-4, 0.0, 0.0))
    dm.update_zones()
    assert check_boundary_conditions(dm) == HYBRID  # moving wall + two pressure zones (solver.rs:755-758)
    a.set_zone("INLET", H.BC_VINLET, 0.0, (1e-3, 0.0, 0.0))
    dm.update_zones()
    assert check_boundary_conditions(dm) == VELOCITY_ONLY  # one pressure zone left
    u, v, w, p = initialize_flow_new(dm, 1e-3, 1000.0, 10)  # VelocityOnly -> initialize_velocity_field, pressure stays zero
    assert not p.any() and np.isfinite(u).all() and np.abs(u).max() > 0
    for z in ("INLET", "OUTLET", "WALL"):
        a.set_zone(z, H.BC_WALL)
    dm.update_zones()
    for fn in (lambda: check_boundary_conditions(dm), lambda: initialize_flow(dm, 1e-3, 1000.0, 5)):
        with pytest.raises(OrcError) as e:
            fn()
        assert e.value.status == 17 and "You must set boundary conditions." in str(e.value)


def test_initialize_flow_new_is_pressure_initialisation(gpu, oracle, mesh_path):
    from orc_amd.solver import initialize_flow_new
    om, dm, _ = setup(oracle, mesh_path, "channel_flow")
    u, v, w, p = initialize_flow_new(dm, 1e-3, 1000.0, 1000)
    st, po = oracle.initialize_pressure_field(om)
    assert st == 0 and np.array_equal(p, po)
    assert not u.any() and not v.any() and not w.any()


@pytest.mark.parametrize("guard", [0, 1])
def test_initialize_flow_short_ramp_matches_oracle(gpu, oracle, mesh_path, guard):
    """Three BiCGSTAB iterations per solve (18 solves): assembly, blend and SpMV are bit-exact, the dot products
    associate differently (block partials vs nalgebra's 8 accumulators) — a few ulps per iteration."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import initialize_flow
    om, dm, _ = setup(oracle, mesh_path, "channel_flow")
    st, uo, vo, wo, po = oracle.initialize_flow(om, 1e-3, 1000.0, 3)
    assert st == 0
    u, v, w, p = initialize_flow(dm, 1e-3, 1000.0, 3, NumericalSettings.default(breakdown_guard=guard))
    assert np.array_equal(p, po)
    assert H.rel_l2(u, uo) < 1e-9, H.rel_l2(u, uo)
    assert H.rel_l2(v, vo) < 1e-6, H.rel_l2(v, vo)
    assert np.abs(w).max() < 1e-18 and np.abs(wo).max() < 1e-18  # a 2-D case: w stays at round-off


@pytest.mark.parametrize("iteration_count", [100, 1000])
def test_initialize_flow_converged_fields(gpu, oracle, mesh_path, iteration_count):
    """tests.rs:85 calls initialize_flow(mesh, mu, rho, 1000): every solve runs to stagnation, so the fields are the
    solutions of the last (pure advection + diffusion) systems — the north-star tolerance applies: 1e-6 rel-L2."""
    from orc_amd.solver import initialize_flow
    om, dm, _ = setup(oracle, mesh_path, "channel_flow")
    st, uo, vo, wo, po = oracle.initialize_flow(om, 1e-3, 1000.0, iteration_count)
    assert st == 0 and not np.isnan(uo).any()
    u, v, w, p = initialize_flow(dm, 1e-3, 1000.0, iteration_count)
    assert np.array_equal(p, po)
    assert H.rel_l2(u, uo) < 1e-6, H.rel_l2(u, uo)
    assert H.rel_l2(v, vo) < 1e-6, H.rel_l2(v, vo)
    assert np.abs(w).max() < 1e-15


def test_initialize_flow_true_3d(gpu, oracle):
    """Synthetic hex channel (576 cells): every solve has converged after ~30 iterations.  At 100 the oracle still holds
    the converged fields; at 200 its unguarded v solve has hit 0/0 (NaN), while the guarded device solve keeps them."""
    from orc_amd.solver import initialize_flow
    om, dm, _ = hex_setup(oracle, 12, 8, 6)
    st, uo, vo, wo, po = oracle.initialize_flow(om, 1e-3, 1000.0, 100)
    assert st == 0 and not (np.isnan(uo).any() or np.isnan(vo).any() or np.isnan(wo).any())
    for iteration_count in (100, 200):
        u, v, w, p = initialize_flow(dm, 1e-3, 1000.0, iteration_count)
        assert np.array_equal(p, po)
        for a, b in ((u, uo), (v, vo), (w, wo)):
            assert H.rel_l2(a, b) < 1e-6, (iteration_count, H.rel_l2(a, b))
    assert np.abs(vo).max() > 1e-6 and np.abs(wo).max() > 1e-6  # non-zero through SURVEY Q1 (z := y)


def test_initialize_flow_where_the_unguarded_reference_breaks_down(gpu, oracle, mesh_path):
    """couette_flow_8x8x1 (64 cells): BiCGSTAB converges exactly within 100 iterations and the reference's unguarded
    recurrences then divide 0 by 0 (the oracle returns NaN fields, and solve_steady would panic "solution diverged").
    The device's breakdown guard (an extension, on by default) stops updating x at that point, so the fields are the
    converged ones: they match the oracle's result for an iteration count short of the breakdown, and they start a
    SIMPLE run that stays finite."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import initialize_flow, solve_steady
    om, dm, _ = setup(oracle, mesh_path, "couette_flow_8x8x1", top_wall_velocity=5e-4)
    st, uo, *_ = oracle.initialize_flow(om, 1e-3, 1000.0, 1000)
    assert st == 0 and np.isnan(uo).all()
    st, u20, v20, w20, p20 = oracle.initialize_flow(om, 1e-3, 1000.0, 20)
    assert st == 0 and not np.isnan(u20).any()
    u, v, w, p = initialize_flow(dm, 1e-3, 1000.0, 1000)
    assert np.isfinite(u).all() and np.isfinite(v).all() and np.isfinite(w).all()
    assert np.array_equal(p, p20)
    assert H.rel_l2(u, u20) < 1e-6, H.rel_l2(u, u20)
    assert np.abs(v - v20).max() < 1e-6 * np.abs(u20).max()
    solve_steady(dm, u, v, w, p, NumericalSettings.default(solver_type=3, iterations=20), 1000.0, 1e-3, 5)
    assert np.isfinite(u).all() and np.isfinite(p).all()


def test_initializers_return_orc_order_on_an_internally_reordered_mesh(gpu):
    """orc_mesh_create_reordered renumbers the cells inside the library only: initialize_pressure_field / initialize_flow_new
    (like solve_steady and the solver's set/get_fields) hand their fields back in ORC cell order.  A randomly renumbered channel
    with the internal RCM ordering against the same mesh without it: equal up to the summation order of the face loops."""
    from orc_amd.mesh import Mesh, hex_channel, renumber_cells, set_channel_bcs
    from orc_amd.solver import initialize_flow_new, initialize_pressure_field
    a = set_channel_bcs(hex_channel(12, 9, 7))
    n = a.n_cells
    perm = np.random.default_rng(5).permutation(n)
    sh = renumber_cells(a, perm)
    plain, reordered = Mesh(sh), Mesh(sh, ordering=1)
    assert not np.array_equal(reordered.cell_order(), np.arange(n))
    p0, p1 = initialize_pressure_field(plain), initialize_pressure_field(reordered)
    assert np.linalg.norm(p1 - p0) <= 1e-12 * np.linalg.norm(p0) and np.linalg.norm(p0) > 0
    f0, f1 = initialize_flow_new(plain, 1e-3, 1000.0, 10), initialize_flow_new(reordered, 1e-3, 1000.0, 10)
    for x, y in zip(f0, f1):
        assert np.linalg.norm(x - y) <= 1e-12 * max(np.linalg.norm(x), 1e-300)
