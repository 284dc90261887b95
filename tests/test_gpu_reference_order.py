"""Reference-order reductions (OrcSettings.reduction_order = 1): with every dot product and norm summed in nalgebra's
`dotx` association, the device's BiCGSTAB, Jacobi and Multigrid arms — and whole SIMPLE iterations built on them —
reproduce the CPU oracle BIT FOR BIT at the reference's default iteration counts (50 inner iterations,
linear_algebra.rs:247-296, :66-141), where the tree-reduction product path can only be compared through tolerances.
Everything else in the data path (assembly, SpMV, scaling, aggregation, Galerkin product, restriction, prolongation,
the BiCGSTAB recurrences) is the product's own kernels; only the fold of the reductions differs from the default."""
import numpy as np
import pytest

import helpers as H
import meshgen
from conftest import fv_like_matrix, splitmix64_uniform, unit_test_system

pytestmark = pytest.mark.gpu

JACOBI, MULTIGRID, BICGSTAB = 1, 2, 3
PRE_NONE, PRE_JACOBI = 0, 1
REFERENCE = 1


@pytest.fixture()
def reference_order(gpu):
    from orc_amd.linear_algebra import set_breakdown_guard, set_reduction_order
    set_reduction_order(REFERENCE)
    set_breakdown_guard(False)  # the reference has no guard (linear_algebra.rs:255-268)
    yield
    set_reduction_order(0)
    set_breakdown_guard(True)


def same_bits(a, b):
    """identical bit patterns; NaNs must sit in the same places (their sign/payload is hardware business)"""
    a, b = np.asarray(a), np.asarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint64), b[~nb].view(np.uint64))


def test_reference_unit_test_bit_exact(reference_order, oracle):
    """validate_iterative_solvers (linear_algebra.rs:309-378): Jacobi(50, omega 0.5) then BiCGSTAB(50) on the 100x100
    system, x carried over, Jacobi preconditioner — every bit of both iterates."""
    from orc_amd.linear_algebra import iterative_solve
    a, b, sol = unit_test_system()
    n = len(b)
    A = oracle.Csr.from_scipy(a)
    x, xo = np.zeros(n), np.zeros(n)
    for method in (JACOBI, BICGSTAB):
        iterative_solve(a, b, x, 50, method, 0.5, 1e-3 / n ** 3, PRE_JACOBI)
        assert oracle.iterative_solve(A, b, xo, 50, method, 0.5, 1e-3 / n ** 3, PRE_JACOBI) == 0
        assert same_bits(x, xo)
        assert np.linalg.norm(a @ x - b) < 1e-3  # the reference's assertion


@pytest.mark.parametrize("shape", [(7, 5, 3), (33, 9, 4), (64, 40, 12)])
@pytest.mark.parametrize("precond", [PRE_NONE, PRE_JACOBI])
def test_bicgstab_default_count_bit_exact(reference_order, oracle, shape, precond):
    """50 BiCGSTAB iterations (the reference's default count, lib.rs:80) on non-symmetric FV-like systems."""
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(*shape)
    n = a.shape[0]
    b = a @ splitmix64_uniform(n, 7)
    x0 = 0.1 * splitmix64_uniform(n, 8)
    x, xo = x0.copy(), x0.copy()
    st = iterative_solve(a, b, x, 50, BICGSTAB, 0.5, 1e-3, precond, raise_on_error=False)
    sto = oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 50, BICGSTAB, 0.5, 1e-3, precond)
    assert st == sto
    assert same_bits(x, xo)


@pytest.mark.parametrize("shape", [(7, 5, 3), (20, 17, 9), (64, 40, 12)])
def test_multigrid_arm_default_count_bit_exact(reference_order, oracle, shape):
    """The whole Multigrid arm at 50 smoother iterations per level (Q4 nested scaling, Q5 r' recursion, Q6 weight-2
    rows): status and every bit of x."""
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(*shape)
    n = a.shape[0]
    b = a @ splitmix64_uniform(n, 7)
    x0 = 0.1 * splitmix64_uniform(n, 8)
    x, xo = x0.copy(), x0.copy()
    st = iterative_solve(a, b, x, 50, MULTIGRID, 0.5, 1e-3, PRE_JACOBI, raise_on_error=False)
    sto = oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 50, MULTIGRID, 0.5, 1e-3, PRE_JACOBI)
    assert st == sto
    assert same_bits(x, xo)


def test_jacobi_convergence_break_bit_exact(reference_order, oracle):
    """The Jacobi arm's break (:210-213) compares residual norms: with the reference's association the sweep count and
    the iterate match whatever the threshold."""
    from orc_amd.linear_algebra import iterative_solve, last_jacobi_sweeps
    a = fv_like_matrix(20, 17, 9)
    n = a.shape[0]
    b = a @ splitmix64_uniform(n, 3)
    for thr in (1e-1, 1e-2, 1e-4):
        x, xo = np.zeros(n), np.zeros(n)
        iterative_solve(a, b, x, 200, JACOBI, 0.7, thr, PRE_JACOBI)
        assert oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 200, JACOBI, 0.7, thr, PRE_JACOBI) == 0
        assert last_jacobi_sweeps() == oracle.lib().or_last_jacobi_sweeps()
        assert same_bits(x, xo)


def _fixture_mesh(oracle, mesh_path, name):
    from orc_amd.mesh import Mesh, MeshArrays
    om = oracle.Mesh.read(mesh_path(name))
    if name == "3x3_cube":
        H.cube_bcs_mixed(om)
    else:
        H.channel_bcs(om)
    a = MeshArrays(om.arrays())
    return om, Mesh(a), a


@pytest.mark.parametrize("name,inner", [("channel_flow", 50), ("3x3_cube", 50), ("3x3_cube", 20), ("couette_flow_8x8x1", 50),
                                        ("couette_flow_8x8x1", 20)])
@pytest.mark.parametrize("solver,momentum", [(MULTIGRID, 1), (MULTIGRID, 5), (BICGSTAB, 5)])
def test_solve_steady_default_stack_bit_exact(gpu, oracle, mesh_path, name, inner, solver, momentum):
    """Six SIMPLE iterations of NumericalSettings::default() (Multigrid + Jacobi preconditioner, 50 inner iterations,
    Rhie-Chow, SecondOrder; CD1 and TVD-UMIST momentum) and of the BiCGSTAB solver: u, v, w, p identical to the oracle
    (frozen-diagonal mode on both sides) after every one of them.  On the 27- and 64-cell meshes the reference's
    unguarded BiCGSTAB reaches an exactly zero residual within 50 iterations and panics ("Multigrid diverged" /
    "solution diverged"): the device must report the same status in the same iteration (fields are not observable
    after a panic); 20 inner iterations keep those meshes finite."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    om, dm, a = _fixture_mesh(oracle, mesh_path, name)
    kw = dict(momentum=momentum, solver_type=solver, iterations=inner, frozen_diagonals=1, breakdown_guard=0)
    u, v, w, p = H.seeded_fields(a, seed=5, scale_u=4e-4)
    fo = [x.copy() for x in (u, v, w, p)]
    compared = 0
    s = Solver(dm, NumericalSettings.default(reduction_order=REFERENCE, **kw), 1000.0, 1e-3)
    s.set_fields(u, v, w, p)
    for it in range(6):
        std = s.iterate(1, raise_on_error=False)
        # the oracle keeps its matrices inside one call: it + 1 iterations from the start for the comparison
        ref = [x.copy() for x in fo]
        sto, _ = oracle.solve_steady(om, *ref, oracle.default_settings(**kw), 1000.0, 1e-3, it + 1)
        assert std == sto, "iteration %d" % (it + 1)
        if std != 0:
            break
        for x, y in zip(s.get_fields(), ref):
            assert same_bits(x, y), "iteration %d" % (it + 1)
        compared += 1
    if name == "channel_flow" or inner == 20:
        assert compared == 6


def test_solve_steady_mixed_prism_hex_bit_exact(gpu, oracle, tmp_path):
    """The same on the skewed prism + hexahedron mesh (ragged rows, non-orthogonal faces), product reader included."""
    from orc_amd import io as orc_io
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    path = str(tmp_path / "mixed.msh")
    info = meshgen.write_mixed_channel_msh(path, 10, 6, 4, skew=0.2)
    om = oracle.Mesh.read(path)
    meshgen.mixed_channel_bcs(om.set_zone, info["zone_names"], top_wall_velocity=5e-4)
    d = orc_io.read_mesh(path)
    meshgen.mixed_channel_bcs(d.set_zone, info["zone_names"], top_wall_velocity=5e-4)
    dm, a = d.upload(), d.arrays()
    kw = dict(momentum=5, frozen_diagonals=1, breakdown_guard=0)
    u, v, w, p = H.seeded_fields(a, seed=9)
    uo, vo, wo, po = (x.copy() for x in (u, v, w, p))
    sto, _ = oracle.solve_steady(om, uo, vo, wo, po, oracle.default_settings(**kw), 1000.0, 1e-3, 5)
    std = solve_steady(dm, u, v, w, p, NumericalSettings.default(reduction_order=REFERENCE, **kw), 1000.0, 1e-3, 5, raise_on_error=False)
    assert std == sto == 0
    for x, y in ((u, uo), (v, vo), (w, wo), (p, po)):
        assert same_bits(x, y)


def test_initialize_flow_bit_exact(gpu, oracle, mesh_path):
    """initialize_flow (solver.rs:246-352): Laplace pressure start + six blended systems x three 40-iteration BiCGSTAB
    solves, every bit."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import initialize_flow
    om, dm, a = _fixture_mesh(oracle, mesh_path, "channel_flow")
    sto, uo, vo, wo, po = oracle.initialize_flow(om, 1e-3, 1000.0, 40)
    assert sto == 0
    fields = initialize_flow(dm, 1e-3, 1000.0, 40, NumericalSettings.default(reduction_order=REFERENCE, breakdown_guard=0))
    for x, y in zip(fields, (uo, vo, wo, po)):
        assert same_bits(x, y)


def test_tree_and_reference_orders_agree_to_rounding(gpu):
    """Sanity of the product default against the verification mode on a well-conditioned solve: a few ulps."""
    from orc_amd.linear_algebra import iterative_solve, set_reduction_order
    a = fv_like_matrix(33, 9, 4)
    n = a.shape[0]
    b = a @ splitmix64_uniform(n, 7)
    xs = []
    for order in (0, 1):
        set_reduction_order(order)
        x = np.zeros(n)
        iterative_solve(a, b, x, 8, BICGSTAB, 0.5, 1e-3, PRE_JACOBI)
        xs.append(x)
    set_reduction_order(0)
    assert H.rel_l2(xs[0], xs[1]) < 1e-10


@pytest.mark.parametrize("name", ["channel_flow", "3x3_cube", "couette_flow_8x8x1"])
@pytest.mark.parametrize("momentum,solver,inner", [(1, MULTIGRID, 50), (5, BICGSTAB, 20), (0, JACOBI, 50)])
def test_solve_steady_in_place_diagonals_bit_exact(gpu, oracle, mesh_path, name, momentum, solver, inner):
    """frozen_diagonals = 0: the reference's OWN mode — Rhie-Chow reads the momentum diagonals while the assembly loop
    rewrites them (discretization.rs:182-197, 340-351), evaluated on the device level by level over the cell
    order.  With the reference's reduction order on top, five SIMPLE iterations reproduce the oracle's default mode bit for
    bit: no deviation is left between the device and the restated reference in a transient."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    om, dm, a = _fixture_mesh(oracle, mesh_path, name)
    kw = dict(momentum=momentum, solver_type=solver, iterations=inner, frozen_diagonals=0, breakdown_guard=0)
    u, v, w, p = H.seeded_fields(a, seed=7, scale_u=4e-4)
    fo = [x.copy() for x in (u, v, w, p)]
    s = Solver(dm, NumericalSettings.default(reduction_order=REFERENCE, **kw), 1000.0, 1e-3)
    s.set_fields(u, v, w, p)
    compared = 0
    for it in range(5):
        std = s.iterate(1, raise_on_error=False)
        ref = [x.copy() for x in fo]
        sto, _ = oracle.solve_steady(om, *ref, oracle.default_settings(**kw), 1000.0, 1e-3, it + 1)
        assert std == sto, "iteration %d" % (it + 1)
        if std != 0:
            break
        for x, y in zip(s.get_fields(), ref):
            assert same_bits(x, y), "iteration %d" % (it + 1)
        compared += 1
    assert compared >= (5 if name == "channel_flow" else 1)


def test_in_place_and_frozen_diagonals_differ_in_the_transient_only(gpu, oracle, mesh_path):
    """The two modes give different iterates (that is the point of having both) and the same converged fields."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    om, dm, a = _fixture_mesh(oracle, mesh_path, "channel_flow")
    u, v, w, p = H.seeded_fields(a, seed=7, scale_u=4e-4)
    out = {}
    for frozen in (0, 1):
        s = Solver(dm, NumericalSettings.default(momentum=1, solver_type=BICGSTAB, frozen_diagonals=frozen), 1000.0, 1e-3)
        s.set_fields(u, v, w, p)
        s.iterate(3)
        early = s.get_fields()
        s.iterate(1500)
        out[frozen] = (early, s.get_fields())
    assert not same_bits(out[0][0][0], out[1][0][0])
    assert H.rel_l2(out[0][1][0], out[1][1][0]) < 1e-8 and H.rel_l2(out[0][1][3], out[1][1][3]) < 1e-8
