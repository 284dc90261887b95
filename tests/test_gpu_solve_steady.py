"""solve_steady on the device vs the CPU oracle (frozen diagonals) and vs the analytical profile."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

JACOBI, MULTIGRID, BICGSTAB = 1, 2, 3


def setup(oracle, mesh_path, name, **bc):
    from orc_amd.mesh import Mesh, MeshArrays
    om = oracle.Mesh.read(mesh_path(name))
    H.channel_bcs(om, **bc)
    a = MeshArrays(om.arrays())
    return om, Mesh(a), a


def test_config1_ud_jacobi_bit_exact(gpu, oracle, mesh_path):
    """BASELINE config 1: couette_flow_8x8x1.msh, UD momentum, Jacobi smoother.  The Jacobi arm has no
    reductions in its data path, so ten full SIMPLE iterations reproduce the oracle (frozen mode) bit for bit."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    om, dm, a = setup(oracle, mesh_path, "couette_flow_8x8x1")
    kw = dict(momentum=0, solver_type=JACOBI, frozen_diagonals=1)
    u, v, w, p = H.seeded_fields(a, seed=2, w_zero=True)
    uo, vo, wo, po_ = (x.copy() for x in (u, v, w, p))
    st, _ = oracle.solve_steady(om, uo, vo, wo, po_, oracle.default_settings(**kw), 1000.0, 1e-3, 10)
    assert st == 0
    reports = []
    solve_steady(dm, u, v, w, p, NumericalSettings.default(**kw), 1000.0, 1e-3, 10, reporting_interval=5,
                 report=lambda *r: reports.append(r))
    assert np.array_equal(u, uo) and np.array_equal(v, vo) and np.array_equal(w, wo) and np.array_equal(p, po_)
    assert [r[0] for r in reports] == [5, 10]
    assert abs(reports[-1][1][0] - u.mean()) < 1e-15


def _association_sensitivity(oracle, om, fields, kw, iters):
    """rel-L2 change of (u, p) when the oracle's dot products are re-associated (pairwise instead of
    nalgebra's 8-accumulator order).  The reference's BiCGSTAB uses r_hat_0 = 1 (linear_algebra.rs:252):
    rho = sum(r) is a cancelling sum and there is no breakdown guard, so its iterates are
    ill-conditioned with respect to that association — the only freedom a parallel reduction takes."""
    res = []
    for mode in (0, 1):
        oracle.set_dot_mode(mode)
        try:
            f = [x.copy() for x in fields]
            st, _ = oracle.solve_steady(om, *f, oracle.default_settings(**kw), 1000.0, 1e-3, iters)
        finally:
            oracle.set_dot_mode(0)
        assert st == 0
        res.append(f)
    return res[0], (H.rel_l2(res[1][0], res[0][0]), H.rel_l2(res[1][3], res[0][3]))


@pytest.mark.parametrize("momentum", [1, 5])
def test_channel_flow_one_iteration_short_bicgstab(gpu, oracle, mesh_path, momentum):
    """channel_flow.msh (1008 cells), BiCGSTAB (no AMG), Rhie-Chow + SecondOrder.  Assembly and SpMV are
    bit-exact; the dot products associate differently, and the reference's unguarded BiCGSTAB (r_hat_0 = 1,
    rho = sum(r) cancels) amplifies that chaotically once it runs tens of iterations
    (tests/test_oracle_sensitivity.py measures 1e-2 after five SIMPLE iterations for the oracle against
    itself).  With 5 BiCGSTAB iterations per solve the two paths still agree to 1e-9 after a full SIMPLE
    iteration, which pins the wiring of the device loop (order of solves, p' reset, correction)."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    om, dm, a = setup(oracle, mesh_path, "channel_flow")
    st, u0, v0, w0, p0 = oracle.initialize_flow(om, 1e-3, 1000.0, 200)
    assert st == 0 and not np.isnan(u0).any()
    kw = dict(momentum=momentum, solver_type=BICGSTAB, iterations=5, frozen_diagonals=1)
    uo, vo, wo, po_ = (x.copy() for x in (u0, v0, w0, p0))
    assert oracle.solve_steady(om, uo, vo, wo, po_, oracle.default_settings(**kw), 1000.0, 1e-3, 1)[0] == 0
    u, v, w, p = (x.copy() for x in (u0, v0, w0, p0))
    solve_steady(dm, u, v, w, p, NumericalSettings.default(**kw), 1000.0, 1e-3, 1)
    assert H.rel_l2(u, uo) < 1e-9 and H.rel_l2(p, po_) < 1e-9 and H.rel_l2(v, vo) < 1e-7


@pytest.mark.parametrize("momentum", [1, 5])
def test_channel_flow_converged_fields_match_reference_mode(gpu, oracle, mesh_path, momentum):
    """North-star parity criterion: converged u/v/w/p within 1e-6 rel-L2 of the CPU reference path.
    The oracle runs in the REFERENCE's mode (in-place diagonal reads, SURVEY Q2, frozen_diagonals=0); the
    device always uses frozen diagonals; the SIMPLE fixed point is the same."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    om, dm, a = setup(oracle, mesh_path, "channel_flow")
    from conftest import splitmix64_uniform
    n = dm.n_cells
    cc = np.asarray(a["cell_centroid"])
    u0 = H.analytical_poiseuille(cc[:, 1]) * (1 + 0.02 * splitmix64_uniform(n, 1))
    v0 = 1e-7 * splitmix64_uniform(n, 2)
    w0 = 1e-12 * splitmix64_uniform(n, 3)
    p0 = -0.01 * (1 - cc[:, 0] / 0.002) * (1 + 0.01 * splitmix64_uniform(n, 4))
    kw = dict(momentum=momentum, solver_type=BICGSTAB, iterations=50)
    uo, vo, wo, po_ = (x.copy() for x in (u0, v0, w0, p0))
    st, rep = oracle.solve_steady(om, uo, vo, wo, po_, oracle.default_settings(frozen_diagonals=0, **kw), 1000.0, 1e-3, 1500, report=True)
    assert st == 0 and rep[-1][4] < 1e-8  # velocity correction norm: converged
    s = Solver(dm, NumericalSettings.default(frozen_diagonals=1, **kw), 1000.0, 1e-3)
    s.set_fields(u0, v0, w0, p0)
    s.iterate(1500)
    u, v, w, p = s.get_fields()
    assert H.rel_l2(u, uo) < 1e-6 and H.rel_l2(p, po_) < 1e-6
    assert np.linalg.norm(v - vo) < 1e-6 * np.linalg.norm(uo) and np.linalg.norm(w - wo) < 1e-6 * np.linalg.norm(uo)
    y = np.asarray(a["cell_centroid"])[:, 1]
    assert H.rel_l2(u, H.analytical_poiseuille(y)) < (0.01 if momentum == 1 else 0.05)  # "exactly matches the analytical profile" (README.md:59-63)


def test_config2_couette_cd_bicgstab_analytical(gpu, oracle, mesh_path):
    """BASELINE config 2: couette_flow_128x64x1.msh, CD momentum, BiCGSTAB (no AMG), validated against the
    analytical u-profile (tests.rs:26-40) with the reference's 10 % criterion on mean/extremum and against
    the oracle run from the same start."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    om, dm, a = setup(oracle, mesh_path, "couette_flow_128x64x1", top_wall_velocity=5e-4, dp_dx=10.0)
    y = np.asarray(a["cell_centroid"])[:, 1]
    ua = H.analytical_poiseuille(y, dp_dx=10.0, u_top=5e-4)
    # start from the analytical field + noise so 40 iterations suffice on the device and 3 on the CPU oracle
    n = dm.n_cells
    x = np.asarray(a["cell_centroid"])[:, 0]
    from conftest import splitmix64_uniform
    u0 = ua * (1 + 0.05 * splitmix64_uniform(n, 1))
    v0 = 1e-6 * splitmix64_uniform(n, 2)
    w0 = 1e-9 * splitmix64_uniform(n, 3)
    p0 = -0.02 * (1 - x / 0.002) * (1 + 0.01 * splitmix64_uniform(n, 4))
    # 50 inner iterations = the reference's default; with 30 or fewer the reference algorithm itself diverges
    # on this mesh (oracle: |u| ~ 1e30 after 64 iterations) — "stability issues with fewer than ~50" (lib.rs:43)
    kw = dict(momentum=1, solver_type=BICGSTAB, iterations=50, frozen_diagonals=1)
    s = Solver(dm, NumericalSettings.default(**kw), 1000.0, 1e-3)
    s.set_fields(u0, v0, w0, p0)
    # one iteration with 5 BiCGSTAB steps per solve: tight agreement with the oracle at this size too
    kw1 = dict(kw, iterations=5)
    s1 = Solver(dm, NumericalSettings.default(**kw1), 1000.0, 1e-3)
    s1.set_fields(u0, v0, w0, p0)
    s1.iterate(1)
    u1, v1, w1, p1 = s1.get_fields()
    uo, vo, wo, po_ = (z.copy() for z in (u0, v0, w0, p0))
    assert oracle.solve_steady(om, uo, vo, wo, po_, oracle.default_settings(**kw1), 1000.0, 1e-3, 1)[0] == 0
    assert H.rel_l2(u1, uo) < 1e-9 and H.rel_l2(p1, po_) < 1e-9
    s.iterate(62)
    u, v, w, p = s.get_fields()
    # the reference's validation (tests.rs:118-124): max/min - 1 < 10 % for mean, min, max
    h, mu, dp, ut = 1e-3, 1e-3, 10.0, 5e-4
    u_ext = -(2 * mu * ut - h * h * dp) ** 2 / (8 * h * h * dp * mu)
    u_avg = ut / 2 - h * h / (12 * mu) * dp
    cmp = lambda a_, b_: max(a_, b_) / min(a_, b_) - 1.0
    assert cmp(u.mean(), u_avg) < 0.1
    assert cmp(u.min(), min(min(ut, 0.0), u_ext)) < 0.1
    assert H.rel_l2(u, ua) < 0.05


def test_nan_field_reports_solution_diverged(gpu, oracle, mesh_path):
    """u = v = w = p = 0 with zero RHS in w: BiCGSTAB divides 0/0 and solve_steady panics with
    "solution diverged" (solver.rs:217-221) — same status from oracle and device."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    om, dm, a = setup(oracle, mesh_path, "couette_flow_8x8x1")
    n = dm.n_cells
    kw = dict(solver_type=BICGSTAB, frozen_diagonals=1)
    z = [np.zeros(n) for _ in range(4)]
    st_o, _ = oracle.solve_steady(om, *[x.copy() for x in z], oracle.default_settings(**kw), 1000.0, 1e-3, 2)
    st_d = solve_steady(dm, *[x.copy() for x in z], NumericalSettings.default(breakdown_guard=0, **kw), 1000.0, 1e-3, 2, raise_on_error=False)
    assert st_o == st_d == 1
    # product default (breakdown_guard=1): the zero-RHS solves freeze instead of dividing 0/0
    st_g = solve_steady(dm, *z, NumericalSettings.default(**kw), 1000.0, 1e-3, 2, raise_on_error=False)
    assert st_g == 0 and all(np.isfinite(x).all() for x in z) and np.abs(z[0]).max() > 0


@pytest.mark.parametrize("momentum", [1])
def test_channel_flow_converged_fields_multigrid_reference_mode(gpu, oracle, mesh_path, momentum):
    """The same north-star criterion with the reference's DEFAULT solver — solver_type = Multigrid, 50 smoother
    iterations, Jacobi preconditioner (lib.rs:76-85), the configuration the benchmark runs: device (tree reductions,
    frozen diagonals) against the oracle in the reference's in-place mode, converged u/v/w/p within 1e-6 rel-L2."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    from conftest import splitmix64_uniform
    om, dm, a = setup(oracle, mesh_path, "channel_flow")
    n = dm.n_cells
    cc = np.asarray(a["cell_centroid"])
    u0 = H.analytical_poiseuille(cc[:, 1]) * (1 + 0.02 * splitmix64_uniform(n, 1))
    v0 = 1e-7 * splitmix64_uniform(n, 2)
    w0 = 1e-12 * splitmix64_uniform(n, 3)
    p0 = -0.01 * (1 - cc[:, 0] / 0.002) * (1 + 0.01 * splitmix64_uniform(n, 4))
    kw = dict(momentum=momentum, solver_type=MULTIGRID, iterations=50)
    uo, vo, wo, po_ = (x.copy() for x in (u0, v0, w0, p0))
    st, rep = oracle.solve_steady(om, uo, vo, wo, po_, oracle.default_settings(frozen_diagonals=0, **kw), 1000.0, 1e-3, 1500, report=True)
    assert st == 0 and rep[-1][4] < 1e-8  # velocity correction norm: converged
    s = Solver(dm, NumericalSettings.default(frozen_diagonals=1, **kw), 1000.0, 1e-3)
    s.set_fields(u0, v0, w0, p0)
    s.iterate(1500)
    u, v, w, p = s.get_fields()
    assert H.rel_l2(u, uo) < 1e-6 and H.rel_l2(p, po_) < 1e-6
    assert np.linalg.norm(v - vo) < 1e-6 * np.linalg.norm(uo) and np.linalg.norm(w - wo) < 1e-6 * np.linalg.norm(uo)
    y = cc[:, 1]
    assert H.rel_l2(u, H.analytical_poiseuille(y)) < 0.01


def test_couette_converged_fields_multigrid_golden(gpu, oracle, mesh_path):
    """couette_flow_128x64x1.msh (8001 cells, moving top wall) with the reference's default stack: the device's converged
    fields against tests/golden/couette_multigrid_converged.npz — the oracle's converged fields in the reference's
    in-place mode (1500 iterations, about three minutes of CPU, hence stored; tests/test_golden_cpu.py checks that the
    stored fields are a fixed point of the oracle)."""
    import os
    from conftest import GOLDEN
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    sys_path = os.path.join(GOLDEN, "make_golden_couette_multigrid.py")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_couette_multigrid", sys_path)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    om, dm, a = setup(oracle, mesh_path, "couette_flow_128x64x1", top_wall_velocity=5e-4, dp_dx=10.0)
    g = np.load(os.path.join(GOLDEN, "couette_multigrid_converged.npz"))
    u0, v0, w0, p0 = gen.start_fields(np.asarray(a["cell_centroid"]))
    s = Solver(dm, NumericalSettings.default(momentum=1, solver_type=MULTIGRID, iterations=50, frozen_diagonals=1), 1000.0, 1e-3)
    s.set_fields(u0, v0, w0, p0)
    st, rep = s.iterate(int(g["iterations"]), report=True, raise_on_error=False)
    assert st == 0 and rep[-1][6] < 1e-8
    u, v, w, p = s.get_fields()
    assert H.rel_l2(u, g["u"]) < 1e-6 and H.rel_l2(p, g["p"]) < 1e-6
    assert np.linalg.norm(v - g["v"]) < 1e-6 * np.linalg.norm(g["u"]) and np.linalg.norm(w - g["w"]) < 1e-6 * np.linalg.norm(g["u"])
    y = np.asarray(a["cell_centroid"])[:, 1]
    assert H.rel_l2(u, H.analytical_poiseuille(y, dp_dx=10.0, u_top=5e-4)) < 0.01
