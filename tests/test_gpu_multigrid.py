"""Device Multigrid arm (exact greedy aggregation, Galerkin product, V-recursion) vs the CPU oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import fv_like_matrix, splitmix64_uniform, unit_test_system

pytestmark = pytest.mark.gpu

MULTIGRID, BICGSTAB = 2, 3
PRE_NONE, PRE_JACOBI = 0, 1


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def oracle_partner(oracle, a):
    """strongest_unmerged_neighbor per row from the oracle's R (linear_algebra.rs:53-58)."""
    import scipy.sparse as sp
    n = a.shape[0]
    R = oracle.build_restriction_matrix(oracle.Csr.from_scipy(a)).to_scipy()
    return R


@pytest.mark.parametrize("shape", [(2, 1, 1), (5, 1, 1), (7, 5, 3), (16, 16, 1), (33, 9, 4), (64, 40, 3)])
def test_aggregation_and_galerkin_bit_exact(gpu, oracle, shape):
    """The parallel fixed-point iteration must land on the SEQUENTIAL greedy pairing of the reference, and the
    LDS Galerkin product on nalgebra-sparse's (R a) R^T bit for bit (same summation order)."""
    from orc_amd.linear_algebra import amg_coarsen
    a = fv_like_matrix(*shape)
    n = a.shape[0]
    partner, ac, rounds = amg_coarsen(a)
    A = oracle.Csr.from_scipy(a)
    R = oracle.build_restriction_matrix(A)
    # rebuild R from the device's partners exactly as the reference pushes triplets
    rows, cols = [], []
    for i in range(n):
        if partner[i] >= 0:
            rows += [i // 2, i // 2]
            cols += [i, int(partner[i])]
    Rd = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=((n + 1) // 2, n)).tocsr()
    Rd.sum_duplicates()
    assert abs(Rd - R.to_scipy()).max() == 0
    ref = R.matmul(A).matmul(R.transpose())
    rp, ci, v = ref.arrays()
    assert ac.shape == ref.shape
    assert np.array_equal(ac.indptr, rp) and np.array_equal(ac.indices, ci)
    assert np.array_equal(ac.data, v)
    assert rounds >= 1


@pytest.mark.parametrize("env", [{"ORC_AMG_DA": "0"}, {"ORC_AMG_DA_STEPS": "3"}, {"ORC_AMG_DA_GROUP": "16"}, {"ORC_AMG_DA_GROUP": "8", "ORC_AMG_L0_MIRROR": "0"},
                                 {"ORC_GALERKIN_GROUPS": "64,64,64,64"}, {"ORC_GALERKIN_GROUPS": "16,16,16,32"}])
def test_set_up_forms_agree(gpu, monkeypatch, env):
    """Every form of the set-up lands on the same pairing and the same coarse operator, bit for bit.  The pairing [r05] is by deferred
    acceptance (da_first_k + da_chase_k); ORC_AMG_DA=0 is the fallback (slice-sequential sweeps until one changes nothing), also reached by
    cutting every chain of proposals after three steps (ORC_AMG_DA_STEPS=3); chains followed by 16 and by 8 lanes (default: 4 or 8 by row
    length), without the fine level's row mirror; the Galerkin merge with every group size."""
    from orc_amd.linear_algebra import amg_coarsen
    results = []
    for form in ({}, env):
        for k, v in form.items():
            monkeypatch.setenv(k, v)
        per_shape = []
        for shape in ((33, 9, 4), (64, 40, 3), (40, 40, 12)):
            partner, ac, rounds = amg_coarsen(fv_like_matrix(*shape))
            _, ac2, _ = amg_coarsen(ac)  # a coarse level's rows (15 - 30 entries) through the set-up as well
            per_shape.append((partner, ac, ac2))
        results.append(per_shape)
        for k in form:
            monkeypatch.delenv(k)
    for (p0, a0, b0), (p1, a1, b1) in zip(*results):
        assert np.array_equal(p0, p1)
        for x, y in ((a0, a1), (b0, b1)):
            assert np.array_equal(x.indptr, y.indptr) and np.array_equal(x.indices, y.indices) and np.array_equal(x.data, y.data)


def test_aggregation_dependency_chain(gpu, oracle):
    """1-D chain with monotone coefficients: every row's choice depends on its predecessor's, i.e. the
    longest possible dependency chain (n/2 rounds)."""
    from orc_amd.linear_algebra import amg_coarsen
    n = 301
    lo = -(1.0 + 0.001 * np.arange(n - 1))   # a[i, i-1]: stronger than the upper neighbour
    up = -0.5 * np.ones(n - 1)
    a = sp.diags([lo, 3.0 * np.ones(n), up], [-1, 0, 1], format="csr")
    partner, ac, rounds = amg_coarsen(a)
    R = oracle.build_restriction_matrix(oracle.Csr.from_scipy(a)).to_scipy()
    rows, cols = [], []
    for i in range(n):
        if partner[i] >= 0:
            rows += [i // 2, i // 2]
            cols += [i, int(partner[i])]
    Rd = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=((n + 1) // 2, n)).tocsr()
    Rd.sum_duplicates()
    assert abs(Rd - R).max() == 0
    assert rounds >= 1  # (deferred acceptance: one lane follows the whole chain; r04's sweeps needed one round per slice crossed)


def test_aggregation_asymmetric_pattern(gpu, oracle):
    """The reference's unit-test matrix has a structurally asymmetric pattern: general fall-back path."""
    from orc_amd.linear_algebra import amg_coarsen
    a, b, _ = unit_test_system(57)
    partner, ac, rounds = amg_coarsen(a)
    A = oracle.Csr.from_scipy(a)
    R = oracle.build_restriction_matrix(A)
    ref = R.matmul(A).matmul(R.transpose()).to_scipy()
    assert abs(ac - ref).max() == 0 and np.array_equal(ac.indptr, ref.indptr)


@pytest.mark.parametrize("shape,iters", [((7, 5, 3), 3), ((20, 17, 9), 4), ((64, 40, 12), 5)])
def test_multigrid_arm_vs_oracle(gpu, oracle, shape, iters):
    """Multigrid arm end to end (nested re-scaling Q4, r' recursion Q5, weight-2 rows Q6) with few smoother
    iterations: 1e-8 rel-L2 (the hierarchy is bit-identical; only dot association differs)."""
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(*shape)
    n = a.shape[0]
    b = a @ splitmix64_uniform(n, 7)
    x0 = 0.1 * splitmix64_uniform(n, 8)
    x, xo = x0.copy(), x0.copy()
    iterative_solve(a, b, x, iters, MULTIGRID, 0.5, 1e-3, PRE_JACOBI)
    assert oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, iters, MULTIGRID, 0.5, 1e-3, PRE_JACOBI) == 0
    assert rel(x, xo) < 1e-8


def test_multigrid_diverged_status(gpu, oracle):
    """the reference's disabled unit test: Multigrid on the 100x100 system runs the fixed-count BiCGSTAB into 0/0
    -> "Multigrid diverged" (linear_algebra.rs:103-105); same status from device (guard off) and oracle."""
    from orc_amd.linear_algebra import iterative_solve, set_breakdown_guard
    a, b, sol = unit_test_system()
    xo = np.zeros(len(b))
    sto = oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 50, MULTIGRID, 0.5, 1e-9, PRE_JACOBI)
    try:
        set_breakdown_guard(False)
        st = iterative_solve(a, b, np.zeros(len(b)), 50, MULTIGRID, 0.5, 1e-9, PRE_JACOBI, raise_on_error=False)
    finally:
        set_breakdown_guard(True)
    assert sto == 2
    assert st == 2


def test_solve_steady_default_stack_one_iteration(gpu, oracle, mesh_path):
    """NumericalSettings::default() (Multigrid + Jacobi precond, CD1, Rhie-Chow, SecondOrder) with 5 smoother
    iterations: one full SIMPLE iteration on channel_flow.msh agrees with the oracle to 1e-8."""
    import helpers as H
    from orc_amd.mesh import Mesh, MeshArrays
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    om = oracle.Mesh.read(mesh_path("channel_flow"))
    H.channel_bcs(om)
    a = MeshArrays(om.arrays())
    dm = Mesh(a)
    u, v, w, p = H.seeded_fields(a, seed=9, scale_u=4e-4, w_zero=False)
    uo, vo, wo, po_ = (x.copy() for x in (u, v, w, p))
    kw = dict(iterations=5, frozen_diagonals=1)
    assert oracle.solve_steady(om, uo, vo, wo, po_, oracle.default_settings(**kw), 1000.0, 1e-3, 1)[0] == 0
    solve_steady(dm, u, v, w, p, NumericalSettings.default(**kw), 1000.0, 1e-3, 1)
    assert H.rel_l2(u, uo) < 1e-8 and H.rel_l2(p, po_) < 1e-8


def test_default_stack_is_reproducible_run_to_run(gpu, oracle, mesh_path):
    """Reductions fold their partials in a fixed order and the aggregation's fixed point is unique, so two runs of the
    same four SIMPLE iterations (default stack) give identical bits even though the aggregation rounds use atomics."""
    import helpers as H
    from orc_amd.mesh import Mesh, MeshArrays
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    om = oracle.Mesh.read(mesh_path("channel_flow"))
    H.channel_bcs(om)
    a = MeshArrays(om.arrays())
    dm = Mesh(a)
    u, v, w, p = H.seeded_fields(a, seed=4, scale_u=4e-4)
    runs = []
    for _ in range(2):
        s = Solver(dm, NumericalSettings.default(iterations=10), 1000.0, 1e-3)
        s.set_fields(u, v, w, p)
        s.iterate(4)
        runs.append(s.get_fields())
    for x, y in zip(*runs):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("solver", [MULTIGRID, BICGSTAB, 17, 18])  # 17 / 18: GS-preconditioned BiCGSTAB / Multigrid with GS smoother
def test_stream_scheduling_does_not_change_a_bit(gpu, oracle, mesh_path, monkeypatch, solver):
    """The three-system momentum solve (u, v, w in lock-step on their shared pattern: one column stream, interleaved vectors),
    the momentum lanes (u, v, w on three streams / host threads), the two-stream Multigrid arm (set-up beside
    smoothing) and the early p' hierarchy (built beside the momentum solves) only reorder independent work, and the
    sibling pairing (v and w take u's fine-level pairing when it verifies as their own) only skips work whose result is
    known: three SIMPLE iterations give identical bits with any of them switched off."""
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    import helpers as H
    a = set_channel_bcs(hex_channel(24, 16, 10))
    s = NumericalSettings.default(momentum=5, solver_type=solver, iterations=6, momentum_relaxation=0.1, pressure_relaxation=0.001)
    out = []
    # the first and the last but one run the three momentum systems in lock-step on their shared pattern (ORC_TRIPLE_MOMENTUM,
    # the default); the others the per-system lanes or the plain sequential loop
    for lanes, two, early, sibling, triple in (("1", "1", "1", "1", "1"), ("0", "0", "0", "1", "0"), ("1", "0", "0", "1", "0"), ("0", "1", "0", "1", "0"),
                                               ("0", "0", "1", "1", "0"), ("1", "1", "0", "1", "0"), ("1", "1", "1", "0", "0"), ("1", "1", "1", "1", "0"),
                                               ("1", "1", "0", "0", "1"), ("0", "0", "0", "0", "0")):
        monkeypatch.setenv("ORC_TRIPLE_MOMENTUM", triple)
        monkeypatch.setenv("ORC_CONCURRENT_MOMENTUM", lanes)
        monkeypatch.setenv("ORC_TWO_STREAM_MULTIGRID", two)
        monkeypatch.setenv("ORC_EARLY_P_HIERARCHY", early)
        monkeypatch.setenv("ORC_AMG_SIBLING", sibling)
        dm = Mesh(a)
        u, v, w, p = H.seeded_fields(a, seed=8)
        solve_steady(dm, u, v, w, p, s, 1000.0, 1e-3, 3)
        out.append((u, v, w, p))
    assert np.isfinite(out[0][0]).all()
    for other in out[1:]:
        for x, y in zip(out[0], other):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("triple", ["1", "0"])
def test_cache_policy_of_the_matrix_loads_does_not_change_a_bit(gpu, monkeypatch, triple):
    """The products load their matrix streams with the non-temporal hint when the stream exceeds 128 MB (launch_spmv; kernel
    instantiations of their own) — a size no other test of this file reaches.  Forced on and forced off on a small channel,
    lock-step and per-system momentum solves: three default-stack SIMPLE iterations, identical bits."""
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    import helpers as H
    a = set_channel_bcs(hex_channel(40, 24, 16))  # level 2 has 3 840 rows of ~30 entries: window products on the coarse levels
    s = NumericalSettings.default(momentum=5, solver_type=MULTIGRID, iterations=8, momentum_relaxation=0.1, pressure_relaxation=0.001)
    monkeypatch.setenv("ORC_TRIPLE_MOMENTUM", triple)
    out = []
    for nt in ("0", "1"):
        monkeypatch.setenv("ORC_SPMV_NT", nt)
        dm = Mesh(a)
        u, v, w, p = H.seeded_fields(a, seed=11)
        solve_steady(dm, u, v, w, p, s, 1000.0, 1e-3, 3)
        out.append((u, v, w, p))
    assert np.isfinite(out[0][0]).all()
    for x, y in zip(*out):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("triple", ["1", "0"])
def test_shared_jacobi_scaling_of_a_levels_two_smoothing_solves_does_not_change_a_bit(gpu, monkeypatch, triple):
    """multigrid_solve's pre- and post-smoothing solves of a level (linear_algebra.rs:87-96, :123-132) scale the same coarse matrix by the
    same inverse diagonal (:159-166).  [r05] The second reuses the first one's inverse diagonal and scaled values (ScaledOperator,
    linalg.hpp); ORC_AMG_SHARED_SCALING=0 computes both twice, as r04 did.  Lock-step and per-system momentum solves, window products on
    the coarse levels (8 iterations: values materialised) and entry-by-entry scaling (3 iterations: not materialised): identical bits."""
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    import helpers as H
    a = set_channel_bcs(hex_channel(40, 24, 16))
    monkeypatch.setenv("ORC_TRIPLE_MOMENTUM", triple)
    for iterations in (8, 3):
        s = NumericalSettings.default(momentum=5, solver_type=MULTIGRID, iterations=iterations, momentum_relaxation=0.1, pressure_relaxation=0.001)
        out = []
        for shared in ("1", "0"):
            monkeypatch.setenv("ORC_AMG_SHARED_SCALING", shared)
            dm = Mesh(a)
            u, v, w, p = H.seeded_fields(a, seed=12)
            solve_steady(dm, u, v, w, p, s, 1000.0, 1e-3, 3)
            out.append((u, v, w, p))
        assert np.isfinite(out[0][0]).all()
        for x, y in zip(*out):
            assert np.array_equal(x, y)


def test_in_launch_fold_of_the_window_products_is_reproducible(gpu, oracle, monkeypatch):
    """[r05] Window products launch one workgroup per ~12 000 entries (one 256-row block here); every workgroup leaves its partial sums in the
    level's scratch and whichever arrives last adds them up in index order (spmv_xwin_k, XWinDev::fold_scratch): the sums must not depend on the
    order of arrival.  The Multigrid arm on a matrix whose levels 2 and 3 have such blocks (8 400 and 18 000 entries per block), five smoother
    iterations: identical bits run after run, the oracle's result to 1e-8, and with ORC_XWIN_WG_PER_BLOCK=0 (2 048 persistent workgroups, one
    partial sum each: another association of the same dot products) the same solution to the rounding of those sums."""
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(64, 40, 12)
    n = a.shape[0]
    b = a @ splitmix64_uniform(n, 17)
    x0 = 0.1 * splitmix64_uniform(n, 18)
    out = []
    for mode in ("1", "1", "0"):
        monkeypatch.setenv("ORC_XWIN_WG_PER_BLOCK", mode)
        x = x0.copy()
        iterative_solve(a, b, x, 5, MULTIGRID, 0.5, 1e-3, PRE_JACOBI)
        out.append(x)
    xo = x0.copy()
    assert oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 5, MULTIGRID, 0.5, 1e-3, PRE_JACOBI) == 0
    assert np.array_equal(out[0], out[1])
    assert rel(out[0], xo) < 1e-8 and rel(out[2], xo) < 1e-8
    assert rel(out[0], out[2]) < 1e-9


def test_last_slice_dead_lanes_regression(gpu, oracle, mesh_path):
    """Regression (round 2, fix 5d036f6): channel_flow.msh has 1008 = 15 x 64 + 48 cells, so the last slice of every level has
    dead lanes; those lanes once gathered through never-written columns of the device-packed coarse operators and the process
    aborted in the second solver of a run.  Two solvers in one process, four default-stack iterations each: statuses 0,
    finite fields, identical results."""
    from orc_amd.mesh import Mesh, MeshArrays
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    import helpers as H
    om = oracle.Mesh.read(mesh_path("channel_flow"))
    H.channel_bcs(om)
    a = MeshArrays(om.arrays())
    dm = Mesh(a)
    u, v, w, p = H.seeded_fields(a, seed=4, scale_u=4e-4)
    out = []
    for _ in range(2):
        s = Solver(dm, NumericalSettings.default(iterations=10), 1000.0, 1e-3)
        s.set_fields(u, v, w, p)
        for _it in range(4):
            assert s.iterate(1, raise_on_error=False) == 0
        out.append(s.get_fields())
    for x, y in zip(*out):
        assert np.isfinite(x).all() and np.array_equal(x, y)
