"""Multicolour Gauss-Seidel extension (K6; no reference counterpart: SURVEY Q8) vs a numpy emulation of the same sweep."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import fv_like_matrix, splitmix64_uniform

pytestmark = pytest.mark.gpu

MULTICOLOR_GS, BICGSTAB_GS, MULTIGRID_GS, BICGSTAB, MULTIGRID = 16, 17, 18, 3, 2


def gs_numpy(a, b, x, sweeps, omega, colors, scale=None):
    """x_i = x_i (1 - w) + w (b_i - sum_{j != i} a_ij x_j) / a_ii  (linear_algebra.rs:225-239), rows in colour order,
    row sums in ascending-column order with the literal 0 for j == i."""
    a = a.tocsr()
    x = x.copy()
    order = np.argsort(colors, kind="stable")
    for _ in range(sweeps):
        for i in order:
            s = 0.0
            d = None
            for q in range(a.indptr[i], a.indptr[i + 1]):
                j = a.indices[q]
                v = a.data[q] if scale is None else scale[i] * a.data[q]
                if j == i:
                    d = v
                    s += 0.0
                else:
                    s += v * x[j]
            bi = b[i] if scale is None else 0.0 + scale[i] * b[i]
            x[i] = x[i] * (1.0 - omega) + omega * (bi - s) / d
    return x


@pytest.mark.parametrize("shape", [(5, 4, 3), (17, 9, 2), (33, 20, 7)])
def test_coloring_is_proper_and_small(gpu, shape):
    from orc_amd.linear_algebra import debug_coloring
    a = fv_like_matrix(*shape)
    colors, nc = debug_coloring(a)
    coo = a.tocoo()
    off = coo.row != coo.col
    assert np.all(colors[coo.row[off]] != colors[coo.col[off]])
    assert colors.min() == 0 and nc == colors.max() + 1 and nc <= 12  # 7-point pattern: 2 suffice in theory


@pytest.mark.parametrize("precond", [0, 1])
def test_gs_sweeps_bit_exact_vs_numpy(gpu, precond):
    from orc_amd.linear_algebra import debug_coloring, iterative_solve
    a = fv_like_matrix(9, 7, 5)
    n = a.shape[0]
    b = splitmix64_uniform(n, 5)
    x0 = splitmix64_uniform(n, 6)
    colors, _ = debug_coloring(a)
    x = x0.copy()
    iterative_solve(a, b, x, 4, MULTICOLOR_GS, 0.8, 1e-3, precond)
    scale = (1.0 / a.diagonal()) if precond else None
    ref = gs_numpy(a, b, x0, 4, 0.8, colors, scale)
    assert np.array_equal(x, ref)


def test_gs_converges_and_reports_missing_diagonal(gpu):
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(12, 10, 8)
    n = a.shape[0]
    xs = splitmix64_uniform(n, 2)
    b = a @ xs
    x = np.zeros(n)
    iterative_solve(a, b, x, 200, MULTICOLOR_GS, 1.0, 1e-3, 1)
    assert np.linalg.norm(x - xs) < 1e-6 * np.linalg.norm(xs)
    a2 = a.tolil()
    a2[3, 3] = 0
    a2 = a2.tocsr()
    a2.eliminate_zeros()
    st = iterative_solve(a2, b, np.zeros(n), 2, MULTICOLOR_GS, 1.0, 1e-3, 0, raise_on_error=False)
    assert st == 6


def test_gs_preconditioned_bicgstab_beats_plain(gpu):
    """Same recurrences as linear_algebra.rs:247-269 with p^ = M^-1 p, s^ = M^-1 s (M^-1 = one GS sweep): fewer
    iterations to a given residual than the reference's Jacobi-scaled BiCGSTAB."""
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(30, 24, 12)
    n = a.shape[0]
    xs = splitmix64_uniform(n, 3)
    b = a @ xs
    res = {}
    for method in (BICGSTAB, BICGSTAB_GS):
        x = np.zeros(n)
        iterative_solve(a, b, x, 8, method, 0.5, 1e-3, 1)
        res[method] = np.linalg.norm(a @ x - b) / np.linalg.norm(b)
    assert res[BICGSTAB_GS] < 0.2 * res[BICGSTAB] and res[BICGSTAB_GS] < 1e-3


def test_multigrid_with_gs_smoother(gpu):
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(30, 24, 12)
    n = a.shape[0]
    xs = splitmix64_uniform(n, 4)
    b = a @ xs
    x = np.zeros(n)
    iterative_solve(a, b, x, 10, MULTIGRID_GS, 1.0, 1e-3, 1)
    assert np.isfinite(x).all() and np.linalg.norm(a @ x - b) < 0.1 * np.linalg.norm(b)


def test_solve_steady_with_gs_preconditioned_bicgstab(gpu, oracle, mesh_path):
    """BASELINE config 3's solver choice on channel_flow.msh: QUICK + Rhie-Chow, GS-preconditioned BiCGSTAB; converged u within
    1e-6 rel-L2 of the oracle run with the reference's BiCGSTAB (same fixed point, different inner solver)."""
    import helpers as H
    from orc_amd.mesh import Mesh, MeshArrays
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    om = oracle.Mesh.read(mesh_path("channel_flow"))
    H.channel_bcs(om)
    a = MeshArrays(om.arrays())
    dm = Mesh(a)
    n = dm.n_cells
    cc = np.asarray(a["cell_centroid"])
    u0 = H.analytical_poiseuille(cc[:, 1]) * (1 + 0.02 * splitmix64_uniform(n, 1))
    v0 = 1e-7 * splitmix64_uniform(n, 2)
    w0 = 1e-12 * splitmix64_uniform(n, 3)
    p0 = -0.01 * (1 - cc[:, 0] / 0.002)
    uo, vo, wo, po_ = (x.copy() for x in (u0, v0, w0, p0))
    st, rep = oracle.solve_steady(om, uo, vo, wo, po_, oracle.default_settings(momentum=4, solver_type=oracle.BICGSTAB, frozen_diagonals=0),
                                  1000.0, 1e-3, 1500, report=True)
    assert st == 0
    # 30 inner iterations: with 15 the outer loop itself diverges (the reference's SIMPLE needs well-converged inner solves)
    s = Solver(dm, NumericalSettings.default(momentum=4, solver_type=BICGSTAB_GS, iterations=30), 1000.0, 1e-3)
    s.set_fields(u0, v0, w0, p0)
    s.iterate(1500)
    u, v, w, p = s.get_fields()
    assert H.rel_l2(u, uo) < 1e-6 and H.rel_l2(p, po_) < 1e-6


def test_slot_space_solver_follows_the_row_space_recurrences(gpu, monkeypatch):
    """[r04] The GS-preconditioned BiCGSTAB now lives in the colour-sorted numbering (gs.hip: gsx_*): vectors permuted once, the
    preconditioner sweep from zero without a zero fill (columns of the colour's own and later colours are skipped: exact zeros),
    sums folded by their consumers.  Row sums and sweeps keep their bits; only the partial sums group rows differently, so after a few
    iterations the two forms agree to rounding (ORC_GS_SLOTSPACE=0 = r03's row-space form), with and without the Jacobi scaling, and
    both reach the solution."""
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(30, 24, 12)
    n = a.shape[0]
    xs = splitmix64_uniform(n, 3)
    b = a @ xs
    for precond in (0, 1):
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("ORC_GS_SLOTSPACE", mode)
            x = 0.01 * splitmix64_uniform(n, 9)
            iterative_solve(a, b, x, 5, BICGSTAB_GS, 1.0, 1e-3, precond)
            out[mode] = x
        assert np.isfinite(out["1"]).all()
        assert np.linalg.norm(out["1"] - out["0"]) < 1e-10 * np.linalg.norm(out["0"]), precond
        x = np.zeros(n)
        monkeypatch.setenv("ORC_GS_SLOTSPACE", "1")
        iterative_solve(a, b, x, 40, BICGSTAB_GS, 1.0, 1e-3, precond)
        assert np.linalg.norm(x - xs) < 1e-9 * np.linalg.norm(xs)


def test_gs_momentum_systems_in_lock_step_are_bit_identical_to_their_own_solves(gpu, monkeypatch):
    """u, v, w per colour in ONE launch (interleaved slot-space vectors, three value streams on the shared colour-sorted pattern):
    whole SIMPLE iterations with BASELINE configs[2]'s solver are bit-identical to the runs that solve the systems one at a time
    (ORC_TRIPLE_MOMENTUM=0: three lanes, same kernels with S = 1), frozen systems included (w == 0 exactly on the one-cell-deep
    channel: its solve breaks down at once and the guard freezes it while u and v carry on)."""
    import helpers as H
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    for shape, w_zero in (((24, 16, 10), False), ((40, 30, 1), True)):
        a = set_channel_bcs(hex_channel(*shape))
        s = NumericalSettings.default(momentum=4, solver_type=BICGSTAB_GS, iterations=12)
        out = []
        for triple in ("1", "0"):
            monkeypatch.setenv("ORC_TRIPLE_MOMENTUM", triple)
            u, v, w, p = H.seeded_fields(a, seed=8, scale_u=4e-4, w_zero=w_zero)
            solve_steady(Mesh(a), u, v, w, p, s, 1000.0, 1e-3, 3)
            out.append((u, v, w, p))
        assert np.isfinite(out[0][0]).all()
        for x, y in zip(*out):
            assert np.array_equal(x, y), shape
