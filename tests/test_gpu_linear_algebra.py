"""HIP iterative solvers (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import fv_like_matrix, splitmix64_uniform, unit_test_system

pytestmark = pytest.mark.gpu

JACOBI, MULTIGRID, BICGSTAB = 1, 2, 3
PRE_NONE, PRE_JACOBI = 0, 1


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 1, 1), (7, 5, 3), (64, 1, 1), (65, 3, 1), (40, 33, 17)])
def test_spmv_bit_exact(gpu, oracle, shape):
    """y = A x accumulates each row in ascending-column order from 0.0: bit-identical to the CPU product."""
    from orc_amd.linear_algebra import csr_spmv
    a = fv_like_matrix(*shape)
    x = splitmix64_uniform(a.shape[0], 3)
    y, _ = csr_spmv(a, x)
    ref = oracle.Csr.from_scipy(a).spmv(x)
    assert np.array_equal(y, ref)


def test_spmv_ragged_rows_and_empty(gpu, oracle):
    from orc_amd.linear_algebra import csr_spmv
    rng = np.random.default_rng(5)
    n = 517
    dens = sp.random(n, n, density=0.02, random_state=rng, format="csr")
    dens = dens + sp.diags(np.where(np.arange(n) % 7 == 0, 0.0, 1.0))  # some rows without diagonal
    dens = dens.tocsr()
    dens[11, :] = 0  # an empty row
    dens.eliminate_zeros()
    dens.sort_indices()
    x = splitmix64_uniform(n, 8)
    y, _ = csr_spmv(dens, x)
    assert np.array_equal(y, oracle.Csr.from_scipy(dens).spmv(x))
    # n = 0
    y0, _ = csr_spmv(sp.csr_matrix((0, 0)), np.zeros(0))
    assert y0.shape == (0,)


def test_validate_iterative_solvers_kat(gpu, oracle):
    """The reference's only #[test] (linear_algebra.rs:309-378) through the HIP path."""
    from orc_amd.linear_algebra import iterative_solve
    a, b, sol = unit_test_system()
    n = len(b)
    x = np.zeros(n)
    xo = np.zeros(n)
    A = oracle.Csr.from_scipy(a)
    for method in (JACOBI, BICGSTAB):
        iterative_solve(a, b, x, 50, method, 0.5, 1e-3 / n ** 3, PRE_JACOBI)
        assert oracle.iterative_solve(A, b, xo, 50, method, 0.5, 1e-3 / n ** 3, PRE_JACOBI) == 0
        assert np.linalg.norm(a @ x - b) < 1e-3  # the reference's assertion
        # Jacobi is reduction-free per row: bit-exact; BiCGSTAB differs only by dot-product association
        if method == JACOBI:
            assert np.array_equal(x, xo)
        else:
            assert np.abs(x - sol).max() < 1e-10 and rel(x, xo) < 1e-10


@pytest.mark.parametrize("precond", [PRE_NONE, PRE_JACOBI])
@pytest.mark.parametrize("shape,iters", [((7, 5, 3), 8), ((40, 33, 17), 12), ((130, 9, 5), 5)])
def test_bicgstab_vs_oracle(gpu, oracle, shape, iters, precond):
    """fixed-count BiCGSTAB (linear_algebra.rs:247-269); tolerance 1e-9 rel-L2: same recurrences, dot
    products associate differently (wave tree vs nalgebra's 8-accumulator order)."""
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(*shape)
    n = a.shape[0]
    b = splitmix64_uniform(n, 11)
    x0 = splitmix64_uniform(n, 12)
    x, xo = x0.copy(), x0.copy()
    iterative_solve(a, b, x, iters, BICGSTAB, 0.5, 1e-3, precond)
    assert oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, iters, BICGSTAB, 0.5, 1e-3, precond) == 0
    assert rel(x, xo) < 1e-9


def test_bicgstab_zero_iterations_and_nan_propagation(gpu, oracle):
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(5, 4, 3)
    n = a.shape[0]
    b = np.zeros(n)
    x = np.zeros(n)
    iterative_solve(a, b, x, 0, BICGSTAB, 0.5, 1e-3, PRE_JACOBI)
    assert np.array_equal(x, np.zeros(n))
    # b = 0, x = 0: rho = 0 and r_hat.nu = 0 -> alpha = 0/0: the reference has no breakdown guard
    from orc_amd.linear_algebra import set_breakdown_guard
    try:
        set_breakdown_guard(False)  # reference behaviour
        iterative_solve(a, b, x, 2, BICGSTAB, 0.5, 1e-3, PRE_JACOBI)
    finally:
        set_breakdown_guard(True)
    xo = np.zeros(n)
    oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 2, BICGSTAB, 0.5, 1e-3, PRE_JACOBI)
    assert np.isnan(xo).all() and np.isnan(x).all()
    # with the guard (product default) the solve freezes: x keeps its (exact) value
    x = np.zeros(n)
    iterative_solve(a, b, x, 2, BICGSTAB, 0.5, 1e-3, PRE_JACOBI)
    assert np.array_equal(x, np.zeros(n))


def test_breakdown_guard_is_inert_without_breakdown_and_stops_past_convergence(gpu, oracle):
    """guard on/off give identical bits while no denominator is 0; far past convergence (where the
    reference divides 0/0) the guarded solve returns the converged iterate."""
    from orc_amd.linear_algebra import iterative_solve, set_breakdown_guard
    a = fv_like_matrix(7, 5, 3)
    n = a.shape[0]
    xs = splitmix64_uniform(n, 4)
    b = a @ xs
    res = []
    for on in (True, False):
        try:
            set_breakdown_guard(on)
            x = np.zeros(n)
            iterative_solve(a, b, x, 10, BICGSTAB, 0.5, 1e-3, PRE_JACOBI)
            res.append(x)
        finally:
            set_breakdown_guard(True)
    assert np.array_equal(res[0], res[1])
    x = np.zeros(n)
    iterative_solve(a, b, x, 2000, BICGSTAB, 0.5, 1e-3, PRE_JACOBI)
    assert np.isfinite(x).all() and rel(x, xs) < 1e-9


@pytest.mark.parametrize("precond", [PRE_NONE, PRE_JACOBI])
def test_jacobi_vs_oracle_bit_exact_and_sweep_count(gpu, oracle, precond):
    """Jacobi arm (linear_algebra.rs:172-218): per-row arithmetic only, so x is bit-identical; the
    convergence break (skipping sweeps 0 and 1, Q7) must trigger on the same sweep."""
    from orc_amd.linear_algebra import iterative_solve, last_jacobi_sweeps
    a = fv_like_matrix(9, 7, 5)
    n = a.shape[0]
    b = a @ splitmix64_uniform(n, 4)
    for thr, iters in ((1e-30, 25), (0.2, 50), (0.9, 50)):
        x, xo = np.zeros(n), np.zeros(n)
        iterative_solve(a, b, x, iters, JACOBI, 0.7, thr, precond)
        assert oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, iters, JACOBI, 0.7, thr, precond) == 0
        assert last_jacobi_sweeps() == oracle.lib().or_last_jacobi_sweeps()
        assert np.array_equal(x, xo)


def test_jacobi_divergence_statuses(gpu, oracle):
    from orc_amd.linear_algebra import iterative_solve
    n = 50
    a = sp.diags([np.full(n - 1, 5.0), np.ones(n), np.full(n - 1, 5.0)], [-1, 0, 1], format="csr")
    b = np.ones(n)
    x, xo = np.zeros(n), np.zeros(n)
    st = iterative_solve(a, b, x, 200, JACOBI, 1.0, 1e-30, PRE_NONE, raise_on_error=False)
    sto = oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 200, JACOBI, 1.0, 1e-30, PRE_NONE)
    assert st == sto == 4  # "Diverged - max solution value > 10^10"
    x = np.full(n, np.nan)
    st = iterative_solve(a, b, x, 3, JACOBI, 0.5, 1e-3, PRE_NONE, raise_on_error=False)
    assert st == 3  # "diverged"
    # missing diagonal -> get(i,i) panics in the reference
    a2 = a.tolil()
    a2[7, 7] = 0
    a2 = a2.tocsr()
    a2.eliminate_zeros()
    st = iterative_solve(a2, b, np.zeros(n), 3, JACOBI, 0.5, 1e-3, PRE_NONE, raise_on_error=False)
    assert st == 6


def test_gauss_seidel_reference_arm_is_an_error(gpu):
    from orc_amd.linear_algebra import iterative_solve
    a = fv_like_matrix(4, 3, 2)
    st = iterative_solve(a, np.ones(a.shape[0]), np.zeros(a.shape[0]), 3, 0, 0.5, 1e-3, PRE_NONE, raise_on_error=False)
    assert st == 5


def test_large_spmv_roundtrip_property(gpu):
    """Full-size property (no oracle at this size): A(x + y) == Ax + Ay within rounding, and row sums
    of a Laplacian-like matrix times ones."""
    from orc_amd.linear_algebra import csr_spmv
    a = fv_like_matrix(128, 96, 64)  # 786k rows
    n = a.shape[0]
    ones = np.ones(n)
    y, _ = csr_spmv(a, ones)
    assert np.allclose(y, np.asarray(a.sum(axis=1)).ravel(), rtol=1e-12, atol=1e-12)
    x1, x2 = splitmix64_uniform(n, 1), splitmix64_uniform(n, 2)
    y12, _ = csr_spmv(a, x1 + x2)
    y1, _ = csr_spmv(a, x1)
    y2, _ = csr_spmv(a, x2)
    assert np.allclose(y12, y1 + y2, rtol=1e-11, atol=1e-11)
