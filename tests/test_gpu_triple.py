"""Three systems on one sparsity pattern in lock-step (orc_iterative_solve3; what solver::solve_steady's momentum solves are,
solver.rs:99-136): one column stream and one 24-byte gather per entry serve three value streams, vectors are interleaved.
Per system every kernel keeps the thread -> element map, the order of additions and the fold of its one-system
counterpart, so each system's result must equal orc_iterative_solve on that system alone IN EVERY BIT — BiCGSTAB arm
(linear_algebra.rs:247-269) and Multigrid arm (:270-296), with and without the Jacobi preconditioner (:159-167), with the
breakdown guard firing on one system while the others carry on."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import fv_like_matrix, splitmix64_uniform

pytestmark = pytest.mark.gpu

MULTIGRID, BICGSTAB = 2, 3
PRE_NONE, PRE_JACOBI = 0, 1


def same_bits(a, b):
    a, b = np.asarray(a), np.asarray(b)
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a[~na].view(np.uint64), b[~nb].view(np.uint64))


def three_systems(shape, distinct_pairings=False, seed=1):
    """Three non-symmetric FV-like matrices on one pattern.  By default the off-diagonals of systems 1 and 2 are small
    perturbations of system 0's (as the TVD limiter perturbs a_v, a_w against a_u): the greedy pairings coincide.
    distinct_pairings: independent random values, so the pairings differ."""
    a0 = fv_like_matrix(*shape, seed=seed)
    n = a0.shape[0]
    mats = [a0]
    for k in (1, 2):
        if distinct_pairings == "scaled":
            # every row times a positive factor of its own: other values, the same order inside every row, hence the SAME greedy pairing
            # (linear_algebra.rs:38-52 compares the entries of one row) — the case the shared Galerkin pass is for
            a = a0.copy()
            scale = 1.0 + 0.5 * np.abs(splitmix64_uniform(n, seed + 1000 * k))
            a.data = a0.data * np.repeat(scale, np.diff(a0.indptr))
        elif distinct_pairings:
            a = fv_like_matrix(*shape, seed=seed + 10 * k)
            assert np.array_equal(a.indptr, a0.indptr) and np.array_equal(a.indices, a0.indices)
        else:
            a = a0.copy()
            a.data = a0.data * (1.0 + 1e-3 * splitmix64_uniform(len(a0.data), seed + 100 * k))
        mats.append(a)
    bs = [m @ splitmix64_uniform(n, 7 + k) for k, m in enumerate(mats)]
    xs = [0.1 * splitmix64_uniform(n, 20 + k) for k in range(3)]
    return mats, bs, xs


@pytest.mark.parametrize("shape", [(7, 5, 3), (33, 9, 4), (64, 40, 12), (13, 11, 5)])
@pytest.mark.parametrize("precond", [PRE_NONE, PRE_JACOBI])
@pytest.mark.parametrize("iters", [1, 7, 50])
def test_bicgstab_three_systems_bit_identical_to_one_at_a_time(gpu, shape, precond, iters):
    from orc_amd._lib import last_error
    from orc_amd.linear_algebra import iterative_solve, iterative_solve3
    mats, bs, xs = three_systems(shape)
    x3 = [x.copy() for x in xs]
    st, st3 = iterative_solve3(mats, bs, x3, iters, BICGSTAB, 0.5, 1e-3, precond)
    assert st == 0 and st3 == [0, 0, 0], (st, st3, last_error())
    for k in range(3):
        x1 = xs[k].copy()
        assert iterative_solve(mats[k], bs[k], x1, iters, BICGSTAB, 0.5, 1e-3, precond, raise_on_error=False) == 0
        assert same_bits(x3[k], x1), "system %d" % k


@pytest.mark.parametrize("shape", [(20, 17, 9), (64, 40, 12), (33, 9, 4)])
@pytest.mark.parametrize("distinct", [False, True, "scaled"])
@pytest.mark.parametrize("precond", [PRE_JACOBI, PRE_NONE])
def test_multigrid_arm_three_systems_bit_identical(gpu, shape, distinct, precond):
    """Shared pairings ("scaled": level 1 in lock-step as well, its three operators from ONE Galerkin pass), pairings that differ in a few rows
    (False: the perturbed systems) and distinct pairings (per-system coarse parts): either way the bits of three separate Multigrid solves."""
    from orc_amd.linear_algebra import iterative_solve, iterative_solve3, shared_galerkin
    mats, bs, xs = three_systems(shape, distinct_pairings=distinct)
    for iters in (6, 50):
        x3 = [x.copy() for x in xs]
        shared_galerkin(reset=True)
        st, st3 = iterative_solve3(mats, bs, x3, iters, MULTIGRID, 0.5, 1e-3, precond)
        assert st == 0
        # [r04] shared pairings: ONE Galerkin pass built the first coarse operators of all three systems (two sibling operators)
        if distinct == "scaled":
            assert shared_galerkin() == 2
        elif distinct:
            assert shared_galerkin() == 0
        for k in range(3):
            x1 = xs[k].copy()
            st1 = iterative_solve(mats[k], bs[k], x1, iters, MULTIGRID, 0.5, 1e-3, precond, raise_on_error=False)
            assert st1 == st3[k], "system %d" % k
            if st1 == 0:
                assert same_bits(x3[k], x1), "system %d, %d iterations" % (k, iters)


@pytest.mark.parametrize("shape", [(20, 17, 9), (64, 40, 12)])
@pytest.mark.parametrize("odd_one", [1, 2])
def test_shared_galerkin_pass_with_one_sibling_and_switched_off(gpu, monkeypatch, shape, odd_one):
    """[r04] The shared Galerkin pass (amg.hip: MergeSiblings; linear_algebra.rs:80-84 per system) with TWO value sets: one of v / w has a
    pairing of its own and multiplies for itself, the other rides with u.  And ORC_AMG_SHARED_GALERKIN=0 (every system for itself, r03)
    gives the same bits."""
    from orc_amd.linear_algebra import iterative_solve, iterative_solve3, shared_galerkin
    near, bs, xs = three_systems(shape, distinct_pairings="scaled")
    far, bs_far, _ = three_systems(shape, distinct_pairings=True)
    mats = list(near)
    mats[odd_one] = far[odd_one]
    bs = list(bs)
    bs[odd_one] = bs_far[odd_one]
    results = {}
    for switch in ("1", "0"):
        monkeypatch.setenv("ORC_AMG_SHARED_GALERKIN", switch)
        x3 = [x.copy() for x in xs]
        shared_galerkin(reset=True)
        st, st3 = iterative_solve3(mats, bs, x3, 10, MULTIGRID, 0.5, 1e-3, PRE_JACOBI)
        assert st == 0
        assert shared_galerkin() == (1 if switch == "1" else 0)
        results[switch] = (st3, x3)
    monkeypatch.delenv("ORC_AMG_SHARED_GALERKIN")
    assert results["1"][0] == results["0"][0]
    for k in range(3):
        x1 = xs[k].copy()
        st1 = iterative_solve(mats[k], bs[k], x1, 10, MULTIGRID, 0.5, 1e-3, PRE_JACOBI, raise_on_error=False)
        assert st1 == results["1"][0][k]
        if st1 == 0:
            assert same_bits(results["1"][1][k], x1), "system %d (shared pass)" % k
            assert same_bits(results["0"][1][k], x1), "system %d (every system for itself)" % k


def test_breakdown_guard_acts_per_system(gpu):
    """System 1 starts from its exact solution with a zero right-hand side (rho = 0: frozen at once), system 2 is the identity
    (s = r - alpha nu vanishes in the first half step: x = h, then frozen) — the reference would return NaN for both
    (linear_algebra.rs:255-268 has no test) — while system 0 iterates on: each as in its own one-system solve."""
    from orc_amd.linear_algebra import breakdown_guard_events, iterative_solve, iterative_solve3
    shape = (12, 9, 5)
    mats, bs, xs = three_systems(shape)
    n = mats[0].shape[0]
    bs[1] = np.zeros(n)
    xs[1] = np.zeros(n)
    ident = mats[0].copy()
    ident.data = np.where(np.repeat(np.arange(n), np.diff(ident.indptr)) == ident.indices, 1.0, 0.0)
    mats[2] = ident
    xs[2] = np.zeros(n)
    ev0 = breakdown_guard_events()
    x3 = [x.copy() for x in xs]
    st, st3 = iterative_solve3(mats, bs, x3, 12, BICGSTAB, 0.5, 1e-3, PRE_NONE)
    assert st == 0 and st3 == [0, 0, 0]
    assert breakdown_guard_events() - ev0 == 2
    assert np.array_equal(x3[1], np.zeros(n))
    assert np.array_equal(x3[2], bs[2])  # x = h = x + alpha p = b exactly, and it stays there
    for k in range(3):
        x1 = xs[k].copy()
        assert iterative_solve(mats[k], bs[k], x1, 12, BICGSTAB, 0.5, 1e-3, PRE_NONE, raise_on_error=False) == 0
        assert same_bits(x3[k], x1), "system %d" % k
    assert np.isfinite(x3[0]).all() and np.linalg.norm(mats[0] @ x3[0] - bs[0]) < 1e-2 * np.linalg.norm(bs[0])  # system 0 carried on


def test_half_step_convergence_stays_put(gpu):
    """One-system regression for the guard's x = h branch (bicg_xr_k): the solve that converges exactly in a half step must
    not add alpha p again in the iterations that follow."""
    from orc_amd.linear_algebra import iterative_solve
    n = 1000
    a = sp.identity(n, format="csr")
    b = splitmix64_uniform(n, 3)
    x = np.zeros(n)
    assert iterative_solve(a, b, x, 9, BICGSTAB, 0.5, 1e-3, PRE_NONE, raise_on_error=False) == 0
    assert np.array_equal(x, b)
