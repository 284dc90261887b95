"""BASELINE.json configs[2] at its full size: channel_flow.msh's layout (one cell deep, symmetry planes in z) refined to
512 x 2016 x 1 = 1.03 M hexahedra, QUICK + Rhie-Chow, multicolour-Gauss-Seidel-preconditioned BiCGSTAB.  The CPU oracle
needs minutes per SIMPLE iteration here, so the checks are the size-independent ones: the colouring is proper, a coloured
sweep equals a colour-by-colour numpy sweep with the same row association (bit-exact), the solver reduces the residual
of the assembled momentum system, and whole SIMPLE iterations are reproducible bit for bit.  (Element-wise parity of
this configuration with the oracle is tests/test_gpu_gauss_seidel.py on channel_flow.msh itself.)"""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

NX, NY, NZ = 512, 2016, 1
TVD_QUICK = 4
MULTICOLOR_GS, BICGSTAB_GS = 16, 17


def _fields(cc):
    from conftest import splitmix64_uniform
    import helpers as H
    n = len(cc)
    u = H.analytical_poiseuille(cc[:, 1]) * (1 + 0.02 * splitmix64_uniform(n, 1))
    return u, 1e-7 * splitmix64_uniform(n, 2), 1e-12 * splitmix64_uniform(n, 3), -0.01 * (1 - cc[:, 0] / 0.002)


@pytest.fixture(scope="module")
def c3(gpu):
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    a = set_channel_bcs(hex_channel(NX, NY, NZ))
    m = Mesh(a)
    s = Solver(m, NumericalSettings.default(momentum=TVD_QUICK, solver_type=BICGSTAB_GS, iterations=30), 1000.0, 1e-3)
    s.set_fields(*_fields(np.asarray(a["cell_centroid"])))
    au, av, aw, bu, bv, bw, pe = s.assemble_momentum()
    A = m.csr(au)
    A.sort_indices()
    return a, m, A, bu


def test_counts(c3):
    a, m, A, _ = c3
    assert m.n_cells == NX * NY == 1_032_192
    # one cell deep: the z faces are symmetry planes, a row couples to its 4 in-plane neighbours at most
    nnz_row = np.diff(A.indptr)
    assert nnz_row.min() == 3 and nnz_row.max() == 5 and A.nnz == 5 * NX * NY - 2 * NX - 2 * NY


def test_coloring_is_proper_at_full_size(c3):
    from orc_amd.linear_algebra import debug_coloring
    _, _, A, _ = c3
    colors, nc = debug_coloring(A)
    coo = A.tocoo()
    off = coo.row != coo.col
    assert np.all(colors[coo.row[off]] != colors[coo.col[off]])
    assert colors.min() == 0 and nc == colors.max() + 1 and nc <= 8
    assert np.bincount(colors).min() > 0


@pytest.mark.parametrize("precond", [0, 1])
def test_colored_sweeps_bit_exact_at_full_size(c3, precond):
    """Rows of one colour do not couple, so a sweep is one masked update per colour; scipy's csr_matvec adds a row's
    products in ascending column order from 0.0, which is the association of the sweep kernel (the diagonal contributes a
    literal zero, linear_algebra.rs:225-239)."""
    from conftest import splitmix64_uniform
    from orc_amd.linear_algebra import debug_coloring, iterative_solve
    _, _, A, b = c3
    n = A.shape[0]
    x0 = 1e-3 * splitmix64_uniform(n, 6)
    colors, nc = debug_coloring(A)
    x = x0.copy()
    iterative_solve(A, b, x, 3, MULTICOLOR_GS, 0.8, 1e-3, precond)
    d = A.diagonal()
    As = sp.diags(1.0 / d) @ A if precond else A.copy()  # the Jacobi preconditioner scales row i by 1 / a_ii
    As = As.tocsr()
    As.sort_indices()
    bs = (0.0 + (1.0 / d) * b) if precond else b
    ds = As.diagonal()
    A0 = As.copy()
    A0.setdiag(0.0)  # explicit zeros stay in the pattern
    ref = x0.copy()
    rows_of = [np.flatnonzero(colors == c) for c in range(nc)]
    sub = [A0[r] for r in rows_of]
    for _ in range(3):
        for r, Ar in zip(rows_of, sub):
            s = Ar @ ref
            ref[r] = ref[r] * (1.0 - 0.8) + 0.8 * (bs[r] - s) / ds[r]
    assert np.array_equal(x, ref)


def _colored_sweep_from_zero(A0, d, rows_of, rhs):
    """M^-1 rhs: one Gauss-Seidel sweep (omega 1) in colour order from a zero vector."""
    y = np.zeros_like(rhs)
    for r, Ar in zip(rows_of, A0):
        y[r] = y[r] * (1.0 - 1.0) + 1.0 * (rhs[r] - Ar @ y) / d[r]
    return y


def test_gs_preconditioned_bicgstab_matches_numpy_recurrences(c3):
    """linear_algebra.rs:247-269 (rho = sum(r): the shadow residual is the all-ones vector; no breakdown test) with
    p^ = M^-1 p, s^ = M^-1 s and M^-1 = one coloured sweep from zero, in numpy at the full size.  The sweeps and products
    are bit-exact (test above); only the sums differ in association, so 6 iterations agree to 1e-9 rel-L2."""
    import helpers as H
    from orc_amd.linear_algebra import debug_coloring, iterative_solve
    _, _, A, b = c3
    n = A.shape[0]
    x = np.zeros(n)
    iterative_solve(A, b, x, 6, BICGSTAB_GS, 1.0, 1e-30, 0)
    colors, nc = debug_coloring(A)
    d = A.diagonal()
    A0 = A.copy()
    A0.setdiag(0.0)
    rows_of = [np.flatnonzero(colors == c) for c in range(nc)]
    sub = [A0[r] for r in rows_of]
    xr = np.zeros(n)
    r = b - A @ xr
    pv = r.copy()
    rho = r.sum()
    for _ in range(6):
        ph = _colored_sweep_from_zero(sub, d, rows_of, pv)
        nu = A @ ph
        alpha = rho / nu.sum()
        sv = r - alpha * nu
        sh = _colored_sweep_from_zero(sub, d, rows_of, sv)
        t = A @ sh
        omega = t.dot(sv) / t.dot(t)
        xr = (xr + alpha * ph) + omega * sh
        r = sv - omega * t
        rho_new = r.sum()
        pv = r + (rho_new / rho * alpha / omega) * (pv - omega * nu)
        rho = rho_new
    assert np.isfinite(x).all() and H.rel_l2(x, xr) < 1e-9


def test_simple_iterations_reproducible_at_full_size(c3):
    """Three SIMPLE iterations of the configuration twice from the same state: status 0, finite, identical bits
    (fields and the residual report)."""
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    a, m, *_ = c3
    st = NumericalSettings.default(momentum=TVD_QUICK, solver_type=BICGSTAB_GS, iterations=30)
    out = []
    for _ in range(2):
        s = Solver(m, st, 1000.0, 1e-3)
        s.set_fields(*_fields(np.asarray(a["cell_centroid"])))
        status, rep = s.iterate(3, report=True, raise_on_error=False)
        assert status == 0 and np.isfinite(rep).all()
        out.append(s.get_fields() + (np.asarray(rep),))
        del s
    for x, y in zip(*out):
        assert np.isfinite(x).all() and np.array_equal(x, y)
