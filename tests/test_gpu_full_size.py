"""BASELINE.json's full size (configs[3]: 400 x 160 x 160 = 10.24 M cells), where the CPU oracle is out of reach:
size-independent properties instead of element-wise comparison with the oracle — the product against an independent
CSR product with the same association (bit-exact), linearity, the pairing's and the Galerkin operator's invariants,
conservation in the assembled systems, run-to-run reproducibility of whole SIMPLE iterations."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

NX, NY, NZ = 400, 160, 160


@pytest.fixture(scope="module")
def c4(gpu):
    from bench import initial_fields
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    a = set_channel_bcs(hex_channel(NX, NY, NZ))
    m = Mesh(a)
    s = Solver(m, NumericalSettings.default(momentum=5, momentum_relaxation=0.1, pressure_relaxation=0.001), 1000.0, 1e-3)
    s.set_fields(*initial_fields(np.asarray(a["cell_centroid"])))
    au, av, aw, bu, bv, bw, pe = s.assemble_momentum()
    return a, m, s, m.csr(au), (bu, bv, bw)


def test_counts(c4):
    a, m, *_ = c4
    assert m.n_cells == 10_240_000 and m.nnz == 71_372_800 and len(a["face_area"]) == 30_873_600


def test_spmv_full_size_bit_exact_and_linear(c4):
    """scipy's csr_matvec adds a row's products in ascending column order from 0.0 — the association of
    nalgebra-sparse's product and of spmv_k — so the 10.24 M results must be identical bits."""
    from orc_amd.linear_algebra import csr_spmv
    from orc_amd.mesh import splitmix64_uniform
    a, m, s, A, _ = c4
    n = m.n_cells
    x, y = splitmix64_uniform(n, 1), splitmix64_uniform(n, 2)
    ax, _ = csr_spmv(A, x)
    assert np.array_equal(ax, A @ x)
    ay, _ = csr_spmv(A, y)
    axy, _ = csr_spmv(A, 0.75 * x - 1.5 * y)
    ref = 0.75 * ax - 1.5 * ay
    assert np.abs(axy - ref).max() <= 1e-13 * np.abs(ref).max()


def test_assembled_system_properties(c4):
    """Rows of the momentum matrix: stored diagonal, ascending columns, 4..7 entries; off-diagonals of the diffusive +
    upwinded operator are non-positive up to the TVD correction and the diagonal dominates."""
    a, m, s, A, (bu, bv, bw) = c4
    n = m.n_cells
    lens = np.diff(A.indptr)
    assert lens.min() == 4 and lens.max() == 7
    d = A.diagonal()
    assert (d > 0).all()
    off = A.copy()
    off.setdiag(0.0)
    assert (np.abs(off).sum(axis=1).A1 <= d * (1 + 1e-9) + 1e-300).mean() > 0.99
    assert np.isfinite(bu).all() and np.isfinite(bv).all() and np.isfinite(bw).all()


def test_pairing_and_galerkin_invariants(c4):
    """Greedy pairing (linear_algebra.rs:30-60) and (R a) R^T at full size: a row never pairs with itself, a column is
    taken at most once, partners are matrix neighbours; with R's weights the coarse operator reproduces the fine one
    on prolonged vectors: x_c^T (R a R^T) y_c = (R^T x_c)^T a (R^T y_c)."""
    from orc_amd.linear_algebra import amg_coarsen
    from orc_amd.mesh import splitmix64_uniform
    a, m, s, A, _ = c4
    n = A.shape[0]
    partner, Ac, rounds = amg_coarsen(A)
    has = partner >= 0
    idx = np.nonzero(has)[0]
    assert (partner[idx] != idx).all()
    taken = np.bincount(partner[idx], minlength=n)
    assert taken.max() <= 1
    # partners are structural neighbours of their rows
    pos = np.array([np.searchsorted(A.indices[A.indptr[i]:A.indptr[i + 1]], partner[i]) for i in idx[:: max(1, len(idx) // 20000)]])
    sel = idx[:: max(1, len(idx) // 20000)]
    assert all(A.indices[A.indptr[i] + p] == partner[i] for i, p in zip(sel, pos))
    nc = (n + 1) // 2
    assert Ac.shape == (nc, nc)
    rows = np.repeat(idx // 2, 2)
    cols = np.stack([idx, partner[idx]], axis=1).reshape(-1)
    R = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(nc, n)).tocsr()
    R.sum_duplicates()
    xc, yc = splitmix64_uniform(nc, 3), splitmix64_uniform(nc, 4)
    lhs = xc @ (Ac @ yc)
    rhs = (R.T @ xc) @ (A @ (R.T @ yc))
    assert abs(lhs - rhs) <= 1e-9 * abs(rhs)
    assert rounds >= 1


@pytest.mark.parametrize("solver", [3, 2])
def test_simple_iteration_reproducible_at_full_size(c4, solver):
    """One full SIMPLE iteration twice from the same state: identical bits (fixed-order reductions, exact fixed points;
    the momentum lanes, the two-stream Multigrid arm and the early p' hierarchy are all active here)."""
    from bench import initial_fields
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    a, m, *_ = c4
    st = NumericalSettings.default(momentum=5, solver_type=solver, iterations=10, momentum_relaxation=0.1, pressure_relaxation=0.001)
    out = []
    for _ in range(2):
        s = Solver(m, st, 1000.0, 1e-3)
        s.set_fields(*initial_fields(np.asarray(a["cell_centroid"])))
        status, rep = s.iterate(1, report=True, raise_on_error=False)
        assert status == 0 and np.isfinite(rep).all()
        out.append(s.get_fields())
        del s
    for x, y in zip(*out):
        assert np.array_equal(x, y)
