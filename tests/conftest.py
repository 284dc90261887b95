import gzip
import os
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MESHES = os.path.join(GOLDEN, "meshes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (test infrastructure; never used by orc_amd)."""
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def mesh_path(tmp_path_factory):
    """Path of a reference mesh fixture by stem; the large Couette mesh is stored gzipped."""
    cache = tmp_path_factory.mktemp("meshes")

    def get(stem):
        plain = os.path.join(MESHES, stem + ".msh")
        if os.path.exists(plain):
            return plain
        gz = plain + ".gz"
        out = os.path.join(str(cache), stem + ".msh")
        if not os.path.exists(out):
            with gzip.open(gz, "rb") as fi, open(out, "wb") as fo:
                shutil.copyfileobj(fi, fo)
        return out

    return get


@pytest.fixture
def monkeypatch():
    """pytest's monkeypatch, whose setenv / delenv also tell liborc_amd: the library reads its ORC_* switches ONCE (orc_init) and again only when
    told (orc_reload_environment, orc_amd/csrc/config.hpp) — nothing on its hot path calls getenv.  Undone (and reloaded) at the end of the test."""
    def reload():
        m = sys.modules.get("orc_amd._lib")
        if m is not None and getattr(m, "_lib", None) is not None:
            m._lib.orc_reload_environment()

    class Patch(pytest.MonkeyPatch):
        def setenv(self, name, value, prepend=None):
            super().setenv(name, value, prepend)
            reload()

        def delenv(self, name, raising=True):
            super().delenv(name, raising)
            reload()

    mp = Patch()
    yield mp
    mp.undo()
    reload()


@pytest.fixture(scope="session")
def gpu():
    import orc_amd
    if orc_amd.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box (gpurun)")
    orc_amd.init(0)
    return orc_amd


def unit_test_system(n=100):
    """The system of the reference's only #[test] (linear_algebra.rs:313-337)."""
    import scipy.sparse as sp
    rows, cols, vals = [], [], []
    sol = 2.0 * np.arange(n)
    b = np.zeros(n)
    for i in range(n):
        for j in range(n):
            v = 0.0
            if i == j:
                v = 1.0
            elif j != 0 and j != n - 1 and abs(i - j) == 1:
                v = -0.25
            if v != 0.0:
                rows.append(i)
                cols.append(j)
                vals.append(v)
                b[i] += v * sol[j]
    a = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    a.sort_indices()
    return a, b, sol


def splitmix64_uniform(n, seed=0x4F5243):
    """uniform[-1,1) f64 from splitmix64 (SURVEY §8d synthetic inputs), vectorised."""
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def fv_like_matrix(nx, ny, nz, seed=1):
    """7-point matrix with an ORC-like pattern (diag + one entry per interior face) and
    non-symmetric, diagonally dominant random values."""
    import scipy.sparse as sp
    n = nx * ny * nz
    idx = np.arange(n).reshape(nz, ny, nx)
    rows, cols = [np.arange(n)], [np.arange(n)]
    for ax in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[ax] = slice(0, -1)
        hi[ax] = slice(1, None)
        a, b = idx[tuple(lo)].ravel(), idx[tuple(hi)].ravel()
        rows += [a, b]
        cols += [b, a]
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    r = splitmix64_uniform(len(rows), seed)
    vals = np.where(rows == cols, 0.0, -(0.5 + 0.5 * np.abs(r)))
    a = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    d = -np.asarray(a.sum(axis=1)).ravel() * (1.0 + 0.1 * np.abs(splitmix64_uniform(n, seed + 7))) + 0.01
    a = a + sp.diags(d)
    a = a.tocsr()
    a.sort_indices()
    return a
