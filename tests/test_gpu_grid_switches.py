"""The measurement switches that size grids, swept to their extremes.

r03: `ORC_XWIN_WGS_PER_CU` = 12 / 20 / 40 launched more workgroups than the partial-sum arrays have entries
(kMaxPartials = 2 048 per quantity), the window product's epilogue wrote past them and the solves froze — "fast" and wrong.
Since r04 every such grid ends in common.hpp's clamp_partials_grid (CPU unit test: tests/test_abi_cpu.py); here the switches
are driven past the bound on the device:
  * in the reference's reduction order no partial sum is read (one wavefront sums every dot product in nalgebra's order), so the
    whole Multigrid arm must stay BIT-identical to the oracle whatever the grids are;
  * in the product's tree order a request beyond the bound must give the bits of the bound itself (the clamp, not a crash and
    not another association), and smaller grids must stay within rounding of the default."""
import numpy as np
import pytest

from conftest import fv_like_matrix, splitmix64_uniform

pytestmark = pytest.mark.gpu

MULTIGRID, BICGSTAB = 2, 3
SWEEP = [("ORC_XWIN_WGS_PER_CU", ["1", "3", "8", "12", "40", "100000"]),
         ("ORC_SPMV_GRID", ["8", "24", "1024", "2048", "4096", "100000000"])]
# (r04 swept ORC_AMG_CHASE_GRID too: the cascades whose grid it sized are gone with r05's pairing by deferred acceptance, whose chain kernel
# writes no partial sums and is launched with kMaxGrid workgroups)


def _system():
    a = fv_like_matrix(64, 40, 12)  # levels 2 and 3 multiply through the window product (rows of ~30 / ~60 entries)
    n = a.shape[0]
    return a, a @ splitmix64_uniform(n, 7), 0.1 * splitmix64_uniform(n, 8)


def test_reference_order_is_bit_exact_under_every_grid_switch(gpu, oracle, monkeypatch):
    from orc_amd.linear_algebra import iterative_solve, set_breakdown_guard, set_reduction_order
    a, b, x0 = _system()
    xo = x0.copy()
    sto = oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 50, MULTIGRID, 0.5, 1e-3, 1)
    set_reduction_order(1)
    set_breakdown_guard(False)
    try:
        for name, values in SWEEP:
            for v in values:
                monkeypatch.setenv(name, v)
                x = x0.copy()
                st = iterative_solve(a, b, x, 50, MULTIGRID, 0.5, 1e-3, 1, raise_on_error=False)
                assert st == sto, (name, v)
                assert np.array_equal(x.view(np.uint64), xo.view(np.uint64)), (name, v)
            monkeypatch.delenv(name)
    finally:
        set_reduction_order(0)
        set_breakdown_guard(True)


def test_tree_order_requests_beyond_the_bound_equal_the_bound(gpu, monkeypatch):
    from orc_amd.linear_algebra import iterative_solve
    a, b, x0 = _system()

    def run(method, iters):
        x = x0.copy()
        assert iterative_solve(a, b, x, iters, method, 0.5, 1e-3, 1, raise_on_error=False) == 0
        assert np.isfinite(x).all()
        return x

    for method, iters in ((MULTIGRID, 6), (BICGSTAB, 6)):  # few iterations: the r_hat_0 = 1 recurrence amplifies last-bit differences
        base = run(method, iters)
        assert np.array_equal(run(method, iters), base)  # run-to-run reproducible
        for name, values in SWEEP:
            at_bound = None
            for v in values:
                monkeypatch.setenv(name, v)
                x = run(method, iters)
                rel = np.linalg.norm(x - base) / np.linalg.norm(base)
                assert rel < 1e-9, (name, v, rel)
                if name != "ORC_XWIN_WGS_PER_CU" and int(v) == 2048:
                    at_bound = x
                if name != "ORC_XWIN_WGS_PER_CU" and int(v) > 2048:
                    assert np.array_equal(x, at_bound), (name, v)
                if name == "ORC_XWIN_WGS_PER_CU" and int(v) >= 8:  # 256 CUs x 8 = the bound on MI355X: 12, 40, ... clamp to it
                    assert np.array_equal(x, base), (name, v)
            monkeypatch.delenv(name)
