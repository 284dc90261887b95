"""The branches of the coarse-level products that only exist at bench size, forced at test size (Missing #4).

At 10.24 M cells 1 % of level 3's 256-row blocks exceed the LDS window (5 000 distinct columns) and take the no-window path of
`spmv_xwin_k` (global gathers from the packed mirror); the "column span exceeds the bitmap" branch of `xwin_build_k` needs a
block whose columns span more than 262 144.  No mesh a CPU oracle finishes in seconds reaches either, so the two limits are
run-time arguments bounded by the compiled LDS sizes (ORC_XWIN_CAP, ORC_XWIN_BITWORDS, read per set-up) and these tests shrink
them until a chosen share of the blocks falls back:
  * the product y = a' x of every coarse level, plain and with the smoothing solves' materialised Jacobi scaling, bit for bit
    against the oracle's CSR product (linear_algebra.rs:82-97: `a_prime * e_prime`; nalgebra-sparse's ascending-column sums),
    with the counters saying how many blocks took which branch;
  * the whole Multigrid arm in the reference's reduction order (linear_algebra.rs:66-141, :270-296), 50 smoother iterations
    per level, bit for bit against the oracle under every combination of the limits."""
import numpy as np
import pytest

from conftest import fv_like_matrix, splitmix64_uniform

pytestmark = pytest.mark.gpu

MULTIGRID = 2


def _levels(shape):
    """fine matrix and its coarse operators (device set-up == oracle, tests/test_gpu_multigrid.py) for as long as rows >= 24 entries"""
    from orc_amd.linear_algebra import amg_coarsen
    a = fv_like_matrix(*shape)
    out = [a]
    for _ in range(3):
        _, ac, _ = amg_coarsen(out[-1])
        out.append(ac)
    return out


def _bits(x):
    return np.asarray(x).view(np.uint64)


@pytest.mark.parametrize("cap,bitwords,expect", [
    (None, None, "all"),        # every block has its window (the test-size default)
    (None, -100, "all"),        # [r04] the windows are built in two passes (small bitmap first, the full one for wider blocks): a first pass of 100 words
                                # (3 200 columns of span) leaves part of the blocks to the second — same windows, same counters, same products
    (None, -1, "all"),          # ... and a first pass that takes nothing
    (2000, None, "some_cap"),   # windows of 2 000 entries: 21 of the 44 blocks fall back (their windows hold 1 000 - 2 600), the others keep theirs
    (300, None, "most_cap"),    # next to none fits
    (None, 200, "some_span"),   # bitmap of 200 words = 6 400 columns of span: 6 of level 2's 30 blocks take the span branch
    (2000, 200, "both"),
])
def test_coarse_products_bit_exact_through_every_window_branch(gpu, oracle, monkeypatch, cap, bitwords, expect):
    from orc_amd.linear_algebra import amg_coarse_product, xwin_counters
    levels = _levels((64, 40, 12))
    if cap is not None:
        monkeypatch.setenv("ORC_XWIN_CAP", str(cap))
    if bitwords is not None and bitwords < 0:
        monkeypatch.setenv("ORC_XWIN_SMALL_BITWORDS", str(-bitwords))
    elif bitwords is not None:
        monkeypatch.setenv("ORC_XWIN_BITWORDS", str(bitwords))
    seen_mirror = 0
    tot = [0, 0, 0]
    for lv in (1, 2):  # products of levels 2 and 3 (built from levels 1 and 2): rows of ~30 and ~60 entries
        fine, coarse = levels[lv], levels[lv + 1]
        nc = coarse.shape[0]
        x = splitmix64_uniform(nc, 40 + lv)
        for scaled in (False, True):
            xwin_counters(reset=True)
            y, mirror = amg_coarse_product(fine, x, scaled=scaled)
            blocks, over_cap, over_span = xwin_counters()
            assert mirror, "level %d has no window mirror (rows too short?)" % (lv + 1)
            seen_mirror += 1
            assert 0 < blocks <= (nc + 255) // 256  # (a block of empty rows — SURVEY Q6: rows without a free neighbour — is not counted)
            for i, v in enumerate((blocks, over_cap, over_span)):
                tot[i] += v
            ref = coarse.copy()
            if scaled:  # p_inv * a (linear_algebra.rs:159-166): one multiplication per entry, 1 / diag per row
                with np.errstate(divide="ignore"):  # an empty coarse row (Q6) has no diagonal: nothing of it is scaled
                    dinv = 1.0 / ref.diagonal()
                ref.data = np.repeat(dinv, np.diff(ref.indptr)) * ref.data
            yo = oracle.Csr.from_scipy(ref).spmv(x)
            assert np.array_equal(_bits(y), _bits(yo)), "level %d scaled=%s: product differs from the oracle" % (lv + 1, scaled)
    assert seen_mirror == 4
    blocks, over_cap, over_span = tot
    if expect == "all":
        assert over_cap == 0 and over_span == 0
    elif expect == "some_cap":
        assert 0 < over_cap < blocks and over_span == 0
    elif expect == "most_cap":
        assert over_cap >= 0.9 * blocks
    elif expect == "some_span":
        assert 0 < over_span < blocks and over_cap == 0
    else:
        assert over_cap > 0 and over_span > 0 and over_cap + over_span < blocks


@pytest.mark.parametrize("cap,bitwords", [(2000, None), (300, None), (None, 200), (2000, 200)])
def test_multigrid_arm_reference_order_bit_exact_with_forced_fallbacks(gpu, oracle, monkeypatch, cap, bitwords):
    """The whole arm (Q4 nested scaling, Q5 r' recursion, Q6 weight-2 rows) at the default 50 iterations: status and every bit of
    x against the oracle while part of the coarse blocks multiply without a window."""
    from orc_amd.linear_algebra import iterative_solve, set_breakdown_guard, set_reduction_order, xwin_counters
    a = fv_like_matrix(64, 40, 12)
    n = a.shape[0]
    b = a @ splitmix64_uniform(n, 7)
    x0 = 0.1 * splitmix64_uniform(n, 8)
    if cap is not None:
        monkeypatch.setenv("ORC_XWIN_CAP", str(cap))
    if bitwords is not None:
        monkeypatch.setenv("ORC_XWIN_BITWORDS", str(bitwords))
    set_reduction_order(1)
    set_breakdown_guard(False)
    try:
        xo = x0.copy()
        sto = oracle.iterative_solve(oracle.Csr.from_scipy(a), b, xo, 50, MULTIGRID, 0.5, 1e-3, 1)
        xwin_counters(reset=True)
        x = x0.copy()
        st = iterative_solve(a, b, x, 50, MULTIGRID, 0.5, 1e-3, 1, raise_on_error=False)
        blocks, over_cap, over_span = xwin_counters()
        assert st == sto
        assert blocks > 0 and over_cap + over_span > 0, (blocks, over_cap, over_span)
        assert np.array_equal(_bits(x), _bits(xo))
    finally:
        set_reduction_order(0)
        set_breakdown_guard(True)


def test_certification_rounds_change_nothing(gpu):
    """the asynchronous cascades of the pairing (tail_chase_k, relaxed agent-scope atomics ordered by s_waitcnt) are
    followed by lock-step rounds from scratch that certify the fixed point.  A certification that takes ONE round has changed
    nothing — i.e. the cascades alone had reached the sequential greedy pairing — on every aggregation of three default-stack
    SIMPLE iterations of a true 3-D channel (4 systems x 3 levels x 3 iterations)."""
    import helpers as H
    from orc_amd.linear_algebra import amg_certification
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    a = set_channel_bcs(hex_channel(48, 32, 20))
    s = NumericalSettings.default(momentum=5, solver_type=MULTIGRID, iterations=6, momentum_relaxation=0.1, pressure_relaxation=0.001)
    amg_certification(reset=True)
    u, v, w, p = H.seeded_fields(a, seed=8)
    solve_steady(Mesh(a), u, v, w, p, s, 1000.0, 1e-3, 3)
    aggs, rounds = amg_certification()
    assert aggs >= 12, aggs
    assert rounds == aggs, "a certification round changed a pairing: %d rounds for %d aggregations" % (rounds, aggs)


def test_level0_row_mirror_does_not_change_a_bit(gpu, monkeypatch):
    """[r04] ORC_AMG_L0_MIRROR (on by default since the end of round 4; 0: off): the set-up walks single rows of the level-0 matrices through a
    row-contiguous mirror (the mesh pattern's CSR form + values exported per solve) instead of the SELL image.  Same entries in the same order:
    three default-stack SIMPLE iterations with and without it, lock-step and per-system momentum solves — identical bits.  (The pattern half is
    built at mesh creation, so the switch is set before the first mesh of this test; run in a process whose earlier meshes lack it the test still
    compares two runs, both without.)"""
    import helpers as H
    from orc_amd.mesh import Mesh, hex_channel, set_channel_bcs
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import solve_steady
    a = set_channel_bcs(hex_channel(40, 24, 16))
    s = NumericalSettings.default(momentum=5, solver_type=MULTIGRID, iterations=8, momentum_relaxation=0.1, pressure_relaxation=0.001)
    out = []
    for mirror, triple in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("ORC_AMG_L0_MIRROR", mirror)
        monkeypatch.setenv("ORC_TRIPLE_MOMENTUM", triple)
        u, v, w, p = H.seeded_fields(a, seed=11)
        solve_steady(Mesh(a), u, v, w, p, s, 1000.0, 1e-3, 3)
        out.append((u, v, w, p))
    assert np.isfinite(out[0][0]).all()
    for other in out[1:]:
        for x, y in zip(out[0], other):
            assert np.array_equal(x, y)
