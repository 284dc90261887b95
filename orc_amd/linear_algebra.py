"""linear_algebra::iterative_solve of the reference (src/linear_algebra.rs:144-299) on MI355X."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def iterative_solve(a, b, solution_vector, iteration_count, method, relaxation_factor, convergence_threshold,
                    preconditioner, raise_on_error=True):
    """a: scipy.sparse CSR (sorted columns); solution_vector (float64, contiguous) is updated in place,
    like `&mut DVector` in the reference. Returns the status code (0 = ok)."""
    a = a.tocsr()
    a.sort_indices()
    rp, ci, v = _i64(a.indptr), _i64(a.indices), _f64(a.data)
    b = _f64(b)
    assert solution_vector.dtype == np.float64 and solution_vector.flags.c_contiguous
    st = lib().orc_iterative_solve(C.c_int64(a.shape[0]), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(v, C.c_double),
                                   _p(b, C.c_double), _p(solution_vector, C.c_double), C.c_uint64(iteration_count),
                                   C.c_int(method), C.c_double(relaxation_factor), C.c_double(convergence_threshold),
                                   C.c_int(preconditioner))
    if raise_on_error:
        check(st)
    return st


def iterative_solve3(a_list, b_list, x_list, iteration_count, method, relaxation_factor, convergence_threshold, preconditioner):
    """orc_iterative_solve3: three systems on ONE pattern (scipy CSR matrices with identical indptr / indices) in lock-step;
    x_list is updated in place.  Returns (status, [status per system])."""
    a0 = a_list[0]
    n = a0.shape[0]
    rp, ci = _i64(a0.indptr), _i64(a0.indices)
    for a in a_list[1:]:
        assert np.array_equal(a.indptr, a0.indptr) and np.array_equal(a.indices, a0.indices), "the three systems must share their pattern"
    vals = [_f64(a.data) for a in a_list]
    bs = [_f64(b) for b in b_list]
    for x in x_list:
        assert x.dtype == np.float64 and x.flags.c_contiguous and len(x) == n
    PD = C.POINTER(C.c_double)
    v3 = (PD * 3)(*[_p(v, C.c_double) for v in vals])
    b3 = (PD * 3)(*[_p(b, C.c_double) for b in bs])
    x3 = (PD * 3)(*[_p(x, C.c_double) for x in x_list])
    st3 = (C.c_int * 3)()
    st = lib().orc_iterative_solve3(C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), v3, b3, x3, C.c_uint64(iteration_count), C.c_int(method),
                                    C.c_double(relaxation_factor), C.c_double(convergence_threshold), C.c_int(preconditioner), st3)
    return st, [st3[0], st3[1], st3[2]]


def set_breakdown_guard(on):
    """process-wide OrcSettings.breakdown_guard for iterative_solve (default on); off = NaN like the reference"""
    check(lib().orc_set_breakdown_guard(C.c_int(1 if on else 0)))


def set_reduction_order(order):
    """process-wide OrcSettings.reduction_order for iterative_solve: 0 = wave trees (default), 1 = the reference's
    (nalgebra dotx) association, bit-identical iterates (verification mode)"""
    check(lib().orc_set_reduction_order(C.c_int(int(order))))


def breakdown_guard_events(reset=False):
    """BiCGSTAB solves in which the breakdown guard fired since the last reset (the reference would have returned NaN)"""
    lib().orc_breakdown_guard_events.restype = C.c_int64
    return int(lib().orc_breakdown_guard_events(C.c_int(1 if reset else 0)))


def last_jacobi_sweeps():
    return lib().orc_last_jacobi_sweeps()


def csr_spmv(a, x, reps=1):
    """y = A x on the device; returns (y, avg_ms_per_launch)."""
    a = a.tocsr()
    a.sort_indices()
    rp, ci, v = _i64(a.indptr), _i64(a.indices), _f64(a.data)
    x = _f64(x)
    y = np.empty(a.shape[0])
    ms = C.c_double(0.0)
    check(lib().orc_csr_spmv(C.c_int64(a.shape[0]), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(v, C.c_double),
                             _p(x, C.c_double), _p(y, C.c_double), C.c_int(reps), C.byref(ms)))
    return y, ms.value


def amg_coarsen(a):
    """Test hook: (partner[n], coarse CSR a' = (R a) R^T, rounds) of the device Multigrid set-up."""
    import scipy.sparse as sp
    a = a.tocsr()
    a.sort_indices()
    n = a.shape[0]
    rp, ci, v = _i64(a.indptr), _i64(a.indices), _f64(a.data)
    partner = np.empty(n, np.int64)
    nc, nnz, rounds = C.c_int64(0), C.c_int64(0), C.c_int(0)
    args = (C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(v, C.c_double), _p(partner, C.c_int64), C.byref(nc), C.byref(nnz))
    check(lib().orc_amg_coarsen(*args, None, None, None, C.byref(rounds)))
    orp, oci, ov = np.empty(nc.value + 1, np.int64), np.empty(nnz.value, np.int64), np.empty(nnz.value)
    check(lib().orc_amg_coarsen(*args, _p(orp, C.c_int64), _p(oci, C.c_int64), _p(ov, C.c_double), C.byref(rounds)))
    return partner, sp.csr_matrix((ov, oci, orp), shape=(nc.value, nc.value)), rounds.value


def debug_coloring(a):
    """Test hook: (colors[n], n_colors) of the multicolour Gauss-Seidel extension for the pattern of `a`."""
    a = a.tocsr()
    a.sort_indices()
    n = a.shape[0]
    rp, ci = _i64(a.indptr), _i64(a.indices)
    colors = np.empty(n, np.int32)
    nc = C.c_int32(0)
    check(lib().orc_debug_coloring(C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(colors, C.c_int32), C.byref(nc)))
    return colors, nc.value


def amg_coarse_product(a, x, scaled=False):
    """Test hook: y = a' x with a' = (R a) R^T launched as the Multigrid solves launch it (packed mirror + LDS x windows where the
    set-up builds them); x has ceil(n / 2) entries.  Returns (y, has_window_mirror)."""
    a = a.tocsr()
    a.sort_indices()
    n = a.shape[0]
    rp, ci, v = _i64(a.indptr), _i64(a.indices), _f64(a.data)
    x = _f64(x)
    assert len(x) == (n + 1) // 2
    y = np.empty(len(x))
    mirror = C.c_int(0)
    check(lib().orc_debug_amg_coarse_product(C.c_int64(n), _p(rp, C.c_int64), _p(ci, C.c_int64), _p(v, C.c_double), C.c_int(1 if scaled else 0),
                                             _p(x, C.c_double), _p(y, C.c_double), C.byref(mirror)))
    return y, bool(mirror.value)


def xwin_counters(reset=False):
    """Test hook: (blocks described, blocks without a window because of the cap, ... because of the column span) since the last reset"""
    out = (C.c_longlong * 3)()
    check(lib().orc_debug_xwin_counters(out, C.c_int(1 if reset else 0)))
    return out[0], out[1], out[2]


def shared_galerkin(reset=False):
    """Test hook: sibling coarse operators built by a shared Galerkin pass since the last reset (two per lock-step momentum solve)"""
    f = lib().orc_debug_shared_galerkin
    f.restype = C.c_longlong
    return int(f(C.c_int(1 if reset else 0)))


def amg_certification(reset=False):
    """Test hook: (aggregations certified after their cascades, certification rounds in total); equal = nothing was changed"""
    out = (C.c_longlong * 2)()
    check(lib().orc_debug_amg_certification(out, C.c_int(1 if reset else 0)))
    return out[0], out[1]
