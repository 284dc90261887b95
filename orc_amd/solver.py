"""solver::* of the reference (src/solver.rs) on MI355X."""
import ctypes as C

import numpy as np

from ._lib import OrcError, check, lib

_F64 = C.POINTER(C.c_double)
_REPORT_FN = C.CFUNCTYPE(None, C.c_uint64, _F64, _F64, C.c_double, C.c_double, C.c_double, C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_F64)


def solve_steady(mesh, u, v, w, p, numerical_settings, rho, mu, iteration_count, reporting_interval=0, report=None,
                 raise_on_error=True):
    """solver::solve_steady (solver.rs:26-37): u, v, w, p are updated in place.
    report(iteration, mean_velocity[3], peclet[3], velocity_correction, pressure_correction, ms_per_iter)."""
    for a in (u, v, w, p):
        assert a.dtype == np.float64 and a.flags.c_contiguous and len(a) == mesh.n_cells
    cb = None
    if report is not None:
        def _cb(it, mv, pe, vc, pc, ms, _user):
            report(it, (mv[0], mv[1], mv[2]), (pe[0], pe[1], pe[2]), vc, pc, ms)
        cb = _REPORT_FN(_cb)
    st = lib().orc_solve_steady(mesh.ptr, _p(u), _p(v), _p(w), _p(p), C.byref(numerical_settings), C.c_double(rho),
                                C.c_double(mu), C.c_uint64(iteration_count), C.c_uint64(reporting_interval),
                                cb if cb is not None else C.cast(None, _REPORT_FN), None)
    if raise_on_error:
        check(st)
    return st


def calculate_gradients(mesh, u, v, w, p, settings, velocity=True):
    """Green-Gauss arms of calculate_pressure_gradient / calculate_velocity_gradient for every cell."""
    n = mesh.n_cells
    gp = np.empty((n, 3))
    gu = np.empty((n, 3, 3)) if velocity else None
    u, v, w, p = map(_f64, (u, v, w, p))
    check(lib().orc_calculate_gradients(mesh.ptr, _p(u), _p(v), _p(w), _p(p), C.byref(settings), _p(gp),
                                        _p(gu) if velocity else None))
    return gp, gu


class Solver:
    """Device-resident state of one solve_steady call (OrcSolver*): what bench.py drives."""

    def __init__(self, mesh, settings, rho, mu):
        st = C.c_int(0)
        self.mesh = mesh
        self.ptr = lib().orc_solver_create(mesh.ptr, C.byref(settings), C.c_double(rho), C.c_double(mu), C.byref(st))
        check(st.value)
        self.ptr = C.c_void_p(self.ptr)
        self.n = mesh.n_cells

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().orc_solver_destroy(self.ptr)
            self.ptr = None

    def set_fields(self, u, v, w, p):
        u, v, w, p = map(_f64, (u, v, w, p))
        check(lib().orc_solver_set_fields(self.ptr, _p(u), _p(v), _p(w), _p(p)))

    def get_fields(self):
        u, v, w, p = (np.empty(self.n) for _ in range(4))
        check(lib().orc_solver_get_fields(self.ptr, _p(u), _p(v), _p(w), _p(p)))
        return u, v, w, p

    def iterate(self, iterations=1, report=False, raise_on_error=True):
        rep = np.zeros((iterations, 8)) if report else None
        st = lib().orc_solver_iterate(self.ptr, C.c_uint64(iterations), _p(rep) if report else None)
        if raise_on_error:
            check(st)
        return (st, rep) if report else st

    def assemble_momentum(self):
        nnz, n = self.mesh.nnz, self.n
        au, av, aw = np.empty(nnz), np.empty(nnz), np.empty(nnz)
        bu, bv, bw = np.empty(n), np.empty(n), np.empty(n)
        pe = np.zeros(3)
        check(lib().orc_solver_assemble_momentum(self.ptr, _p(au), _p(av), _p(aw), _p(bu), _p(bv), _p(bw), _p(pe)))
        return au, av, aw, bu, bv, bw, tuple(pe)

    def assemble_pressure(self):
        a, b = np.empty(self.mesh.nnz), np.empty(self.n)
        check(lib().orc_solver_assemble_pressure(self.ptr, _p(a), _p(b)))
        return a, b

    def assemble_momentum_only(self):
        """the momentum assembly of the current state, matrices left on the device (bench.py's product measurements)"""
        check(lib().orc_solver_assemble_momentum(self.ptr, None, None, None, None, None, None, None))

    def snapshot(self):
        """device-side copy of the state the next SIMPLE iteration starts from"""
        check(lib().orc_solver_snapshot(self.ptr))

    def restore(self):
        check(lib().orc_solver_restore(self.ptr))

    def bench_amg_levels(self, reps=20):
        """per level of a_u's Multigrid hierarchy: (rows, nnz, padded SELL entries, ms per product)"""
        rows, nnz, padded = (np.zeros(4, dtype=np.int64) for _ in range(3))
        ms = np.zeros(4)
        nl = C.c_int(0)
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        check(lib().orc_bench_amg_levels(self.ptr, C.c_int(reps), i64(rows), i64(nnz), i64(padded), _p(ms), C.byref(nl)))
        return [(int(rows[k]), int(nnz[k]), int(padded[k]), float(ms[k])) for k in range(nl.value)]

    def bench_spmv(self, reps=50):
        ms, cs = C.c_double(0.0), C.c_double(0.0)
        check(lib().orc_bench_spmv(self.ptr, C.c_int(reps), C.byref(ms), C.byref(cs)))
        return ms.value, cs.value

    def bench_inloop_products(self, reps=50):
        """orc_bench_inloop_products: ms per launch of the level-0 products as the BiCGSTAB loop launches them —
        (one system: EpiStoreSum, EpiTs; three systems in one launch: EpiStoreSum3, EpiTs3)."""
        ms = (C.c_double * 4)()
        check(lib().orc_bench_inloop_products(self.ptr, C.c_int(reps), ms))
        lib().orc_bench_inloop_variant.restype = C.c_char_p
        self.inloop_variant = lib().orc_bench_inloop_variant().decode()  # "<narrow>, <scaled>, <non-temporal>" template arguments of those launches
        return [ms[k] for k in range(4)]

    def bench_gs_sweep0(self, reps=50):
        """orc_bench_gs_sweep0: (ms per sweep-from-zero of one system, ms per sweep of u, v, w in one launch per colour, colours)"""
        ms = (C.c_double * 2)()
        nc = C.c_int(0)
        check(lib().orc_bench_gs_sweep0(self.ptr, C.c_int(reps), ms, C.byref(nc)))
        return ms[0], ms[1], nc.value

    def bench_gs_sweep(self, reps=50):
        """orc_bench_gs_sweep: (ms per multicolour Gauss-Seidel sweep over a_u, number of colours = launches per sweep)"""
        ms, nc = C.c_double(0.0), C.c_int(0)
        check(lib().orc_bench_gs_sweep(self.ptr, C.c_int(reps), C.byref(ms), C.byref(nc)))
        return ms.value, nc.value

    def bench_bicgstab_iteration(self, reps=20):
        ms = C.c_double(0.0)
        check(lib().orc_bench_bicgstab_iteration(self.ptr, C.c_int(reps), C.byref(ms)))
        return ms.value


# ------------------------------------------------------------------ solver::initialize_* (solver.rs:246-509)
PRESSURE_ONLY, VELOCITY_ONLY, HYBRID = 0, 1, 2


def check_boundary_conditions(mesh):
    """solver::check_boundary_conditions (solver.rs:710-772) -> PRESSURE_ONLY | VELOCITY_ONLY | HYBRID;
    OrcError(ORC_ERR_NO_BOUNDARY_CONDITIONS) for "You must set boundary conditions."."""
    kind = C.c_int(0)
    check(lib().orc_check_boundary_conditions(mesh.ptr, C.byref(kind)))
    return kind.value


def initialize_pressure_field(mesh, p=None):
    """solver::initialize_pressure_field (solver.rs:414-509); p defaults to zeros like the reference's callers."""
    p = np.zeros(mesh.n_cells) if p is None else _f64(p).copy()
    check(lib().orc_initialize_pressure_field(mesh.ptr, _p(p)))
    return p


def initialize_flow(mesh, mu, rho, iteration_count, settings=None):
    """solver::initialize_flow (solver.rs:246-352) -> (u, v, w, p)."""
    u, v, w, p = (np.zeros(mesh.n_cells) for _ in range(4))
    check(lib().orc_initialize_flow(mesh.ptr, C.c_double(mu), C.c_double(rho), C.c_uint64(iteration_count),
                                    C.byref(settings) if settings is not None else None, _p(u), _p(v), _p(w), _p(p)))
    return u, v, w, p


def initialize_velocity_field(mesh, settings=None):
    """solver::initialize_velocity_field (solver.rs:511-696) -> (u, v, w, psi)."""
    u, v, w, psi = (np.zeros(mesh.n_cells) for _ in range(4))
    check(lib().orc_initialize_velocity_field(mesh.ptr, C.byref(settings) if settings is not None else None, _p(u), _p(v), _p(w), _p(psi)))
    return u, v, w, psi


def initialize_flow_new(mesh, mu, rho, iteration_count):
    """solver::initialize_flow_new (solver.rs:354-410) -> (u, v, w, p)."""
    u, v, w, p = (np.zeros(mesh.n_cells) for _ in range(4))
    check(lib().orc_initialize_flow_new(mesh.ptr, C.c_double(mu), C.c_double(rho), C.c_uint64(iteration_count),
                                        _p(u), _p(v), _p(w), _p(p)))
    return u, v, w, p
