"""discretization::* of the reference (src/discretization.rs) on MI355X: host arrays in, host arrays out."""
import ctypes as C

import numpy as np

from ._lib import check, lib

_F64 = C.POINTER(C.c_double)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_F64)


def build_momentum_diffusion_matrix(mesh, mu, diffusion_scheme=0):
    """discretization.rs:39-131 -> (a_di values in pattern order, b_u, b_v, b_w)"""
    n = mesh.n_cells
    a = np.empty(mesh.nnz)
    bu, bv, bw = np.empty(n), np.empty(n), np.empty(n)
    check(lib().orc_build_momentum_diffusion_matrix(mesh.ptr, C.c_int(diffusion_scheme), C.c_double(mu), _p(a), _p(bu), _p(bv), _p(bw)))
    return a, bu, bv, bw


def initialize_momentum_matrix(mesh):
    """discretization.rs:450-472"""
    a = np.empty(mesh.nnz)
    check(lib().orc_initialize_momentum_matrix(mesh.ptr, _p(a)))
    return a


def build_momentum_advection_matrices(mesh, a_u, a_v, a_w, a_di, u, v, w, p, settings, rho):
    """discretization.rs:134-356. a_u/a_v/a_w (value arrays) are updated in place, as `&mut CsrMatrix`.
    Returns (b_u, b_v, b_w, (peclet_avg, min, max)); b_* exclude the diffusion RHS, as in the reference."""
    n = mesh.n_cells
    for a in (a_u, a_v, a_w):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    bu, bv, bw = np.empty(n), np.empty(n), np.empty(n)
    pe = np.zeros(3)
    a_di, u, v, w, p = _f64(a_di), _f64(u), _f64(v), _f64(w), _f64(p)
    check(lib().orc_build_momentum_advection_matrices(mesh.ptr, _p(a_u), _p(a_v), _p(a_w), _p(bu), _p(bv), _p(bw), _p(a_di),
                                                      _p(u), _p(v), _p(w), _p(p), C.byref(settings), C.c_double(rho), _p(pe)))
    return bu, bv, bw, tuple(pe)


def build_pressure_correction_matrices(mesh, u, v, w, p, a_u, a_v, a_w, settings, rho):
    """discretization.rs:359-448 -> LinearSystem (a values, b)"""
    a = np.empty(mesh.nnz)
    b = np.empty(mesh.n_cells)
    u, v, w, p, a_u, a_v, a_w = map(_f64, (u, v, w, p, a_u, a_v, a_w))
    check(lib().orc_build_pressure_correction_matrices(mesh.ptr, _p(u), _p(v), _p(w), _p(p), _p(a_u), _p(a_v), _p(a_w),
                                                       C.byref(settings), C.c_double(rho), _p(a), _p(b)))
    return a, b
