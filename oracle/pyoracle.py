"""ctypes binding of liborc_oracle.so — the CPU restatement of ORC's hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under orc_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "liborc_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Settings(C.Structure):
    """include/orc_types.h OrcSettings (same layout for oracle and product)."""
    _fields_ = [
        ("momentum", C.c_int32), ("diffusion", C.c_int32), ("pressure_interpolation", C.c_int32),
        ("velocity_interpolation", C.c_int32), ("gradient_reconstruction", C.c_int32),
        ("solver_type", C.c_int32), ("preconditioner", C.c_int32), ("q1_compat", C.c_int32),
        ("iterations", C.c_uint64), ("momentum_relaxation", C.c_double), ("pressure_relaxation", C.c_double),
        ("relaxation", C.c_double), ("relative_convergence_threshold", C.c_double),
        ("frozen_diagonals", C.c_int32), ("breakdown_guard", C.c_int32),
        ("reduction_order", C.c_int32), ("reserved0", C.c_int32),
    ]


# enums (include/orc_types.h)
UD, CD1, CD2, TVD_LUD, TVD_QUICK, TVD_UMIST, TVD_UD, TVD_CD1 = range(8)
P_LINEAR, P_LINEAR_WEIGHTED, P_STANDARD, P_SECOND_ORDER, P_NONE = range(5)
V_LINEAR, V_LINEAR_WEIGHTED, V_RHIE_CHOW, V_NONE = range(4)
GAUSS_SEIDEL, JACOBI, MULTIGRID, BICGSTAB = range(4)
MULTICOLOR_GS, BICGSTAB_GS_PRECOND, MULTIGRID_GS = 16, 17, 18
PRECOND_NONE, PRECOND_JACOBI = 0, 1
BC_INTERIOR, BC_WALL, BC_PRESSURE_INLET, BC_PRESSURE_OUTLET, BC_SYMMETRY, BC_VELOCITY_INLET = 2, 3, 4, 5, 7, 10


class _Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]


class _Csr(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("ncols", C.c_int64), ("nnz", C.c_int64),
                ("row_ptr", C.POINTER(C.c_int64)), ("col", C.POINTER(C.c_int64)), ("val", C.POINTER(C.c_double))]


class _Zone(C.Structure):
    _fields_ = [("id", C.c_uint64), ("zone_type", C.c_int32), ("scalar_value", C.c_double),
                ("vector_value", _Vec3), ("name", C.c_char * 64)]


class _Mesh(C.Structure):
    _fields_ = [("dimensions", C.c_int32), ("n_vertices", C.c_int64), ("n_faces", C.c_int64),
                ("n_cells", C.c_int64), ("n_zones", C.c_int64), ("vertices", C.POINTER(C.c_double)),
                ("face_zone", C.POINTER(C.c_int32)), ("face_c0", C.POINTER(C.c_int64)),
                ("face_c1", C.POINTER(C.c_int64)), ("face_node_ptr", C.POINTER(C.c_int64)),
                ("face_nodes", C.POINTER(C.c_int64)), ("face_area", C.POINTER(C.c_double)),
                ("face_centroid", C.POINTER(C.c_double)), ("face_normal", C.POINTER(C.c_double)),
                ("cell_face_ptr", C.POINTER(C.c_int64)), ("cell_faces", C.POINTER(C.c_int64)),
                ("cell_volume", C.POINTER(C.c_double)), ("cell_centroid", C.POINTER(C.c_double)),
                ("zones", C.POINTER(_Zone))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.POINTER
        L.or_read_mesh.restype = P(_Mesh)
        L.or_read_mesh.argtypes = [C.c_char_p]
        L.or_mesh_from_arrays.restype = P(_Mesh)
        L.or_mesh_free.argtypes = [P(_Mesh)]
        L.or_mesh_set_zone.argtypes = [P(_Mesh), C.c_char_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]
        L.or_csr_from_arrays.restype = P(_Csr)
        L.or_csr_from_coo.restype = P(_Csr)
        L.or_csr_clone.restype = P(_Csr)
        L.or_spgemm.restype = P(_Csr)
        L.or_transpose.restype = P(_Csr)
        L.or_build_restriction_matrix.restype = P(_Csr)
        L.or_initialize_momentum_matrix.restype = P(_Csr)
        L.or_dot.restype = C.c_double
        L.or_norm.restype = C.c_double
        L.or_sum.restype = C.c_double
        L.or_last_jacobi_sweeps.restype = C.c_int64
        L.or_status_string.restype = C.c_char_p
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_settings(**kw):
    s = Settings()
    lib().or_settings_default(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


class Csr:
    """Owning handle on an OrCsr."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().or_csr_free(self.ptr)
            self.ptr = None

    @classmethod
    def from_scipy(cls, a):
        a = a.tocsr()
        a.sort_indices()
        rp = np.ascontiguousarray(a.indptr, dtype=np.int64)
        ci = np.ascontiguousarray(a.indices, dtype=np.int64)
        v = _f64(a.data)
        return cls(lib().or_csr_from_arrays(C.c_int64(a.shape[0]), C.c_int64(a.shape[1]), _ip(rp), _ip(ci), _dp(v)))

    @classmethod
    def from_coo(cls, nrows, ncols, ri, ci, v):
        ri = np.ascontiguousarray(ri, dtype=np.int64)
        ci = np.ascontiguousarray(ci, dtype=np.int64)
        v = _f64(v)
        return cls(lib().or_csr_from_coo(C.c_int64(nrows), C.c_int64(ncols), C.c_int64(len(v)), _ip(ri), _ip(ci), _dp(v)))

    @property
    def shape(self):
        c = self.ptr.contents
        return (c.nrows, c.ncols)

    @property
    def nnz(self):
        return self.ptr.contents.nnz

    def arrays(self):
        """(row_ptr, col, val) as numpy views into the C storage (val is writable)."""
        c = self.ptr.contents
        rp = np.ctypeslib.as_array(c.row_ptr, shape=(c.nrows + 1,))
        ci = np.ctypeslib.as_array(c.col, shape=(max(c.nnz, 1),))[: c.nnz]
        v = np.ctypeslib.as_array(c.val, shape=(max(c.nnz, 1),))[: c.nnz]
        return rp, ci, v

    def to_scipy(self):
        import scipy.sparse as sp
        rp, ci, v = self.arrays()
        return sp.csr_matrix((v.copy(), ci.copy(), rp.copy()), shape=self.shape)

    def clone(self):
        return Csr(lib().or_csr_clone(self.ptr))

    def diag(self):
        rp, ci, v = self.arrays()
        n = self.shape[0]
        rows = np.repeat(np.arange(n), np.diff(rp))
        d = np.full(n, np.nan)
        m = rows == ci
        d[rows[m]] = v[m]
        return d

    def spmv(self, x):
        x = _f64(x)
        y = np.empty(self.shape[0])
        lib().or_spmv(self.ptr, _dp(x), _dp(y))
        return y

    def matmul(self, other):
        return Csr(lib().or_spgemm(self.ptr, other.ptr))

    def transpose(self):
        return Csr(lib().or_transpose(self.ptr))


def dot(a, b):
    a, b = _f64(a), _f64(b)
    return lib().or_dot(_dp(a), _dp(b), C.c_int64(len(a)))


def norm(a):
    a = _f64(a)
    return lib().or_norm(_dp(a), C.c_int64(len(a)))


def build_restriction_matrix(a, injection=False):
    return Csr(lib().or_build_restriction_matrix(a.ptr, C.c_int(1 if injection else 0)))


def iterative_solve(a, b, x, iteration_count, method, relaxation_factor, convergence_threshold, preconditioner):
    """linear_algebra::iterative_solve; x is updated in place; returns the status code."""
    b = _f64(b)
    assert x.dtype == np.float64 and x.flags.c_contiguous
    return lib().or_iterative_solve(a.ptr, _dp(b), _dp(x), C.c_uint64(iteration_count), C.c_int(method),
                                    C.c_double(relaxation_factor), C.c_double(convergence_threshold),
                                    C.c_int(preconditioner))


def set_dot_mode(mode):
    """diagnostic: 1 = pairwise association of every dot/norm (measures a run's sensitivity to it)"""
    lib().or_set_dot_mode(C.c_int(mode))


def status_string(st):
    return lib().or_status_string(C.c_int(st)).decode()


class Mesh:
    def __init__(self, ptr):
        if not ptr:
            raise RuntimeError("oracle: mesh could not be read")
        self.ptr = ptr

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().or_mesh_free(self.ptr)
            self.ptr = None

    @classmethod
    def read(cls, path):
        return cls(lib().or_read_mesh(path.encode()))

    @classmethod
    def from_arrays(cls, d):
        """d: dict in the layout of orc_amd.mesh.MeshArrays.as_dict()."""
        k = {n: np.ascontiguousarray(d[n]) for n in d if isinstance(d[n], np.ndarray)}
        i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)
        fc0, fc1 = i64(k["face_c0"]), i64(k["face_c1"])
        fz = np.ascontiguousarray(k["face_zone"], dtype=np.int32)
        cfp, cf = i64(k["cell_face_ptr"]), i64(k["cell_faces"])
        zt = np.ascontiguousarray(k["zone_type"], dtype=np.int32)
        zs, zv = _f64(k["zone_scalar"]), _f64(k["zone_vector"])
        fa, fn, fcn = _f64(k["face_area"]), _f64(k["face_normal"]), _f64(k["face_centroid"])
        cc, cv = _f64(k["cell_centroid"]), _f64(k["cell_volume"])
        ptr = lib().or_mesh_from_arrays(
            C.c_int32(3), C.c_int64(len(cv)), C.c_int64(len(fa)), C.c_int32(len(zt)), _ip(fc0), _ip(fc1),
            fz.ctypes.data_as(C.POINTER(C.c_int32)), _dp(fa), _dp(fn), _dp(fcn), _dp(cc), _dp(cv), _ip(cfp), _ip(cf),
            zt.ctypes.data_as(C.POINTER(C.c_int32)), _dp(zs), _dp(zv))
        m = cls(ptr)
        names = d.get("zone_names")
        if names:
            for i, nm in enumerate(names):
                m.ptr.contents.zones[i].name = nm.encode()
        return m

    # --- sizes
    @property
    def n_cells(self):
        return self.ptr.contents.n_cells

    @property
    def n_faces(self):
        return self.ptr.contents.n_faces

    @property
    def n_vertices(self):
        return self.ptr.contents.n_vertices

    @property
    def dimensions(self):
        return self.ptr.contents.dimensions

    def zone_names(self):
        c = self.ptr.contents
        return [c.zones[i].name.decode() for i in range(c.n_zones)]

    def set_zone(self, name, zone_type, scalar=0.0, vector=(0.0, 0.0, 0.0)):
        st = lib().or_mesh_set_zone(self.ptr, name.encode(), C.c_int32(zone_type), C.c_double(scalar),
                                    C.c_double(vector[0]), C.c_double(vector[1]), C.c_double(vector[2]))
        if st:
            raise KeyError("face zone '%s' should exist in mesh" % name)

    def arrays(self):
        """Copy of the mesh as plain numpy arrays (the layout orc_mesh_create takes)."""
        c = self.ptr.contents
        F, n, Z = c.n_faces, c.n_cells, c.n_zones
        A = np.ctypeslib.as_array
        cfp = A(c.cell_face_ptr, shape=(n + 1,)).copy()
        out = dict(
            face_c0=A(c.face_c0, shape=(F,)).copy(), face_c1=A(c.face_c1, shape=(F,)).copy(),
            face_zone=A(c.face_zone, shape=(F,)).copy(), face_area=A(c.face_area, shape=(F,)).copy(),
            face_normal=A(c.face_normal, shape=(F, 3)).copy(), face_centroid=A(c.face_centroid, shape=(F, 3)).copy(),
            cell_centroid=A(c.cell_centroid, shape=(n, 3)).copy(), cell_volume=A(c.cell_volume, shape=(n,)).copy(),
            cell_face_ptr=cfp, cell_faces=A(c.cell_faces, shape=(int(cfp[-1]),)).copy(),
            zone_type=np.array([c.zones[i].zone_type for i in range(Z)], dtype=np.int32),
            zone_scalar=np.array([c.zones[i].scalar_value for i in range(Z)]),
            zone_vector=np.array([[c.zones[i].vector_value.x, c.zones[i].vector_value.y, c.zones[i].vector_value.z]
                                  for i in range(Z)]).reshape(Z, 3),
            zone_names=self.zone_names(),
        )
        return out


def _check(st):
    if st:
        raise RuntimeError("oracle: %s (status %d)" % (status_string(st), st))


def build_momentum_diffusion_matrix(mesh, mu, diffusion_scheme=0):
    n = mesh.n_cells
    bu, bv, bw = np.zeros(n), np.zeros(n), np.zeros(n)
    out = C.POINTER(_Csr)()
    _check(lib().or_build_momentum_diffusion_matrix(mesh.ptr, C.c_int(diffusion_scheme), C.c_double(mu), C.byref(out),
                                                    _dp(bu), _dp(bv), _dp(bw)))
    return Csr(out), bu, bv, bw


def initialize_momentum_matrix(mesh):
    return Csr(lib().or_initialize_momentum_matrix(mesh.ptr))


def build_momentum_advection_matrices(a_u, a_v, a_w, a_di, mesh, u, v, w, p, settings, rho):
    """Overwrites a_u/a_v/a_w values in place; returns (b_u, b_v, b_w, (pe_avg, pe_min, pe_max))."""
    n = mesh.n_cells
    bu, bv, bw = np.zeros(n), np.zeros(n), np.zeros(n)
    pe = np.zeros(3)
    u, v, w, p = _f64(u), _f64(v), _f64(w), _f64(p)
    _check(lib().or_build_momentum_advection_matrices(a_u.ptr, a_v.ptr, a_w.ptr, _dp(bu), _dp(bv), _dp(bw), a_di.ptr,
                                                      mesh.ptr, _dp(u), _dp(v), _dp(w), _dp(p), C.byref(settings),
                                                      C.c_double(rho), _dp(pe)))
    return bu, bv, bw, tuple(pe)


def build_pressure_correction_matrices(mesh, u, v, w, p, a_u, a_v, a_w, settings, rho):
    n = mesh.n_cells
    b = np.zeros(n)
    out = C.POINTER(_Csr)()
    u, v, w, p = _f64(u), _f64(v), _f64(w), _f64(p)
    _check(lib().or_build_pressure_correction_matrices(mesh.ptr, _dp(u), _dp(v), _dp(w), _dp(p), a_u.ptr, a_v.ptr,
                                                       a_w.ptr, C.byref(settings), C.c_double(rho), C.byref(out), _dp(b)))
    return Csr(out), b


def apply_pressure_correction(mesh, du, dv, dw, p_prime, u, v, w, p, settings):
    norms = np.zeros(2)
    du, dv, dw, p_prime = _f64(du), _f64(dv), _f64(dw), _f64(p_prime)
    _check(lib().or_apply_pressure_correction(mesh.ptr, _dp(du), _dp(dv), _dp(dw), _dp(p_prime), _dp(u), _dp(v), _dp(w),
                                              _dp(p), C.byref(settings), _dp(norms)))
    return tuple(norms)


def pressure_gradient(mesh, p, q1=1):
    p = _f64(p)
    n = mesh.n_cells
    g = np.zeros((n, 3))
    out = _Vec3()
    for c in range(n):
        _check(lib().or_calculate_pressure_gradient(mesh.ptr, _dp(p), C.c_int64(c), C.c_int(0), C.c_int(q1), C.byref(out)))
        g[c] = (out.x, out.y, out.z)
    return g


def velocity_gradient(mesh, u, v, w):
    """calculate_velocity_gradient for every cell -> [n, 3, 3] (rows = Tensor.x, .y, .z)."""
    u, v, w = _f64(u), _f64(v), _f64(w)
    n = mesh.n_cells
    g = np.zeros((n, 3, 3))
    out = (_Vec3 * 3)()
    for c in range(n):
        _check(lib().or_calculate_velocity_gradient(mesh.ptr, _dp(u), _dp(v), _dp(w), C.c_int64(c), C.c_int(0), out))
        g[c] = [(r.x, r.y, r.z) for r in out]
    return g


def solve_steady(mesh, u, v, w, p, settings, rho, mu, iteration_count, report=False):
    """solver::solve_steady; fields updated in place. Returns (status, report[iters,6] or None)."""
    for a in (u, v, w, p):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    rep = np.zeros((iteration_count, 6)) if report else None
    st = lib().or_solve_steady(mesh.ptr, _dp(u), _dp(v), _dp(w), _dp(p), C.byref(settings), C.c_double(rho),
                               C.c_double(mu), C.c_uint64(iteration_count), _dp(rep) if report else None)
    return st, rep


def initialize_flow(mesh, mu, rho, iteration_count, q1_compat=1):
    n = mesh.n_cells
    u, v, w, p = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    st = lib().or_initialize_flow(mesh.ptr, C.c_double(mu), C.c_double(rho), C.c_uint64(iteration_count),
                                  C.c_int(q1_compat), _dp(u), _dp(v), _dp(w), _dp(p))
    return st, u, v, w, p


def initialize_velocity_field(mesh):
    """solver::initialize_velocity_field (solver.rs:511-696) -> (status, u, v, w, psi)"""
    n = mesh.n_cells
    u, v, w, psi = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    st = lib().or_initialize_velocity_field(mesh.ptr, _dp(u), _dp(v), _dp(w), _dp(psi))
    return st, u, v, w, psi


def initialize_pressure_field(mesh):
    p = np.zeros(mesh.n_cells)
    st = lib().or_initialize_pressure_field(mesh.ptr, _dp(p))
    return st, p
