"""mesh::Mesh of the reference (src/mesh.rs) as flat arrays + the device-resident OrcMesh handle."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .settings import FaceConditionTypes

_F64 = C.POINTER(C.c_double)
_I64 = C.POINTER(C.c_int64)
_I32 = C.POINTER(C.c_int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class MeshArrays(dict):
    """Plain-array image of mesh::Mesh: the argument list of orc_mesh_create.

    keys: face_c0, face_c1 (-1 = boundary), face_zone, face_area, face_normal[F,3], face_centroid[F,3],
    cell_centroid[n,3], cell_volume, cell_face_ptr, cell_faces, zone_type, zone_scalar, zone_vector[Z,3],
    zone_names.
    """

    @property
    def n_cells(self):
        return len(self["cell_volume"])

    @property
    def n_faces(self):
        return len(self["face_area"])

    def get_face_zone(self, name):
        """mesh.get_face_zone(name) (mesh.rs:189-195): index of the zone; KeyError like the reference's panic."""
        try:
            return self["zone_names"].index(name)
        except ValueError:
            raise KeyError("face zone '%s' should exist in mesh" % name)

    def set_zone(self, name, zone_type, scalar=0.0, vector=(0.0, 0.0, 0.0)):
        k = self.get_face_zone(name)
        self["zone_type"][k] = zone_type
        self["zone_scalar"][k] = scalar
        self["zone_vector"][k] = vector


def splitmix64_uniform(n, seed=0x4F5243, ids=None):
    """uniform[-1,1) f64 from splitmix64, seed "ORC" (SURVEY §8d synthetic inputs); vectorised, stateless.
    ids: the stream positions (0-based) to evaluate instead of 0 .. n-1 — a rank's cells by their GLOBAL ids get the whole mesh's numbers."""
    idx = np.arange(1, n + 1, dtype=np.uint64) if ids is None else np.asarray(ids, dtype=np.uint64) + np.uint64(1)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


ZONE_NAMES = ["FLUID", "INLET", "OUTLET", "PERIODIC_-Z", "PERIODIC_+Z", "TOP_WALL", "BOTTOM_WALL"]


def hex_channel(nx, ny, nz, lx=0.002, ly=0.001, lz=None):
    """Synthetic structured hex channel in ORC's conventions; lz defaults to 0.1 mm per cell."""
    if lz is None:
        lz = 1e-4 * nz
    L = lib()
    nc, nf, ncf = C.c_int64(), C.c_int64(), C.c_int64()
    check(L.orc_hex_channel_sizes(C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.byref(nc), C.byref(nf), C.byref(ncf)))
    n, F, ncf = nc.value, nf.value, ncf.value
    a = MeshArrays(
        face_c0=np.empty(F, np.int64), face_c1=np.empty(F, np.int64), face_zone=np.empty(F, np.int32),
        face_area=np.empty(F), face_normal=np.empty((F, 3)), face_centroid=np.empty((F, 3)),
        cell_centroid=np.empty((n, 3)), cell_volume=np.empty(n), cell_face_ptr=np.empty(n + 1, np.int64),
        cell_faces=np.empty(ncf, np.int64),
        zone_type=np.array([FaceConditionTypes.Interior] + [FaceConditionTypes.Wall] * 6, dtype=np.int32),
        zone_scalar=np.zeros(7), zone_vector=np.zeros((7, 3)), zone_names=list(ZONE_NAMES))
    check(L.orc_hex_channel_generate(
        C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_double(lx), C.c_double(ly), C.c_double(lz),
        a["face_c0"].ctypes.data_as(_I64), a["face_c1"].ctypes.data_as(_I64), a["face_zone"].ctypes.data_as(_I32),
        a["face_area"].ctypes.data_as(_F64), a["face_normal"].ctypes.data_as(_F64), a["face_centroid"].ctypes.data_as(_F64),
        a["cell_centroid"].ctypes.data_as(_F64), a["cell_volume"].ctypes.data_as(_F64),
        a["cell_face_ptr"].ctypes.data_as(_I64), a["cell_faces"].ctypes.data_as(_I64)))
    return a


def write_hex_channel_msh(path, nx, ny, nz, lx=0.002, ly=0.001, lz=None):
    if lz is None:
        lz = 1e-4 * nz
    check(lib().orc_hex_channel_write_msh(path.encode(), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_double(lx),
                                          C.c_double(ly), C.c_double(lz)))


def write_mixed_channel_msh(path, nx, ny, nz, lx=0.002, ly=0.001, lz=None, polyhedra=False):
    """BASELINE config 5 workload (orc_mixed_channel_write_msh): tet / pyramid / prism / hex channel as a TGRID file;
    polyhedra=True (orc_poly_channel_write_msh) adds a region of agglomerated polyhedral cells (12-face rhombic dodecahedra).
    Returns (n_cells, n_faces)."""
    if lz is None:
        lz = 1e-4 * nz
    nc, nf = C.c_int64(), C.c_int64()
    fn = lib().orc_poly_channel_write_msh if polyhedra else lib().orc_mixed_channel_write_msh
    check(fn(path.encode(), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_double(lx), C.c_double(ly),
             C.c_double(lz), C.byref(nc), C.byref(nf)))
    return nc.value, nf.value


def renumber_cells(a, new_of_old):
    """The same mesh with cell `c` renamed new_of_old[c] (faces keep their ids, orientation and per-cell ascending order):
    what a mesh generator with another cell numbering would have written.  Used to measure how much the products depend
    on the numbering (config 5) and to undo it with the RCM ordering of orc_mesh_partition."""
    new_of_old = np.asarray(new_of_old, dtype=np.int64)
    n = len(new_of_old)
    old_of_new = np.empty(n, np.int64)
    old_of_new[new_of_old] = np.arange(n)
    c0, c1 = np.asarray(a["face_c0"]), np.asarray(a["face_c1"])
    cfp, cf = np.asarray(a["cell_face_ptr"]), np.asarray(a["cell_faces"])
    counts = np.diff(cfp)[old_of_new]
    new_cfp = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    # face-list positions of the new cells, one after the other: old start of each cell + offset inside it
    idx = (np.repeat(cfp[:-1][old_of_new], counts) + (np.arange(int(new_cfp[-1]), dtype=np.int64) - np.repeat(new_cfp[:-1], counts))) if n else np.zeros(0, np.int64)
    out = MeshArrays(a)
    out.update(face_c0=new_of_old[c0], face_c1=np.where(c1 >= 0, new_of_old[np.maximum(c1, 0)], -1),
               cell_centroid=np.asarray(a["cell_centroid"])[old_of_new].copy(), cell_volume=np.asarray(a["cell_volume"])[old_of_new].copy(),
               cell_face_ptr=new_cfp, cell_faces=cf[idx].copy())
    return out


def set_channel_bcs(a, top_wall_velocity=0.0, dp_dx=5.0, dx=0.002):
    """Boundary conditions of tests::channel_flow::solve_channel_flow (tests.rs:60-76)."""
    T = FaceConditionTypes
    if "TOP_WALL" in a["zone_names"]:
        a.set_zone("TOP_WALL", T.Wall, 0.0, (top_wall_velocity, 0.0, 0.0))
        a.set_zone("BOTTOM_WALL", T.Wall)
    else:
        a.set_zone("WALL", T.Wall)
    a.set_zone("INLET", T.PressureInlet, -dp_dx * dx)
    a.set_zone("OUTLET", T.PressureOutlet, 0.0)
    a.set_zone("PERIODIC_-Z", T.Symmetry)
    a.set_zone("PERIODIC_+Z", T.Symmetry)
    return a


def set_mixed_channel_bcs(a, top_wall_velocity=0.0, dp_dx=5.0, dx=0.002):
    """tests.rs:60-76 on the zones of orc_mixed_channel_write_msh (every location has a quadrilateral zone and a "_TRI" twin)."""
    T = FaceConditionTypes
    for name in list(a["zone_names"]):
        base = name[:-4] if name.endswith("_TRI") else name
        if base == "WALL":
            a.set_zone(name, T.Wall, 0.0, (top_wall_velocity, 0.0, 0.0))
        elif base == "INLET":
            a.set_zone(name, T.PressureInlet, -dp_dx * dx)
        elif base == "OUTLET":
            a.set_zone(name, T.PressureOutlet, 0.0)
        elif base.startswith("PERIODIC"):
            a.set_zone(name, T.Symmetry)
    return a


class Mesh:
    """Device-resident mesh (OrcMesh*)."""

    def __init__(self, arrays, ordering=None):
        """ordering: None = ORC's numbering; 1 = RCM, 2 = geometric (orc_mesh_create_reordered: internal renumbering,
        fields still in ORC order)."""
        a = arrays
        self.arrays = a
        self._keep = [_i64(a["face_c0"]), _i64(a["face_c1"]), _i32(a["face_zone"]), _f64(a["face_area"]),
                      _f64(a["face_normal"]), _f64(a["face_centroid"]), _f64(a["cell_centroid"]), _f64(a["cell_volume"]),
                      _i64(a["cell_face_ptr"]), _i64(a["cell_faces"]), _i32(a["zone_type"]), _f64(a["zone_scalar"]),
                      _f64(a["zone_vector"])]
        k = self._keep
        st = C.c_int(0)
        args = [C.c_int64(len(k[7])), C.c_int64(len(k[3])), C.c_int32(len(k[10])), k[0].ctypes.data_as(_I64),
                k[1].ctypes.data_as(_I64), k[2].ctypes.data_as(_I32), k[3].ctypes.data_as(_F64), k[4].ctypes.data_as(_F64),
                k[5].ctypes.data_as(_F64), k[6].ctypes.data_as(_F64), k[7].ctypes.data_as(_F64), k[8].ctypes.data_as(_I64),
                k[9].ctypes.data_as(_I64), k[10].ctypes.data_as(_I32), k[11].ctypes.data_as(_F64), k[12].ctypes.data_as(_F64)]
        if ordering:
            lib().orc_mesh_create_reordered.restype = C.c_void_p
            self.ptr = lib().orc_mesh_create_reordered(*args, C.c_int32(int(ordering)), C.byref(st))
        else:
            self.ptr = lib().orc_mesh_create(*args, C.byref(st))
        check(st.value)
        self.ptr = C.c_void_p(self.ptr)
        self._keep = None
        self.n_cells = lib().orc_mesh_n_cells(self.ptr)
        self.nnz = lib().orc_mesh_nnz(self.ptr)

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().orc_mesh_destroy(self.ptr)
            self.ptr = None

    def cell_order(self):
        """global_ids[c] = ORC index of internal cell c (identity unless the mesh was created reordered)"""
        g = np.empty(self.n_cells, np.int64)
        check(lib().orc_mesh_cell_order(self.ptr, g.ctypes.data_as(_I64)))
        return g

    def update_zones(self):
        a = self.arrays
        zt, zs, zv = _i32(a["zone_type"]), _f64(a["zone_scalar"]), _f64(a["zone_vector"])
        check(lib().orc_mesh_update_zones(self.ptr, zt.ctypes.data_as(_I32), zs.ctypes.data_as(_F64), zv.ctypes.data_as(_F64)))

    def matrix_pattern(self):
        rp = np.empty(self.n_cells + 1, np.int64)
        ci = np.empty(self.nnz, np.int64)
        check(lib().orc_mesh_matrix_pattern(self.ptr, rp.ctypes.data_as(_I64), ci.ctypes.data_as(_I64)))
        return rp, ci

    def csr(self, values):
        import scipy.sparse as sp
        rp, ci = self.matrix_pattern()
        return sp.csr_matrix((values, ci, rp), shape=(self.n_cells, self.n_cells))
