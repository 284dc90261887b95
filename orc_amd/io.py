"""io.rs of the reference: read_mesh (io.rs:32-515), read_data / write_data / write_data_with_precision /
write_gradients (io.rs:519-662), carried by liborc_amd.so's host-side C++ (orc_amd/csrc/mesh_io.cpp)."""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .mesh import Mesh, MeshArrays

_F64 = C.POINTER(C.c_double)
_I64 = C.POINTER(C.c_int64)
_I32 = C.POINTER(C.c_int32)


def _p(a, t=_F64):
    return a.ctypes.data_as(t)


class MeshData:
    """Host image of mesh::Mesh as read_mesh builds it (OrcMeshData*)."""

    def __init__(self, mesh_path, _handle=None):
        L = lib()
        if _handle is None:
            L.orc_read_mesh.restype = C.c_void_p
            st = C.c_int(0)
            _handle = L.orc_read_mesh(str(mesh_path).encode(), C.byref(st))
            check(st.value)
        self.ptr = C.c_void_p(_handle)
        dims, nv, nc, nf, ncf, nfn, nz = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        check(L.orc_mesh_data_sizes(self.ptr, C.byref(dims), C.byref(nv), C.byref(nc), C.byref(nf), C.byref(ncf), C.byref(nfn), C.byref(nz)))
        self.dimensions, self.n_vertices, self.n_cells, self.n_faces = dims.value, nv.value, nc.value, nf.value
        self.n_cell_faces, self.n_face_nodes, self.n_zones = ncf.value, nfn.value, nz.value

    def __del__(self):
        if getattr(self, "ptr", None):
            lib().orc_mesh_data_destroy(self.ptr)
            self.ptr = None

    @classmethod
    def mixed_channel(cls, nx, ny, nz, lx=0.002, ly=0.001, lz=None, polyhedra=False):
        """orc_mixed_channel_generate: the mesh of mesh.write_mixed_channel_msh(...) built in memory — bit for bit what reading that file gives."""
        L = lib()
        L.orc_mixed_channel_generate.restype = C.c_void_p
        st = C.c_int(0)
        h = L.orc_mixed_channel_generate(C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_double(lx), C.c_double(ly),
                                         C.c_double(1e-4 * nz if lz is None else lz), C.c_int(1 if polyhedra else 0), C.byref(st))
        check(st.value)
        return cls(None, _handle=h)

    def zones(self):
        """[(zone id, FaceConditionTypes code, scalar_value, vector_value, name)] in file order."""
        out = []
        for k in range(self.n_zones):
            zid, zt, sc = C.c_uint64(), C.c_int32(), C.c_double()
            vec = (C.c_double * 3)()
            name = C.create_string_buffer(256)
            check(lib().orc_mesh_data_zone(self.ptr, C.c_int32(k), C.byref(zid), C.byref(zt), C.byref(sc), vec, name, C.c_int64(256)))
            out.append((zid.value, zt.value, sc.value, tuple(vec), name.value.decode()))
        return out

    def get_face_zone(self, zone_name):
        """Index of the zone called zone_name; the reference panics when there is none (mesh.rs:189-195)."""
        k = lib().orc_mesh_data_zone_index(self.ptr, zone_name.encode())
        if k < 0:
            raise KeyError("face zone '%s' should exist in mesh" % zone_name)
        return k

    def set_zone(self, zone_name, zone_type, scalar_value=0.0, vector_value=(0.0, 0.0, 0.0)):
        vec = (C.c_double * 3)(*vector_value)
        check(lib().orc_mesh_data_set_zone(self.ptr, zone_name.encode(), C.c_int32(int(zone_type)), C.c_double(scalar_value), vec))

    def arrays(self):
        """The argument arrays of orc_mesh_create as a MeshArrays (zone tables included)."""
        F, n = self.n_faces, self.n_cells
        a = MeshArrays(
            face_c0=np.empty(F, np.int64), face_c1=np.empty(F, np.int64), face_zone=np.empty(F, np.int32), face_area=np.empty(F),
            face_normal=np.empty((F, 3)), face_centroid=np.empty((F, 3)), cell_centroid=np.empty((n, 3)), cell_volume=np.empty(n),
            cell_face_ptr=np.empty(n + 1, np.int64), cell_faces=np.empty(self.n_cell_faces, np.int64))
        check(lib().orc_mesh_data_arrays(
            self.ptr, _p(a["face_c0"], _I64), _p(a["face_c1"], _I64), _p(a["face_zone"], _I32), _p(a["face_area"]), _p(a["face_normal"]),
            _p(a["face_centroid"]), _p(a["cell_centroid"]), _p(a["cell_volume"]), _p(a["cell_face_ptr"], _I64), _p(a["cell_faces"], _I64)))
        z = self.zones()
        a["zone_id"] = np.array([t[0] for t in z], np.uint64)
        a["zone_type"] = np.array([t[1] for t in z], np.int32)
        a["zone_scalar"] = np.array([t[2] for t in z], np.float64)
        a["zone_vector"] = np.array([t[3] for t in z], np.float64).reshape(-1, 3)
        a["zone_names"] = [t[4] for t in z]
        return a

    def nodes(self):
        """(vertices[V,3], face_node_ptr[F+1], face_nodes) — Mesh.vertices and Face.node_indices."""
        vert = np.empty((self.n_vertices, 3))
        ptr = np.empty(self.n_faces + 1, np.int64)
        idx = np.empty(self.n_face_nodes, np.int64)
        check(lib().orc_mesh_data_nodes(self.ptr, _p(vert), _p(ptr, _I64), _p(idx, _I64)))
        return vert, ptr, idx

    def upload(self):
        """Device mesh with the current zone table (orc_mesh_upload)."""
        return Mesh(self.arrays())


def read_mesh(mesh_path):
    """io::read_mesh (io.rs:32)."""
    return MeshData(mesh_path)


def write_data(cell_centroid, u, v, w, p, output_file_name, decimal_precision=None):
    """io::write_data (decimal_precision None: "{:.e}") / write_data_with_precision (io.rs:573-620)."""
    cc = np.ascontiguousarray(cell_centroid, np.float64)
    u, v, w, p = (np.ascontiguousarray(x, np.float64) for x in (u, v, w, p))
    check(lib().orc_write_data(str(output_file_name).encode(), C.c_int64(len(u)), _p(cc), _p(u), _p(v), _p(w), _p(p),
                               C.c_int(-1 if decimal_precision is None else int(decimal_precision))))


def read_data(data_file_path):
    """io::read_data (io.rs:519-571) -> (u, v, w, p); OrcError(ORC_ERR_IO) is the reference's Err."""
    n = C.c_int64(0)
    path = str(data_file_path).encode()
    check(lib().orc_read_data(path, C.c_int64(0), None, None, None, None, C.byref(n)))
    u, v, w, p = (np.empty(n.value) for _ in range(4))
    check(lib().orc_read_data(path, C.c_int64(n.value), _p(u), _p(v), _p(w), _p(p), C.byref(n)))
    return u, v, w, p


def write_gradients(mesh, cell_centroid, u, v, w, p, output_file_name, decimal_precision, settings):
    """io::write_gradients (io.rs:623-662); the gradients are computed on the device."""
    cc = np.ascontiguousarray(cell_centroid, np.float64)
    u, v, w, p = (np.ascontiguousarray(x, np.float64) for x in (u, v, w, p))
    check(lib().orc_write_gradients(mesh.ptr, _p(cc), _p(u), _p(v), _p(w), _p(p), str(output_file_name).encode(),
                                    C.c_int(int(decimal_precision)), C.byref(settings)))
