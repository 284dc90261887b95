"""Cell-partitioned multi-GPU runs: one process per GPU, slabs of cells with one layer of ghost cells.

Host-side logic only: building a rank's slab of the synthetic hex channel (owned cells first, ghost blocks after),
the halo plan (who sends which owned cells to whom), and the communicator bootstrap.  The data path is inside
liborc_amd.so: grouped ncclSend/ncclRecv halo exchange and ncclAllReduce of the BiCGSTAB scalars over RCCL/xGMI.
A host-staged debug transport over torch.distributed (gloo) lets the tests run two ranks on ONE GPU.
"""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .mesh import Mesh, MeshArrays, hex_channel, set_channel_bcs

_I64 = C.POINTER(C.c_int64)
_I32 = C.POINTER(C.c_int32)
_F64 = C.POINTER(C.c_double)


# ------------------------------------------------------------------ slab construction (pure numpy: testable on CPU)
def slab_arrays(nx, ny, nz_local, rank, world, lx=0.002, ly=0.001, dz=1e-4):
    """Rank `rank`'s part of the (nx, ny, nz_local*world) hex channel, cut in z-slabs (contiguous in ORC's cell order).

    Returns (MeshArrays of the local mesh, halo dict, global_ids[n_local]).  Local numbering: owned cells in global
    order, then the ghost plane below (peer rank-1), then the ghost plane above (peer rank+1).  Faces: those touching an
    owned cell, ascending global order, global c0/c1 orientation; ghost cells get empty face lists."""
    g_lo = 1 if rank > 0 else 0
    g_hi = 1 if rank < world - 1 else 0
    nzt = nz_local + g_lo + g_hi
    a = hex_channel(nx, ny, nzt, lx=lx, ly=ly, lz=dz * nzt)
    plane = nx * ny
    n_all = plane * nzt
    n_own = plane * nz_local
    old = np.arange(n_all)
    kk = old // plane
    owned = (kk >= g_lo) & (kk < g_lo + nz_local)
    new_of_old = np.empty(n_all, np.int64)
    new_of_old[owned] = old[owned] - g_lo * plane
    if g_lo:
        new_of_old[kk == 0] = n_own + old[kk == 0]
    if g_hi:
        top = kk == nzt - 1
        new_of_old[top] = n_own + g_lo * plane + (old[top] - (nzt - 1) * plane)
    c0, c1 = np.asarray(a["face_c0"]), np.asarray(a["face_c1"])
    keep = owned[c0] | ((c1 >= 0) & owned[np.maximum(c1, 0)])
    new_face = np.cumsum(keep) - 1
    f0 = new_of_old[c0[keep]]
    f1 = np.where(c1[keep] >= 0, new_of_old[np.maximum(c1[keep], 0)], -1)
    n_local = n_own + (g_lo + g_hi) * plane
    # owned cells keep their 6 faces (all kept); ghost cells: none
    cf_old = np.asarray(a["cell_faces"]).reshape(n_all, 6)
    cf = new_face[cf_old[owned]].reshape(-1)
    cfp = np.concatenate([np.arange(0, 6 * n_own + 1, 6), np.full(n_local - n_own, 6 * n_own)]).astype(np.int64)
    order = np.argsort(new_of_old, kind="stable")  # old ids in new order
    z0 = dz * (rank * nz_local - g_lo)
    cc = np.asarray(a["cell_centroid"])[order].copy()
    cc[:, 2] += z0
    fc = np.asarray(a["face_centroid"])[keep].copy()
    fc[:, 2] += z0
    out = MeshArrays(
        face_c0=f0.astype(np.int64), face_c1=f1.astype(np.int64), face_zone=np.asarray(a["face_zone"])[keep].copy(),
        face_area=np.asarray(a["face_area"])[keep].copy(), face_normal=np.asarray(a["face_normal"])[keep].copy(), face_centroid=fc,
        cell_centroid=cc, cell_volume=np.asarray(a["cell_volume"])[order].copy(), cell_face_ptr=cfp, cell_faces=cf.astype(np.int64),
        zone_type=a["zone_type"].copy(), zone_scalar=a["zone_scalar"].copy(), zone_vector=a["zone_vector"].copy(),
        zone_names=list(a["zone_names"]))
    peers, send, recv_ptr = [], [], [0]
    if g_lo:
        peers.append(rank - 1)
        send.append(np.arange(0, plane, dtype=np.int64))             # my bottom owned plane -> their ghost-above block
        recv_ptr.append(recv_ptr[-1] + plane)
    if g_hi:
        peers.append(rank + 1)
        send.append(np.arange(n_own - plane, n_own, dtype=np.int64))  # my top owned plane -> their ghost-below block
        recv_ptr.append(recv_ptr[-1] + plane)
    send_ptr = np.concatenate([[0], np.cumsum([len(s) for s in send])]).astype(np.int64) if send else np.zeros(1, np.int64)
    halo = dict(n_owned=n_own, n_global=plane * nz_local * world, peers=np.array(peers, np.int32), send_ptr=send_ptr,
                send_idx=np.concatenate(send) if send else np.zeros(0, np.int64), recv_ptr=np.array(recv_ptr, np.int64))
    global_ids = (old[order] - g_lo * plane) + rank * n_own  # global cell id of every local cell (ghosts included)
    return out, halo, global_ids


# ------------------------------------------------------------------ general partitioner (orc_mesh_partition, host only)
ORDER_ORC, ORDER_RCM, ORDER_GEOMETRIC = 0, 1, 2


def partition_arrays(a, n_ranks, rank, ordering=ORDER_ORC):
    """Rank `rank`'s part of ANY mesh `a` (MeshArrays of the whole mesh) cut into n_ranks contiguous blocks of `ordering`:
    (MeshArrays of the local mesh, halo dict, global_ids[n_local]) — the same triple slab_arrays returns."""
    L = lib()
    L.orc_mesh_partition.restype = C.c_void_p
    k = [np.ascontiguousarray(a["face_c0"], np.int64), np.ascontiguousarray(a["face_c1"], np.int64),
         np.ascontiguousarray(a["face_zone"], np.int32), np.ascontiguousarray(a["face_area"], np.float64),
         np.ascontiguousarray(a["face_normal"], np.float64), np.ascontiguousarray(a["face_centroid"], np.float64),
         np.ascontiguousarray(a["cell_centroid"], np.float64), np.ascontiguousarray(a["cell_volume"], np.float64),
         np.ascontiguousarray(a["cell_face_ptr"], np.int64), np.ascontiguousarray(a["cell_faces"], np.int64)]
    st = C.c_int(0)
    ptr = L.orc_mesh_partition(C.c_int64(len(k[7])), C.c_int64(len(k[3])), k[0].ctypes.data_as(_I64), k[1].ctypes.data_as(_I64),
                               k[2].ctypes.data_as(_I32), k[3].ctypes.data_as(_F64), k[4].ctypes.data_as(_F64), k[5].ctypes.data_as(_F64),
                               k[6].ctypes.data_as(_F64), k[7].ctypes.data_as(_F64), k[8].ctypes.data_as(_I64), k[9].ctypes.data_as(_I64),
                               C.c_int32(n_ranks), C.c_int32(rank), C.c_int32(ordering), C.byref(st))
    check(st.value)
    return _partition_result(L, ptr, a)


def _partition_result(L, ptr, a):
    """arrays of an OrcPartition handle -> (MeshArrays, halo dict, global_ids); destroys the handle"""
    ptr = C.c_void_p(ptr)
    try:
        no, nl, ng, nf, ncf, ns = (C.c_int64() for _ in range(6))
        npeer = C.c_int32()
        check(L.orc_partition_sizes(ptr, C.byref(no), C.byref(nl), C.byref(ng), C.byref(nf), C.byref(ncf), C.byref(npeer), C.byref(ns)))
        F, n, P = nf.value, nl.value, npeer.value
        out = MeshArrays(
            face_c0=np.empty(F, np.int64), face_c1=np.empty(F, np.int64), face_zone=np.empty(F, np.int32), face_area=np.empty(F),
            face_normal=np.empty((F, 3)), face_centroid=np.empty((F, 3)), cell_centroid=np.empty((n, 3)), cell_volume=np.empty(n),
            cell_face_ptr=np.empty(n + 1, np.int64), cell_faces=np.empty(ncf.value, np.int64),
            zone_type=np.array(a["zone_type"], dtype=np.int32).copy(), zone_scalar=np.array(a["zone_scalar"], dtype=np.float64).copy(),
            zone_vector=np.array(a["zone_vector"], dtype=np.float64).copy(), zone_names=list(a.get("zone_names", [])))
        gids, gfaces = np.empty(n, np.int64), np.empty(F, np.int64)
        peers, sp, si, rp = np.empty(P, np.int32), np.empty(P + 1, np.int64), np.empty(ns.value, np.int64), np.empty(P + 1, np.int64)
        check(L.orc_partition_arrays(ptr, out["face_c0"].ctypes.data_as(_I64), out["face_c1"].ctypes.data_as(_I64),
                                     out["face_zone"].ctypes.data_as(_I32), out["face_area"].ctypes.data_as(_F64),
                                     out["face_normal"].ctypes.data_as(_F64), out["face_centroid"].ctypes.data_as(_F64),
                                     out["cell_centroid"].ctypes.data_as(_F64), out["cell_volume"].ctypes.data_as(_F64),
                                     out["cell_face_ptr"].ctypes.data_as(_I64), out["cell_faces"].ctypes.data_as(_I64),
                                     gids.ctypes.data_as(_I64), gfaces.ctypes.data_as(_I64), peers.ctypes.data_as(_I32),
                                     sp.ctypes.data_as(_I64), si.ctypes.data_as(_I64), rp.ctypes.data_as(_I64)))
    finally:
        L.orc_partition_destroy(ptr)
    halo = dict(n_owned=no.value, n_global=ng.value, peers=peers, send_ptr=sp, send_idx=si, recv_ptr=rp, global_face_ids=gfaces)
    return out, halo, gids


def partition_owner_arrays(a, cell_owner, n_ranks, rank, n_global=-1):
    """orc_mesh_partition_owner: rank `rank`'s part of a mesh `a` the rank generated or read FOR ITSELF (its share plus ghost
    layers); cell_owner[c] = owning rank of every cell or -1 (nobody's: may not touch an owned cell).  Cells keep their order;
    global_ids index `a`.  Returns (MeshArrays, halo, global_ids) like partition_arrays."""
    L = lib()
    L.orc_mesh_partition_owner.restype = C.c_void_p
    k = [np.ascontiguousarray(a["face_c0"], np.int64), np.ascontiguousarray(a["face_c1"], np.int64),
         np.ascontiguousarray(a["face_zone"], np.int32), np.ascontiguousarray(a["face_area"], np.float64),
         np.ascontiguousarray(a["face_normal"], np.float64), np.ascontiguousarray(a["face_centroid"], np.float64),
         np.ascontiguousarray(a["cell_centroid"], np.float64), np.ascontiguousarray(a["cell_volume"], np.float64),
         np.ascontiguousarray(a["cell_face_ptr"], np.int64), np.ascontiguousarray(a["cell_faces"], np.int64)]
    own = np.ascontiguousarray(cell_owner, np.int32)
    assert len(own) == len(k[7])
    st = C.c_int(0)
    ptr = L.orc_mesh_partition_owner(C.c_int64(len(k[7])), C.c_int64(len(k[3])), k[0].ctypes.data_as(_I64), k[1].ctypes.data_as(_I64),
                                     k[2].ctypes.data_as(_I32), k[3].ctypes.data_as(_F64), k[4].ctypes.data_as(_F64), k[5].ctypes.data_as(_F64),
                                     k[6].ctypes.data_as(_F64), k[7].ctypes.data_as(_F64), k[8].ctypes.data_as(_I64), k[9].ctypes.data_as(_I64),
                                     own.ctypes.data_as(_I32), C.c_int32(n_ranks), C.c_int32(rank), C.c_int64(n_global), C.byref(st))
    check(st.value)
    return _partition_result(L, ptr, a)


# ------------------------------------------------------------------ BASELINE configs[4]: rank-local generation of the mixed tet/hex/poly channel
def mixed_slab_arrays(nx, ny, nz_local, rank, world, lx=0.002, ly=0.001, dz=1e-4, polyhedra=True, tmpdir=None):
    """Rank `rank`'s share of the (nx, ny, nz_local * world)-block mixed channel (orc_poly_channel_write_msh: hexahedra, prism
    columns, Kuhn tetrahedra, pyramids and — polyhedra=True — rhombic dodecahedra), cut into slabs of nz_local block layers along z.

    No process ever holds the whole mesh (40 M cells at BASELINE configs[4]): the rank GENERATES its own layers plus two ghost
    layers per inner side — two, because a polyhedral cell reaches half a block into its neighbour layers: the cells of the first
    ghost layer are then complete (true centroid, volume), those of the second exist only to complete them and belong to nobody —
    writes them as a TGRID file in `tmpdir`, reads it back with the product reader (the generator has no in-memory form) and cuts
    its part out with orc_mesh_partition_owner.  The generator is invariant under translation by an EVEN number of block layers
    (the polyhedral region is a checkerboard in i + j + k) and numbers cells layer by layer, so neighbouring ranks see the cells
    they share in the same relative order: ghost blocks and send lists agree without negotiation (checked over gloo by
    tests/test_partition_cpu.py).  Cells belong to the rank whose layers hold their centroid.

    Returns (MeshArrays of the local mesh with BCs unset, halo dict with n_global = -1 (the caller sums n_owned over the ranks),
    global_ids into the rank's generated sub-box, the sub-box arrays)."""
    import os
    import tempfile
    from . import io as orc_io
    from .mesh import write_mixed_channel_msh
    if nz_local % 2 or nz_local < 2:
        raise ValueError("nz_local must be even (the polyhedral checkerboard is invariant under even shifts only)")
    g_lo = 2 if rank > 0 else 0
    g_hi = 2 if rank < world - 1 else 0
    nzt = nz_local + g_lo + g_hi
    k0 = rank * nz_local - g_lo  # global index of the sub-box's first block layer (even)
    # [r05] built in memory (orc_mixed_channel_generate: the reader's own geometry code on the generator's nodes and faces) — r04 wrote a
    # 580 MB TGRID file per rank and read it back: 11.4 of a rank's 12.5 s of set-up.  ORC_MIXED_SLAB_VIA_FILE=1: the old way (same arrays).
    if os.environ.get("ORC_MIXED_SLAB_VIA_FILE") == "1":
        path = os.path.join(tmpdir or tempfile.gettempdir(), "orc_mixed_slab_%d_%d.msh" % (os.getpid(), rank))
        try:
            write_mixed_channel_msh(path, nx, ny, nzt, lx=lx, ly=ly, lz=dz * nzt, polyhedra=polyhedra)
            d = orc_io.read_mesh(path)
        finally:
            if os.path.exists(path):
                os.remove(path)
    else:
        d = orc_io.MeshData.mixed_channel(nx, ny, nzt, lx=lx, ly=ly, lz=dz * nzt, polyhedra=polyhedra)
    a = MeshArrays(d.arrays())
    layer = np.floor(np.asarray(a["cell_centroid"])[:, 2] / dz + 1e-6).astype(np.int64) + k0  # global block layer of every cell
    owner = (layer // nz_local).astype(np.int32)
    owner[(layer < rank * nz_local - 1) | (layer > (rank + 1) * nz_local)] = -1  # the outer ghost layers: nobody's
    owner[(owner < 0) | (owner >= world)] = -1
    a["cell_centroid"] = np.asarray(a["cell_centroid"]).copy()
    a["face_centroid"] = np.asarray(a["face_centroid"]).copy()
    a["cell_centroid"][:, 2] += dz * k0
    a["face_centroid"][:, 2] += dz * k0
    out, halo, gids = partition_owner_arrays(a, owner, world, rank, -1)
    halo["n_global"] = -1
    return out, halo, gids, a


class PartitionedMesh(Mesh):
    """Device mesh of one rank (orc_mesh_create_partitioned)."""

    def __init__(self, arrays, halo):
        a = arrays
        self.arrays = a
        self.halo = halo
        k = [np.ascontiguousarray(a["face_c0"], np.int64), np.ascontiguousarray(a["face_c1"], np.int64),
             np.ascontiguousarray(a["face_zone"], np.int32), np.ascontiguousarray(a["face_area"], np.float64),
             np.ascontiguousarray(a["face_normal"], np.float64), np.ascontiguousarray(a["face_centroid"], np.float64),
             np.ascontiguousarray(a["cell_centroid"], np.float64), np.ascontiguousarray(a["cell_volume"], np.float64),
             np.ascontiguousarray(a["cell_face_ptr"], np.int64), np.ascontiguousarray(a["cell_faces"], np.int64),
             np.ascontiguousarray(a["zone_type"], np.int32), np.ascontiguousarray(a["zone_scalar"], np.float64),
             np.ascontiguousarray(a["zone_vector"], np.float64)]
        peers = np.ascontiguousarray(halo["peers"], np.int32)
        sp, si, rp = (np.ascontiguousarray(halo[x], np.int64) for x in ("send_ptr", "send_idx", "recv_ptr"))
        st = C.c_int(0)
        L = lib()
        L.orc_mesh_create_partitioned.restype = C.c_void_p
        ptr = L.orc_mesh_create_partitioned(
            C.c_int64(halo["n_owned"]), C.c_int64(len(k[7])), C.c_int64(halo["n_global"]), C.c_int64(len(k[3])), C.c_int32(len(k[10])),
            k[0].ctypes.data_as(_I64), k[1].ctypes.data_as(_I64), k[2].ctypes.data_as(_I32), k[3].ctypes.data_as(_F64),
            k[4].ctypes.data_as(_F64), k[5].ctypes.data_as(_F64), k[6].ctypes.data_as(_F64), k[7].ctypes.data_as(_F64),
            k[8].ctypes.data_as(_I64), k[9].ctypes.data_as(_I64), k[10].ctypes.data_as(_I32), k[11].ctypes.data_as(_F64),
            k[12].ctypes.data_as(_F64), C.c_int32(len(peers)), peers.ctypes.data_as(_I32), sp.ctypes.data_as(_I64),
            si.ctypes.data_as(_I64), rp.ctypes.data_as(_I64), C.byref(st))
        check(st.value)
        self.ptr = C.c_void_p(ptr)
        self.n_cells = L.orc_mesh_n_cells(self.ptr)  # local array length (owned + ghost)
        L.orc_mesh_n_owned.restype = C.c_int64
        self.n_owned = L.orc_mesh_n_owned(self.ptr)
        self.nnz = L.orc_mesh_nnz(self.ptr)


# ------------------------------------------------------------------ communicator bootstrap
def init_comm(dist, rank, world):
    """RCCL: rank 0 creates the unique id, torch.distributed broadcasts the 128 bytes, every rank joins."""
    import torch
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        check(lib().orc_comm_get_unique_id(buf))
    t = torch.tensor(list(bytes(buf)), dtype=torch.uint8)
    dist.broadcast(t, src=0)
    raw = bytes(t.tolist())
    check(lib().orc_comm_init((C.c_ubyte * 128).from_buffer_copy(raw), C.c_int(rank), C.c_int(world)))


_EXCHANGE_FN = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_int), _F64, _I64, _I64, _F64, _I64, _I64, C.c_void_p)
_ALLREDUCE_FN = C.CFUNCTYPE(None, _F64, C.c_int, C.c_int, C.c_void_p)
_keepalive = []


def exchange_over_dist(dist, rank, peers, send, send_off, send_cnt, recv, recv_off, recv_cnt):
    """The halo exchange of one field carried by torch.distributed point-to-point (numpy views in, filled in place).
    Lower rank sends first on each pair to avoid a deadlock with blocking gloo sends."""
    import torch
    for q, peer in enumerate(peers):
        s = torch.from_numpy(np.ascontiguousarray(send[send_off[q]:send_off[q] + send_cnt[q]]))
        r = torch.empty(int(recv_cnt[q]), dtype=torch.float64)
        if rank < peer:
            dist.send(s, dst=int(peer))
            dist.recv(r, src=int(peer))
        else:
            dist.recv(r, src=int(peer))
            dist.send(s, dst=int(peer))
        recv[recv_off[q]:recv_off[q] + recv_cnt[q]] = r.numpy()


def init_host_transport(dist, rank, world):
    """Debug transport (tests: ranks sharing one GPU): halos and all-reduces go through host memory and gloo."""
    import torch

    def ex(n_peers, peers, send, send_off, send_cnt, recv, recv_off, recv_cnt, _user):
        ps = [peers[i] for i in range(n_peers)]
        so = [send_off[i] for i in range(n_peers)]
        sc = [send_cnt[i] for i in range(n_peers)]
        ro = [recv_off[i] for i in range(n_peers)]
        rc = [recv_cnt[i] for i in range(n_peers)]
        n_send = max((o + c for o, c in zip(so, sc)), default=0)
        n_recv = max((o + c for o, c in zip(ro, rc)), default=0)
        sv = np.ctypeslib.as_array(send, shape=(max(n_send, 1),))
        rv = np.ctypeslib.as_array(recv, shape=(max(n_recv, 1),))
        exchange_over_dist(dist, rank, ps, sv, so, sc, rv, ro, rc)

    def ar(values, n, op, _user):
        # every rank's terms are gathered and folded in RANK ORDER: the sum of a scalar does not depend on how many scalars travel
        # with it (gloo's ring all-reduce starts every chunk at another rank, so for three or more ranks a 3-scalar reduction and three
        # 1-scalar reductions associate differently) — the lock-step and the one-system schedules stay bit-identical at any world size
        v = np.ctypeslib.as_array(values, shape=(n,))
        parts = [torch.empty(n, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(v.copy()))
        acc = parts[0].numpy().copy()
        for q in range(1, world):
            acc = acc + parts[q].numpy() if op == 0 else np.maximum(acc, parts[q].numpy())
        v[:] = acc

    cex, car = _EXCHANGE_FN(ex), _ALLREDUCE_FN(ar)
    _keepalive.extend([cex, car])
    check(lib().orc_comm_init(None, C.c_int(rank), C.c_int(world)))
    check(lib().orc_comm_set_host_transport(C.cast(cex, C.c_void_p), C.cast(car, C.c_void_p), None))


def finalize():
    check(lib().orc_comm_finalize())


# ------------------------------------------------------------------ bench helper
def make_slab_solver(nx, ny, nz, rank, world, settings, initial_fields, global_noise=False):
    """Every rank owns an nx x ny x nz slab of the (nx, ny, nz*world) channel (weak scaling: nz fixed; strong: nz = the mesh's layers / world).
    global_noise: the initial fields' perturbation is taken by GLOBAL cell id — the run starts from exactly the one-rank field of that channel."""
    from .solver import Solver
    a, halo, gids = slab_arrays(nx, ny, nz, rank, world)
    set_channel_bcs(a)
    mesh = PartitionedMesh(a, halo)
    u, v, w, p = initial_fields(np.asarray(a["cell_centroid"]), ids=gids) if global_noise else initial_fields(np.asarray(a["cell_centroid"]))
    solver = Solver(mesh, settings, 1000.0, 1e-3)
    solver.set_fields(u, v, w, p)
    return solver, mesh, halo["n_global"], mesh.nnz


def verify_ghost_geometry(a, halo, dist, rank, tol):
    """One exchange of geometry over the control plane: every rank sends centroid x, y, z and volume of the cells on its send lists and
    compares what arrives with ITS OWN copy of those cells (its ghost blocks).  The rank-local generators (mixed_slab_arrays) rely on
    neighbouring ranks numbering the cells they share in the same relative order, which orc_mesh_partition_owner cannot check;
    a ghost block in another order would still exchange the right NUMBER of values — and couple the wrong cells.  Raises on a mismatch."""
    peers = [int(q) for q in halo["peers"]]
    if not peers:
        return 0.0
    sp, si, rp = (np.asarray(halo[k], dtype=np.int64) for k in ("send_ptr", "send_idx", "recv_ptr"))
    n_own = int(halo["n_owned"])
    cc, vol = np.asarray(a["cell_centroid"]), np.asarray(a["cell_volume"])
    worst, failure = 0.0, None
    for comp in range(4):  # (every rank goes through all four exchanges whatever it finds: a rank that left early would strand its peers)
        f = np.ascontiguousarray(cc[:, comp] if comp < 3 else vol, dtype=np.float64)
        send = f[si]
        recv = np.empty(int(rp[-1]))
        exchange_over_dist(dist, rank, peers, send, [int(x) for x in sp[:-1]], [int(x) for x in np.diff(sp)], recv, [int(x) for x in rp[:-1]],
                           [int(x) for x in np.diff(rp)])
        mine = f[n_own:n_own + len(recv)]
        scale = max(float(np.max(np.abs(mine))) if len(mine) else 0.0, 1e-300)
        err = float(np.max(np.abs(recv - mine))) / scale if len(mine) else 0.0
        worst = max(worst, err)
        if err > tol and failure is None:
            bad = int(np.argmax(np.abs(recv - mine)))
            failure = ("rank %d: ghost cell %d (local %d) is not the cell its owner sends: %s %r here, %r there (relative %.3e > %.1e) — the "
                       "ranks number their shared cells differently" % (rank, bad, n_own + bad, "xyzV"[comp], float(mine[bad]), float(recv[bad]), err, tol))
    if failure:
        raise RuntimeError(failure)
    return worst


def make_mixed_slab_solver(nx, ny, nz, rank, world, settings, initial_fields, dist=None, polyhedra=True):
    """BASELINE configs[4] as an N-rank run, weak scaling: every rank owns nz block layers of the (nx, ny, nz * world)-block mixed
    tet / hex / poly channel, generated rank-locally (mixed_slab_arrays).  Returns (solver, mesh, cells of the whole mesh, local nnz,
    dict of set-up facts for the bench line)."""
    import time
    from .mesh import set_mixed_channel_bcs
    from .solver import Solver
    t0 = time.perf_counter()
    a, halo, _lids, sub = mixed_slab_arrays(nx, ny, nz, rank, world, polyhedra=polyhedra)
    t_gen = time.perf_counter() - t0
    set_mixed_channel_bcs(a)
    n_global = halo["n_owned"]
    if world > 1:
        import torch
        t = torch.tensor([float(n_global)], dtype=torch.float64)
        dist.all_reduce(t)
        n_global = int(t.item())
    halo["n_global"] = n_global
    ghost_check = verify_ghost_geometry(a, halo, dist, rank, 1e-9) if world > 1 else None  # (slabs are generated in shifted boxes: equal to rounding, not to the bit)
    nfc = np.diff(np.asarray(a["cell_face_ptr"])[:halo["n_owned"] + 1])
    host_bytes = sum(np.asarray(v).nbytes for v in sub.values() if isinstance(v, np.ndarray)) + sum(np.asarray(v).nbytes for v in a.values() if isinstance(v, np.ndarray))
    facts = dict(generated_cells=int(sub.n_cells), owned_cells=int(halo["n_owned"]), ghost_cells=int(len(a["cell_volume"]) - halo["n_owned"]),
                 faces_per_cell={int(k): int(v) for k, v in zip(*np.unique(nfc, return_counts=True))}, generation_s=round(t_gen, 2),
                 host_arrays_gb=round(host_bytes / 1e9, 2), ghost_geometry_check=ghost_check)
    del sub
    if world > 1:
        mesh = PartitionedMesh(a, halo)
    else:
        mesh = Mesh(a)
        mesh.n_owned = mesh.n_cells
    u, v, w, p = initial_fields(np.asarray(a["cell_centroid"]))
    solver = Solver(mesh, settings, 1000.0, 1e-3)
    solver.set_fields(u, v, w, p)
    return solver, mesh, n_global, mesh.nnz, facts
