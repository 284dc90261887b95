#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: the mixed tet / pyramid / prism / hex channel (orc_mixed_channel_write_msh -> product
reader), the product y = A x of the momentum system and full SIMPLE iterations with the default stack (Multigrid +
Jacobi-preconditioned BiCGSTAB smoothing) under three cell numberings: the generator's, a random permutation (an
"arbitrarily numbered" mesh), and that permutation undone internally by the RCM ordering of orc_mesh_create_reordered
(fields stay in ORC order).  Writes one CSV row per numbering."""
import argparse, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import orc_amd
from orc_amd import _lib
from orc_amd import io as orc_io
from orc_amd.mesh import Mesh, MeshArrays, renumber_cells, set_mixed_channel_bcs, splitmix64_uniform, write_mixed_channel_msh
from orc_amd.settings import NumericalSettings
from orc_amd.solver import Solver

ap = argparse.ArgumentParser()
# default: the per-GPU share of BASELINE config 5 (40 M cells on 8 GPUs = 5 M cells), polyhedral region included
ap.add_argument("--nx", type=int, default=252); ap.add_argument("--ny", type=int, default=100); ap.add_argument("--nz", type=int, default=72)
ap.add_argument("--polyhedra", type=int, default=1)
ap.add_argument("--iterations", type=int, default=2)
ap.add_argument("--csv", default=None)
args = ap.parse_args()
orc_amd.init(0)
path = os.path.join(tempfile.gettempdir(), "orc_config5_%d.msh" % os.getpid())
t0 = time.perf_counter()
nc, nf = write_mixed_channel_msh(path, args.nx, args.ny, args.nz, polyhedra=bool(args.polyhedra))
t_write = time.perf_counter() - t0
t0 = time.perf_counter()
d = orc_io.read_mesh(path)
t_read = time.perf_counter() - t0
size_mb = os.path.getsize(path) / 1e6
os.remove(path)
a = MeshArrays(d.arrays())
set_mixed_channel_bcs(a)
n = a.n_cells
print("mixed channel %dx%dx%d blocks: %d cells, %d faces; .msh %.0f MB written in %.1f s, read in %.1f s" % (args.nx, args.ny, args.nz, nc, nf, size_mb, t_write, t_read), flush=True)
nfc = np.diff(a["cell_face_ptr"])
print("faces per cell:", dict(zip(*[x.tolist() for x in np.unique(nfc, return_counts=True)])), flush=True)
cc = np.asarray(a["cell_centroid"])
y = cc[:, 1]
u0 = 1.0 / 2e-3 * 5.0 * (y * y - 1e-3 * y) * (1 + 1e-6 * splitmix64_uniform(n, 1))
v0 = 1e-12 * splitmix64_uniform(n, 2); w0 = 1e-12 * splitmix64_uniform(n, 3)
p0 = -0.01 * (1 - cc[:, 0] / 0.002) * (1 + 1e-6 * splitmix64_uniform(n, 4))
perm = np.random.default_rng(7).permutation(n)
shuffled = renumber_cells(a, perm)
rows = []
for label, arrays, ordering, fields in (("generator numbering", a, None, (u0, v0, w0, p0)),
                                        ("random numbering", shuffled, None, None),
                                        ("random numbering + internal RCM", shuffled, 1, None)):
    if fields is None:  # the same physical fields on the shuffled mesh: new index perm[c] holds old cell c
        inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
        fields = tuple(f[inv] for f in (u0, v0, w0, p0))
    t0 = time.perf_counter()
    mesh = Mesh(arrays, ordering=ordering)
    t_mesh = time.perf_counter() - t0
    s = Solver(mesh, NumericalSettings.default(momentum=5, momentum_relaxation=0.1, pressure_relaxation=0.001), 1000.0, 1e-3)
    s.set_fields(*fields)
    st0 = s.iterate(1, raise_on_error=False)
    t0 = time.perf_counter()
    st = s.iterate(args.iterations, raise_on_error=False)
    dt = (time.perf_counter() - t0) / args.iterations
    free_b, total_b = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().orc_device_memory(C.byref(free_b), C.byref(total_b)))
    hbm_gb = (total_b.value - free_b.value) / 1e9
    ms, _ = s.bench_spmv(50)
    inloop = s.bench_inloop_products(50)
    nnz = mesh.nnz
    bytes_ = 12.0 * nnz + 20.0 * n
    rp, ci = mesh.matrix_pattern()
    bw = int(np.abs(np.repeat(np.arange(n), np.diff(rp)) - ci).max())
    ms_in = 0.5 * (inloop[0] + inloop[1])
    print("%-34s mesh upload %.1f s  status %d/%d  %.1f ms per SIMPLE iteration  product %.1f us = %.0f GB/s = %.3f of 8 TB/s; in the BiCGSTAB loop %.1f us = %.3f; "
          "three systems per launch %.1f us = %.3f  (nnz %d, bandwidth %d, device memory in use %.1f GB)"
          % (label, t_mesh, st0, st, dt * 1e3, ms * 1e3, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 8000, ms_in * 1e3, bytes_ / ms_in / 1e6 / 8000,
             0.5 * (inloop[2] + inloop[3]) * 1e3, 3 * bytes_ / (0.5 * (inloop[2] + inloop[3]) + 1e-30) / 1e6 / 8000, nnz, bw, hbm_gb), flush=True)
    rows.append((label, n, nnz, bw, ms * 1e3, bytes_ / ms / 1e6, bytes_ / ms / 1e6 / 8000, ms_in * 1e3, bytes_ / ms_in / 1e6 / 8000, dt * 1e3, hbm_gb, st))
    del s, mesh
if args.csv:
    with open(args.csv, "w") as fh:
        fh.write("numbering,cells,nnz,matrix_bandwidth,plain_spmv_us,plain_spmv_GB_per_s,plain_frac_of_8TBs,inloop_spmv_us,inloop_frac_of_8TBs,ms_per_simple_iteration,device_memory_GB,status\n")
        for r in rows:
            fh.write("%s,%d,%d,%d,%.1f,%.0f,%.4f,%.1f,%.4f,%.1f,%.1f,%d\n" % r)
