"""Offline check behind DESIGN.md §3: distinct 128-byte cache lines of x touched per gather wave-instruction on each level of
the momentum hierarchy (oracle-built, 100x40x40 channel) for three lane mappings: SELL-64 (64 rows at one depth), 4 rows x 16
consecutive entries, and 64 consecutive entries of the CSR stream.  Run from the repo root on the CPU."""
import sys, numpy as np, scipy.sparse as sp
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import pyoracle as po
from orc_amd.mesh import hex_channel, set_channel_bcs
import bench
nx,ny,nz = 100,40,40
a = set_channel_bcs(hex_channel(nx,ny,nz))
om = po.Mesh.from_arrays(a)
u,v,w,p = bench.initial_fields(np.asarray(a["cell_centroid"]))
s = po.default_settings(momentum=5, frozen_diagonals=1)
adi, bu, bv, bw = po.build_momentum_diffusion_matrix(om, 1e-3)
au, av, aw = po.initialize_momentum_matrix(om), po.initialize_momentum_matrix(om), po.initialize_momentum_matrix(om)
po.build_momentum_advection_matrices(au, av, aw, adi, om, u, v, w, p, s, 1000.0)
A = au.to_scipy().tocsr()
d = A.diagonal(); A = sp.diags(1.0/d) @ A; A = A.tocsr(); A.sort_indices()
def stats(A, name):
    n = A.shape[0]; rp, ci = A.indptr, A.indices
    lens = np.diff(rp)
    # SELL mapping: per slice, depth k: distinct 128B lines (16 doubles) among lanes with len>k
    tot_instr = 0; tot_lines = 0; tot_lanes=0
    for s0 in range(0, min(n, 64*300), 64):
        rows = range(s0, min(n, s0+64))
        w = max(lens[r] for r in rows)
        for k in range(w):
            cols = [ci[rp[r]+k] for r in rows if lens[r] > k]
            tot_instr += 1; tot_lines += len(set(c//16 for c in cols)); tot_lanes += len(cols)
    sell = (tot_lines/tot_instr, tot_lanes/tot_instr)
    # row-chunk mapping: 4 rows x 16 consecutive entries
    ti=0; tl=0; tn=0
    for s0 in range(0, min(n, 64*300), 64):
        rows = list(range(s0, min(n, s0+64)))
        w = max(lens[r] for r in rows)
        for k0 in range(0, w, 16):
            for g in range(0, len(rows), 4):
                cols=[]
                for r in rows[g:g+4]:
                    cols += list(ci[rp[r]+k0: rp[r]+min(lens[r], k0+16)])
                if cols:
                    ti+=1; tl+=len(set(c//16 for c in cols)); tn+=len(cols)
    # wave-per-row-pair: 64 consecutive entries of the CSR stream
    t3=0; l3=0
    lo = rp[0]; hi = rp[min(n,64*300)]
    for e0 in range(lo, hi, 64):
        cols = ci[e0:e0+64]; t3+=1; l3+=len(set(c//16 for c in cols))
    print("%s: n=%d nnz/row=%.1f | SELL: %.1f lines/instr (%.1f active lanes) | 4x16 chunks: %.1f lines/instr (%.1f lanes) | CSR stream 64: %.1f lines/instr" % (name, n, A.nnz/n, sell[0], sell[1], tl/ti, tn/ti, l3/t3))
stats(A, "level0")
cur = po.Csr.from_scipy(A)
for lvl in (1,2,3):
    R = po.build_restriction_matrix(cur)
    cur = R.matmul(cur).matmul(R.transpose())
    M = cur.to_scipy().tocsr(); M.sort_indices()
    stats(M, "level%d"%lvl)
