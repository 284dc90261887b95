"""Worker for the multi-rank tests (launched by torch.distributed.run with 2+ ranks; see test_partition_*.py).

mode cpu : no GPU — slab construction + halo plan + the gloo exchange that also serves as liborc_amd's debug transport.
mode gpu : ranks share cuda:0 through the host-staged transport; partitioned SIMPLE iterations vs the single-rank run.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from orc_amd import parallel  # noqa: E402
from orc_amd.mesh import hex_channel, set_channel_bcs, splitmix64_uniform  # noqa: E402


def global_fields(a):
    cc = np.asarray(a["cell_centroid"])
    n = len(cc)
    y = cc[:, 1]
    u = 1.0 / 2e-3 * 5.0 * (y * y - 1e-3 * y) * (1 + 0.01 * splitmix64_uniform(n, 1))
    v = 1e-6 * splitmix64_uniform(n, 2)
    w = 1e-6 * splitmix64_uniform(n, 3)
    p = -0.01 * (1 - cc[:, 0] / 0.002) * (1 + 0.01 * splitmix64_uniform(n, 4))
    return u, v, w, p


def main():
    mode = sys.argv[1]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    nx, ny, nzl = 6, 5, 3
    a, halo, gids = parallel.slab_arrays(nx, ny, nzl, rank, world)
    ag = hex_channel(nx, ny, nzl * world)  # the global mesh (every rank builds it: small)
    n_own = halo["n_owned"]
    ok = True
    # ---- structure of the slab against the global mesh
    assert np.array_equal(gids[:n_own], rank * n_own + np.arange(n_own))
    assert np.allclose(np.asarray(a["cell_centroid"]), np.asarray(ag["cell_centroid"])[gids], rtol=0, atol=1e-18)
    assert np.allclose(np.asarray(a["cell_volume"]), np.asarray(ag["cell_volume"])[gids], rtol=1e-14)
    # every owned cell keeps its six faces, in the global relative order and with the global orientation
    gf = np.asarray(ag["cell_faces"]).reshape(-1, 6)
    lf = np.asarray(a["cell_faces"]).reshape(n_own, 6)
    for c in range(n_own):
        gfaces, lfaces = gf[gids[c]], lf[c]
        assert np.all(np.diff(lfaces) > 0)
        for fg, fl in zip(gfaces, lfaces):
            assert gids[a["face_c0"][fl]] == ag["face_c0"][fg]
            c1g = ag["face_c1"][fg]
            assert (a["face_c1"][fl] < 0 and c1g < 0) or gids[a["face_c1"][fl]] == c1g
            assert np.array_equal(np.asarray(a["face_normal"])[fl], np.asarray(ag["face_normal"])[fg])
    # ---- halo exchange over gloo: ghost entries must equal the owner's values
    x = np.full(len(gids), np.nan)
    truth = 1000.0 + np.arange(nx * ny * nzl * world, dtype=np.float64) * 0.5
    x[:n_own] = truth[gids[:n_own]]
    peers = list(halo["peers"])
    sp, rp = halo["send_ptr"], halo["recv_ptr"]
    send = x[halo["send_idx"]]
    recv = np.empty(int(rp[-1]))
    parallel.exchange_over_dist(dist, rank, peers, send, list(sp[:-1]), list(np.diff(sp)), recv, list(rp[:-1]), list(np.diff(rp)))
    x[n_own:] = recv
    assert np.array_equal(x, truth[gids]), "ghost values differ from the owners'"
    # ---- distributed SpMV == global SpMV (host arithmetic on the local pattern incl. ghost columns)
    import scipy.sparse as sps
    n_loc = len(gids)
    rows, cols = [], []
    c0, c1 = np.asarray(a["face_c0"]), np.asarray(a["face_c1"])
    for f in range(len(c0)):
        if c1[f] >= 0:
            for r_, c_ in ((c0[f], c1[f]), (c1[f], c0[f])):
                if r_ < n_own:
                    rows.append(r_); cols.append(c_)
    A_loc = sps.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n_own, n_loc)) + sps.eye(n_own, n_loc) * 7.0
    c0g, c1g = np.asarray(ag["face_c0"]), np.asarray(ag["face_c1"])
    m = c1g >= 0
    ng = len(truth)
    A_glob = sps.csr_matrix((np.ones(2 * m.sum()), (np.r_[c0g[m], c1g[m]], np.r_[c1g[m], c0g[m]])), shape=(ng, ng)) + sps.eye(ng) * 7.0
    assert np.array_equal(A_loc @ x, (A_glob @ truth)[gids[:n_own]])
    if mode == "gpu":
        ok = gpu_checks(rank, world, a, halo, gids, ag) and ok
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    dist.barrier()
    if rank == 0:
        print("MP_WORKER_OK" if t.item() == 1.0 else "MP_WORKER_FAIL", flush=True)
    dist.destroy_process_group()


def gpu_checks(rank, world, a, halo, gids, ag):
    import orc_amd
    from orc_amd.mesh import Mesh
    from orc_amd.settings import NumericalSettings
    from orc_amd.solver import Solver
    orc_amd.init(0)
    parallel.init_host_transport(dist, rank, world)
    set_channel_bcs(a)
    set_channel_bcs(ag)
    ug = global_fields(ag)
    n_own = halo["n_owned"]
    results = {}
    mg_iters = int(os.environ.get("ORC_MG_ITERS", "30"))
    # Jacobi/BiCGSTAB: same algorithm, rows next to the cut add their ghost column last instead of in global column
    # order (1e-15 effects).  Multigrid: aggregates never cross the cut and coarse levels are solved per rank, so the
    # preconditioning differs by design while the outer system is the same.
    for name, kw, tol in (("jacobi", dict(momentum=0, solver_type=1, relative_convergence_threshold=1e-30), 1e-12),
                          ("bicgstab", dict(momentum=5, solver_type=3, iterations=5), 1e-9),
                          ("multigrid", dict(momentum=1, solver_type=2, iterations=mg_iters), float(os.environ.get("ORC_MG_TOL", "1e-5"))),
                          # GS extension: Gauss-Seidel inside a rank, Jacobi across the cut (ghosts refreshed once per sweep)
                          ("bicgstab_gs", dict(momentum=4, solver_type=17, iterations=mg_iters), 1e-5),
                          # GS sweeps as the smoother are far from converged after a fixed count, and a rank's sweep order differs
                          # from the single-rank one (own colouring, Jacobi across the cut): same fixed point, different iterates —
                          # a sanity bound only
                          ("multigrid_gs", dict(momentum=1, solver_type=18, iterations=mg_iters), 0.5)):
        s = NumericalSettings.default(**kw)
        pm = parallel.PartitionedMesh(a, halo)
        sol = Solver(pm, s, 1000.0, 1e-3)
        sol.set_fields(*[f[gids] for f in ug])
        st = sol.iterate(2, raise_on_error=False)
        loc = sol.get_fields()
        # reference: the whole mesh on one rank (no halo: reductions stay local)
        gm = Mesh(ag)
        ref = Solver(gm, s, 1000.0, 1e-3)
        ref.set_fields(*ug)
        st_ref = ref.iterate(2, raise_on_error=False)
        glob = ref.get_fields()
        err = max(np.linalg.norm(l[:n_own] - g[gids[:n_own]]) / max(np.linalg.norm(g), 1e-300) for l, g in zip(loc, glob))
        good = (st == st_ref == 0) and err <= tol and all(np.isfinite(f).all() for f in loc)
        if rank == 0:
            print("  %-10s status %d/%d  max rel err vs single rank %.3e  %s" % (name, st, st_ref, err, "ok" if good else "FAIL"), flush=True)
        results[name] = good
        if name == "multigrid":
            # the partitioned momentum lanes (hierarchies built beside the level-0 work, every collective still issued by
            # this thread on the library stream) against everything on one stream: identical bits
            os.environ["ORC_CONCURRENT_MOMENTUM"] = "0"
            os.environ["ORC_EARLY_P_HIERARCHY"] = "0"
            os.environ["ORC_TWO_STREAM_MULTIGRID"] = "0"
            try:
                seq = Solver(pm, s, 1000.0, 1e-3)
                seq.set_fields(*[f[gids] for f in ug])
                st_seq = seq.iterate(2, raise_on_error=False)
                same = st_seq == st and all(np.array_equal(x[:n_own], y[:n_own]) for x, y in zip(seq.get_fields(), loc))
            finally:
                for k in ("ORC_CONCURRENT_MOMENTUM", "ORC_EARLY_P_HIERARCHY", "ORC_TWO_STREAM_MULTIGRID"):
                    del os.environ[k]
            flags = [None] * world
            dist.all_gather_object(flags, bool(same))
            if rank == 0:
                print("  %-10s lanes vs one stream, per rank: %s  %s" % (name, flags, "ok" if all(flags) else "FAIL"), flush=True)
            results["multigrid_lanes"] = all(flags)
    parallel.finalize()
    return all(results.values())


if __name__ == "__main__":
    main()
