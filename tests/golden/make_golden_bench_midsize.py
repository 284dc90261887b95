#!/usr/bin/env python3
"""Generator of tests/golden/bench_midsize_128x64x64.npz: the oracle — the CPU restatement of
ORC's solve_steady (solver.rs:60-222) — on a MID-SIZE member of the bench workload family: the 128 x 64 x 64 hex channel (524 288
cells; bench.py's generator, BCs, settings and initial-field recipe) in the reference's own mode (in-place Rhie-Chow diagonals,
no breakdown guard, nalgebra's reduction order), one, two and three SIMPLE iterations.
`--mixed`: tests/golden/bench_midsize_mixed_60x30x30.npz, the same for the BASELINE configs[4] family — 60 x 30 x 30 blocks of the mixed
tet / pyramid / prism / hex / polyhedral channel (159 510 cells with rows of 5 to 13 entries; parallel.mixed_slab_arrays, set_mixed_channel_bcs).

At this size the device runs code paths no oracle-in-the-test case reaches: 2 048-workgroup grids (8 192 slices > 2 048 x 4), SliceWalk
with several slices per wavefront, 16-bit column bases on slices that straddle z-layers (64 x 128 = 8 192 cells per layer), coarse
levels with real LDS windows (level 2: 131 072 rows, level 3: 65 536 rows), 16-byte-aligned packed mirrors of many blocks.
The fields are too large to commit (4 x 4 MB per iteration), so the file holds per iteration and field: the SHA-256 of the raw f64
bytes (bit-exactness is a hash comparison), a strided sample of exact values (every 1 021st cell: where do they part, if they do),
min / max / sum, and the six report doubles of solver.rs:206-216.

`--frozen` (combinable with `--mixed`): tests/golden/bench_midsize_frozen_*.npz — the ASSEMBLY bench.py times,
i.e. the oracle with frozen_diagonals = 1 (DESIGN D1: face_k<0> + momentum_k<false> on the device), pinned array by array at a size where the
device's launch is the bench's own (524 288 cells = 2 048 workgroups: the XCD-by-XCD cell walk of momentum_k<false>, assembly.hip, is only
taken on grids that are a multiple of 8): a_u, a_v, a_w values, b_u, b_v, b_w, A_p values and b_p of
  assembly 1: bench.initial_fields, incoming matrices from initialize_momentum_matrix (diagonals = 1: SURVEY Q3), and
  assembly 2: helpers.rough_fields (per-cent-level cell-to-cell noise: every TVD branch is live), incoming matrices = assembly 1's (non-trivial
              old diagonals in Rhie-Chow, discretization.rs:184-197)
as SHA-256 + strided samples + min / max / sum; and the fields after ONE whole frozen-diagonal SIMPLE iteration (samples of every 61st cell + norms)
for the product-default comparison at a stated tolerance (only the association of the dot products differs: tests/test_gpu_bench_family.py).

    python tests/golden/make_golden_bench_midsize.py          # about two minutes on one core
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

SHAPE = (128, 64, 64)
STRIDE = 1021
STRIDE_NNZ = 7919  # samples of the matrix value arrays (--frozen)
STRIDE_FIELDS = 61  # samples of the fields after one frozen iteration (--frozen)
KW = dict(momentum=5, solver_type=2, iterations=50, momentum_relaxation=0.1, pressure_relaxation=0.001, frozen_diagonals=0, breakdown_guard=0)


def digest(out, key, x, stride):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out["sha256_" + key] = np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8)
    out["sample_" + key] = x[::stride].copy()
    with np.errstate(invalid="ignore"):
        out["stats_" + key] = np.array([np.nanmin(x), np.nanmax(x), np.nansum(x), float(np.isnan(x).sum())])


def frozen(po, om, a, shape, mixed):
    """The assembly of the timed configuration (frozen diagonals) array by array, and one whole frozen iteration."""
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    kw = dict(KW, frozen_diagonals=1)
    s = po.default_settings(**kw)
    rho, mu = 1000.0, 1e-3
    cc = np.asarray(a["cell_centroid"])
    out = {"shape": np.array(shape), "stride_n": np.array(STRIDE), "stride_nnz": np.array(STRIDE_NNZ), "stride_fields": np.array(STRIDE_FIELDS),
           "settings": np.array(sorted(kw.items()), dtype=object).astype(str)}
    a_di, *_ = po.build_momentum_diffusion_matrix(om, mu)
    mats = [po.initialize_momentum_matrix(om) for _ in range(3)]
    for k, f in ((1, bench.initial_fields(cc)), (2, H.rough_fields(cc))):
        t0 = time.perf_counter()
        u, v, w, p = (np.ascontiguousarray(x) for x in f)
        bu, bv, bw, pe = po.build_momentum_advection_matrices(mats[0], mats[1], mats[2], a_di, om, u, v, w, p, s, rho)
        a_p, b_p = po.build_pressure_correction_matrices(om, u, v, w, p, mats[0], mats[1], mats[2], s, rho)
        for name, m in zip(("a_u", "a_v", "a_w"), mats):
            digest(out, "%s_%d" % (name, k), m.arrays()[2], STRIDE_NNZ)
        digest(out, "a_p_%d" % k, a_p.arrays()[2], STRIDE_NNZ)
        for name, x in (("b_u", bu), ("b_v", bv), ("b_w", bw), ("b_p", b_p)):
            digest(out, "%s_%d" % (name, k), x, STRIDE)
        out["peclet_%d" % k] = np.array(pe)
        print("assembly %d: %.1f s, peclet %s, NaNs in a_w %d" % (k, time.perf_counter() - t0, pe, int(out["stats_a_w_%d" % k][3])), flush=True)
    f = [np.ascontiguousarray(x).copy() for x in bench.initial_fields(cc)]
    t0 = time.perf_counter()
    st, rep = po.solve_steady(om, *f, s, rho, mu, 1, report=True)
    print("one frozen iteration: status %d, %.1f s, report %s" % (st, time.perf_counter() - t0, rep[-1]), flush=True)
    assert st == 0 and all(np.isfinite(x).all() for x in f)
    out["report_frozen_1"] = np.asarray(rep[-1])
    for name, x in zip("uvwp", f):
        out["fields_sha256_%s" % name] = np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8)
        out["fields_sample_%s" % name] = x[::STRIDE_FIELDS].copy()
        out["fields_norm_%s" % name] = np.array([np.linalg.norm(x)])
    # What the association of the dot products alone does to this iteration in the REFERENCE algorithm (r_hat_0 = 1, no guard:
    # linear_algebra.rs:252-268; tests/test_oracle_sensitivity.py): the oracle against itself with every dot / norm summed pairwise.  The
    # product default (wave trees) differs from the oracle in exactly that respect, so this is the scale its deviation is judged against.
    fp = [np.ascontiguousarray(x).copy() for x in bench.initial_fields(cc)]
    po.set_dot_mode(1)
    try:
        st, rep_p = po.solve_steady(om, *fp, s, rho, mu, 1, report=True)
    finally:
        po.set_dot_mode(0)
    assert st == 0
    scale_u = np.linalg.norm(f[0])
    sens = [float(np.linalg.norm(x - y) / (np.linalg.norm(x) if name in "up" else scale_u)) for name, x, y in zip("uvwp", f, fp)]
    out["association_sensitivity"] = np.array(sens)
    out["association_sensitivity_report"] = np.abs(np.asarray(rep_p[-1]) - np.asarray(rep[-1])) / np.maximum(np.abs(np.asarray(rep[-1])), 1e-300)
    print("oracle against itself, pairwise dot products: rel-L2 u, v, w, p =", sens, "report", out["association_sensitivity_report"], flush=True)
    path = os.path.join(ROOT, "tests", "golden", ("bench_midsize_frozen_mixed_%dx%dx%d.npz" if mixed else "bench_midsize_frozen_%dx%dx%d.npz") % shape)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    import bench
    from oracle import pyoracle as po
    from orc_amd.mesh import hex_channel, set_channel_bcs, set_mixed_channel_bcs
    global SHAPE
    mixed = "--mixed" in sys.argv
    po.build()
    if mixed:
        from orc_amd import parallel
        SHAPE = (60, 30, 30)
        _a, _h, _g, a = parallel.mixed_slab_arrays(*SHAPE, 0, 1)
        set_mixed_channel_bcs(a)
    else:
        a = set_channel_bcs(hex_channel(*SHAPE))
    om = po.Mesh.from_arrays(a)
    if "--frozen" in sys.argv:
        return frozen(po, om, a, SHAPE, mixed)
    f0 = bench.initial_fields(np.asarray(a["cell_centroid"]))
    out = {"shape": np.array(SHAPE), "stride": np.array(STRIDE), "settings": np.array(sorted(KW.items()), dtype=object).astype(str)}
    for its in (1, 2, 3):
        f = [np.ascontiguousarray(x).copy() for x in f0]
        t0 = time.perf_counter()
        st, rep = po.solve_steady(om, *f, po.default_settings(**KW), 1000.0, 1e-3, its, report=True)
        print("iterations %d: status %d, %.1f s, report %s" % (its, st, time.perf_counter() - t0, rep[-1]), flush=True)
        assert st == 0 and all(np.isfinite(x).all() for x in f)
        out["report_%d" % its] = np.asarray(rep[-1], dtype=np.float64)
        for name, x in zip("uvwp", f):
            out["sha256_%s_%d" % (name, its)] = np.frombuffer(hashlib.sha256(x.tobytes()).digest(), dtype=np.uint8)
            out["sample_%s_%d" % (name, its)] = x[::STRIDE].copy()
            out["stats_%s_%d" % (name, its)] = np.array([x.min(), x.max(), x.sum()])
    path = os.path.join(ROOT, "tests", "golden", ("bench_midsize_mixed_%dx%dx%d.npz" if mixed else "bench_midsize_%dx%dx%d.npz") % SHAPE)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
