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
KW = dict(momentum=5, solver_type=2, iterations=50, momentum_relaxation=0.1, pressure_relaxation=0.001, frozen_diagonals=0, breakdown_guard=0)


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
