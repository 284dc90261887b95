#!/usr/bin/env python3
"""The mesh pattern of the config-5 slab (or a hex channel) as the SELL-64 image sees it, on the CPU: row lengths, padding per slice, and per
(slice, depth) the span of the columns — how many slices the narrow column image (16-bit offsets from a per-depth base) cannot hold, and what a
stable sort of the rows of every window of W slices by length would do to the padding.
    python scripts/analysis/sell_pattern_stats.py [--nx 252 --ny 100 --nz 72]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), "..", ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=252)
    ap.add_argument("--ny", type=int, default=100)
    ap.add_argument("--nz", type=int, default=72)
    a = ap.parse_args()
    from orc_amd import parallel
    t0 = time.time()
    _a, _h, _g, arr = parallel.mixed_slab_arrays(a.nx, a.ny, a.nz, 0, 1)
    c0, c1 = np.asarray(arr["face_c0"]), np.asarray(arr["face_c1"])
    n = len(arr["cell_volume"])
    print("generated %d cells in %.1f s" % (n, time.time() - t0))
    interior = c1 >= 0
    r = np.concatenate([c0[interior], c1[interior], np.arange(n)])
    c = np.concatenate([c1[interior], c0[interior], np.arange(n)])
    order = np.lexsort((c, r))
    r, c = r[order], c[order]
    ptr = np.zeros(n + 1, np.int64)
    np.add.at(ptr, r + 1, 1)
    ptr = np.cumsum(ptr)
    length = np.diff(ptr)
    print("nnz %d, row lengths:" % len(c), dict(zip(*np.unique(length, return_counts=True))))
    ns = (n + 63) // 64
    pad = np.zeros(ns * 64, np.int64)
    pad[:n] = length
    width = pad.reshape(ns, 64).max(axis=1)
    print("SELL-64 padded entries %d (x %.4f)" % (width.sum() * 64, width.sum() * 64 / len(c)))
    for W in (4, 16, 64, 256):
        q = pad.copy()
        blk = W * 64
        m = (len(q) // blk) * blk
        q[:m] = np.sort(q[:m].reshape(-1, blk), axis=1).reshape(-1)
        print("rows of every %d slices sorted by length: padded x %.4f" % (W, q.reshape(ns, 64).max(axis=1).sum() * 64 / len(c)))
    # span per (slice, depth)
    k_of = np.arange(len(c)) - ptr[r]
    wide = 0
    wmax = int(width.max())
    sl = r // 64
    bad = np.zeros(ns, bool)
    for k in range(wmax):
        m = k_of == k
        lo = np.full(ns, np.iinfo(np.int64).max)
        hi = np.full(ns, -1)
        np.minimum.at(lo, sl[m], c[m])
        np.maximum.at(hi, sl[m], c[m])
        b = (hi >= 0) & (hi - lo > 65535)
        bad |= b
        print("depth %2d: %d of %d slices span more than 65 535 columns (largest span %d)" % (k, b.sum(), (hi >= 0).sum(), (hi - lo)[hi >= 0].max()))
    print("slices with any such depth: %d of %d" % (bad.sum(), ns))


if __name__ == "__main__":
    main()
