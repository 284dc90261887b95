#!/usr/bin/env python3
"""The sequential greedy pairing of linear_algebra.rs:30-60 is a SERIAL DICTATORSHIP: row i takes its most preferred column (value ascending,
position ascending; the diagonal, NaNs and Float::MAX excluded) that no earlier row holds.  With a common priority order on the columns' side
(lower row index wins) the unique stable matching IS that allocation, and row-proposing deferred acceptance reaches it from any order of
proposals, every row moving DOWN its list only.  This script checks that on random and structured matrices with the proposals processed in a
random order (what the device's da_propose_k does with one lane per chain and atomicMin on the holder table)."""
import sys

import numpy as np
import scipy.sparse as sp


def prefs_of(a):
    a = a.tocsr()
    out = []
    for i in range(a.shape[0]):
        cols = a.indices[a.indptr[i]:a.indptr[i + 1]]
        vals = a.data[a.indptr[i]:a.indptr[i + 1]]
        cand = [(v, k, j) for k, (j, v) in enumerate(zip(cols, vals)) if j != i and v < 1.7976931348623157e308]
        cand.sort(key=lambda t: (t[0], t[1]))
        out.append([j for _, _, j in cand])
    return out


def sequential(a):
    n = a.shape[0]
    pr = prefs_of(a)
    combined = set()
    choice = -np.ones(n, dtype=np.int64)
    for i in range(n):
        for j in pr[i]:
            if j not in combined:
                combined.add(j)
                choice[i] = j
                break
    return choice


def deferred_acceptance(a, rng):
    n = a.shape[0]
    pr = prefs_of(a)
    INF = 1 << 62
    holder = np.full(a.shape[1], INF, dtype=np.int64)
    # a pool of "lanes", each following one chain: (row, position of the column it proposes to next)
    lanes = [(i, 0) for i in range(n)]
    steps = 0
    while lanes:
        q = rng.integers(len(lanes))
        r, p = lanes[q]
        if p >= len(pr[r]):
            lanes.pop(q)  # unmatched: the chain ends
            continue
        c = pr[r][p]
        old = holder[c]
        holder[c] = min(old, r)  # atomicMin
        steps += 1
        if old > r:
            if old == INF:
                lanes.pop(q)
            else:  # `old` is displaced: the lane carries on as that row, STATELESS — its next preference after c
                lanes[q] = (old, pr[old].index(c) + 1)
        else:
            lanes[q] = (r, p + 1)
    choice = -np.ones(n, dtype=np.int64)
    for j, h in enumerate(holder):
        if h != INF:
            assert choice[h] < 0
            choice[h] = j
    return choice, steps


def fv_like(nx, ny, nz, seed):
    rng = np.random.default_rng(seed)
    n = nx * ny * nz
    idx = np.arange(n).reshape(nz, ny, nx)
    rows, cols = [np.arange(n)], [np.arange(n)]
    for ax in range(3):
        lo = [slice(None)] * 3
        hi = [slice(None)] * 3
        lo[ax] = slice(0, -1)
        hi[ax] = slice(1, None)
        a, b = idx[tuple(lo)].ravel(), idx[tuple(hi)].ravel()
        rows += [a, b]
        cols += [b, a]
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    vals = np.where(rows == cols, 6.0, -np.round(rng.uniform(0.5, 1.0, len(rows)), 1))  # rounded: ties are common
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, n))


def main():
    rng = np.random.default_rng(7)
    cases = [fv_like(9, 7, 3, 1), fv_like(30, 1, 1, 2), fv_like(12, 12, 2, 3)]
    cases.append(sp.random(200, 200, density=0.05, random_state=5, format="csr") - sp.random(200, 200, density=0.05, random_state=6, format="csr"))
    m = fv_like(8, 8, 2, 4).tolil()
    m[3, 4] = np.nan  # a NaN is never chosen (`coeff < strongest_coeff` is false)
    cases.append(m.tocsr())
    for a in cases:
        a = a.tocsr()
        a.sort_indices()
        ref = sequential(a)
        for trial in range(5):
            got, steps = deferred_acceptance(a, rng)
            assert np.array_equal(ref, got), "deferred acceptance differs from the sequential pairing"
        print("n = %5d: identical in 5 random orders, %d proposals for %d rows" % (a.shape[0], steps, a.shape[0]))


if __name__ == "__main__":
    sys.exit(main())
