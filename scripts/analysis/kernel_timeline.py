#!/usr/bin/env python3
"""Where the wall time of a SIMPLE iteration goes when several streams run: a rocprofv3 --kernel-trace csv cut into intervals by what is
resident — solve kernels (products, BiCGSTAB updates, restriction / prolongation), set-up kernels (pairing, Galerkin, mirrors), assembly,
copies — alone or together, and idle.  The trace is cut at gaps longer than --cut microseconds: the last --steps pieces are reported.
    python scripts/analysis/kernel_timeline.py <kernel_trace.csv> [--from-kernel momentum_k]"""
import argparse
import collections
import csv
import re

SETUP = ("da_", "agg_", "galerkin_", "xwin_", "narrow_build", "scan_", "slice_sizes", "rows_compact", "scale_packed", "scale_values", "chooser", "sell_", "pack_")
SOLVE = ("spmv", "bicg_", "restrict", "prolong", "reduce_partials", "vec_", "fill_k", "diag_inverse", "scale_vec", "nan_to", "guard_event", "interleave", "deinterleave", "residual")


def kind(name):
    n = re.sub(r"\(.*", "", name).replace("void ", "").replace("orc::", "")
    if n.startswith("__amd_rocclr"):
        return "copy"
    if n.startswith(SETUP):
        return "setup"
    if n.startswith(SOLVE):
        return "solve"
    return "assembly"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--from-kernel", default="momentum_k", help="an iteration starts at the first launch of this kernel after another kind of work")
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"]), r["Kernel_Name"]))
    rows.sort()
    starts = [s for s, e, k, n in rows if a.from_kernel in n]
    # one iteration = from one launch of the marker kernel to the next (markers closer than 50 ms belong together)
    marks = []
    for s in starts:
        if not marks or s - marks[-1] > 50e6:
            marks.append(s)
    marks.append(rows[-1][1])
    for i in range(len(marks) - 1):
        lo, hi = marks[i], marks[i + 1]
        ev = []
        for s, e, k, n in rows:
            if e <= lo or s >= hi:
                continue
            ev.append((max(s, lo), 1, k))
            ev.append((min(e, hi), -1, k))
        ev.sort()
        active = collections.Counter()
        t_prev = lo
        acc = collections.Counter()
        for t, d, k in ev:
            if t > t_prev:
                state = "+".join(sorted(x for x in active if active[x] > 0)) or "idle"
                acc[state] += t - t_prev
                t_prev = t
            active[k] += d
        if hi > t_prev:
            acc["idle"] += hi - t_prev
        total = hi - lo
        print("iteration %d: %.1f ms" % (i, total / 1e6) + "".join("  | %s %.1f" % (k, v / 1e6) for k, v in acc.most_common()))


if __name__ == "__main__":
    main()
