#!/usr/bin/env python3
"""Idle time between consecutive kernels of one rocprofv3 --kernel-trace csv: per (previous kernel -> next kernel) pair the number of
boundaries, the mean / median gap and the total; gaps longer than --cut microseconds (host-side pauses between phases) are listed apart.
    python scripts/analysis/kernel_gaps.py <kernel_trace.csv> [--cut 200] [--top 25]"""
import argparse
import collections
import csv
import re
import statistics


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("orc::", "")
    return name[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--cut", type=float, default=200.0)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    rows = []
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    pairs = collections.defaultdict(list)
    long_gaps = []
    end = rows[0][1]
    prev = rows[0][2]
    overlap = 0
    for s, e, n in rows[1:]:
        g = (s - end) / 1e3
        if g < 0:
            overlap += 1
        elif g > a.cut:
            long_gaps.append((g, prev, n))
        else:
            pairs[(prev, n)].append(g)
        if e > end:
            end, prev = e, n
    total_gap = sum(sum(v) for v in pairs.values())
    print("kernels %d, span %.1f ms, sum of durations %.1f ms, gaps <= %.0f us: %.1f ms in %d boundaries (mean %.2f us), longer: %d (%.1f ms), overlapping starts: %d"
          % (len(rows), span / 1e6, busy / 1e6, a.cut, total_gap / 1e3, sum(len(v) for v in pairs.values()),
             total_gap / max(1, sum(len(v) for v in pairs.values())), len(long_gaps), sum(g for g, _, _ in long_gaps) / 1e3, overlap))
    print("%-62s %-62s %7s %8s %8s %9s" % ("previous", "next", "count", "mean us", "median", "total ms"))
    for (p, n), v in sorted(pairs.items(), key=lambda kv: -sum(kv[1]))[:a.top]:
        print("%-62s %-62s %7d %8.2f %8.2f %9.2f" % (p, n, len(v), statistics.mean(v), statistics.median(v), sum(v) / 1e3))
    for g, p, n in sorted(long_gaps, reverse=True)[:10]:
        print("long gap %.0f us: %s -> %s" % (g, p, n))


if __name__ == "__main__":
    main()
