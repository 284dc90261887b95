#!/usr/bin/env python3
"""Durations of a rocprofv3 --kernel-trace csv grouped by (kernel, grid size, LDS size): count, mean microseconds, total milliseconds.
    python scripts/analysis/kernel_by_grid.py <kernel_trace.csv> [--match bicg] [--top 40]"""
import argparse
import collections
import csv
import re


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("orc::", "")[:64]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--match", default="")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    groups = collections.defaultdict(list)
    with open(a.trace) as f:
        for r in csv.DictReader(f):
            n = short(r["Kernel_Name"])
            if a.match and a.match not in n:
                continue
            groups[(n, int(r["Grid_Size_X"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in groups.values())
    print("%-66s %10s %6s %5s %7s %9s %9s %6s" % ("kernel", "grid", "lds", "vgpr", "count", "mean us", "total ms", "%"))
    for (n, g, l, vg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1]))[:a.top]:
        print("%-66s %10d %6d %5d %7d %9.1f %9.2f %6.2f" % (n, g, l, vg, len(v), sum(v) / len(v), sum(v) / 1e3, 100 * sum(v) / total))


if __name__ == "__main__":
    main()
