#!/usr/bin/env python3
"""Phase/stream analysis of one SIMPLE iteration from a rocprofv3 kernel trace (kernel_trace.csv): wall time of the
momentum and p' phases, busy time per stream, GPU non-idle time, kernel classes by summed duration."""
import collections
import csv
import sys

rows = []
for x in csv.DictReader(open(sys.argv[1])):
    rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"].split("(")[0].replace("void ", "").replace("orc::", ""), x["Stream_Id"]))
rows.sort()
marks = [r[0] for r in rows if r[2].startswith("momentum_k")]
a = marks[-1]
b = min(r[1] for r in rows if r[2].startswith("correction_k") and r[0] >= a)  # the iteration ends with the correction
it = [r for r in rows if a <= r[0] <= b]
pk = [r for r in it if r[2] == "pressure_k"]
split = pk[-1][0]
print("last iteration %.1f ms = momentum phase %.1f + p' phase %.1f" % ((b - a) / 1e6, (split - a) / 1e6, (b - split) / 1e6))
for name, ph in (("momentum", [r for r in it if r[0] < split]), ("p'", [r for r in it if r[0] >= split])):
    busy = collections.defaultdict(float)
    cls = collections.defaultdict(float)
    for r in ph:
        busy[r[3]] += (r[1] - r[0]) / 1e6
        cls[r[2].split("<")[0]] += (r[1] - r[0]) / 1e6
    ev = sorted([(r[0], 1) for r in ph] + [(r[1], -1) for r in ph])
    cur, last, tot = 0, None, 0
    for t, d in ev:
        if cur > 0:
            tot += t - last
        cur += d
        last = t
    print("%s: GPU non-idle %.1f ms; busy per stream %s" % (name, tot / 1e6, {k: round(v, 1) for k, v in sorted(busy.items())}))
    print("   ", ", ".join("%s %.0f" % kv for kv in sorted(cls.items(), key=lambda x: -x[1])[:9]))
