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
corr = [r for r in rows if r[2].startswith("correction_k")]
a = max(m for m in marks if any(c[0] >= m for c in corr))  # the last momentum assembly that a correction follows (bench.py assembles once more for its product timings)
b = min(r[1] for r in corr if r[0] >= a)  # the iteration ends with the correction
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

# ---- three-system schedule: when each family of kernels runs inside the momentum phase (ms from its start)
def family(name):
    if name.startswith("spmv3_uniform_k") or name.startswith("bicg_") and "3_k" in name:
        return "L0/L1 lock-step products" if "spmv3" in name else "lock-step vector kernels"
    if name.startswith("spmv_xwin_k") or name.startswith("spmv_k"):
        return "levels 2-3 products"
    if name.startswith("spmv_uniform_k"):
        return "one-system level 0/1 products"
    if name.startswith(("agg_", "tail_", "chase_", "chooser_k")):
        return "aggregation"
    if name.startswith(("galerkin_", "xwin_build", "slice_sizes", "scan")):
        return "Galerkin + mirrors"
    if name.startswith("bicg_"):
        return "one-system vector kernels"
    return None

mom = [r for r in it if r[0] < split]
fam = collections.defaultdict(list)
for r in mom:
    f = family(r[2])
    if f:
        fam[f].append(r)
print("momentum phase by kernel family (first start .. last end, ms from the phase start; summed kernel time; launches):")
for f, rs in sorted(fam.items(), key=lambda kv: min(r[0] for r in kv[1])):
    print("  %-32s %7.1f .. %7.1f   %7.1f ms  %6d" % (f, (min(r[0] for r in rs) - a) / 1e6, (max(r[1] for r in rs) - a) / 1e6, sum(r[1] - r[0] for r in rs) / 1e6, len(rs)))
l0 = [r for r in mom if r[2].startswith("spmv3_uniform_k") and r[2].rstrip().endswith("true>")]
l1 = [r for r in mom if r[2].startswith("spmv3_uniform_k") and not r[2].rstrip().endswith("true>")]
for nm, rs in (("level 0 lock-step products", l0), ("level 1 lock-step products", l1)):
    if rs:
        d = sorted(r[1] - r[0] for r in rs)
        print("  %s: %d launches, median %.1f us, mean %.1f us, window %.1f .. %.1f ms" % (nm, len(rs), d[len(d) // 2] / 1e3, sum(d) / len(d) / 1e3,
              (min(r[0] for r in rs) - a) / 1e6, (max(r[1] for r in rs) - a) / 1e6))

# ---- [r04] per-stream Gantt of the last iteration: consecutive kernels of one family on one stream merged into a segment (gaps < 0.3 ms
# bridged), segments >= 1 ms listed: who runs when, and what each set-up thread is waiting behind
def family2(name):
    f = family(name)
    if f:
        return f
    if name.startswith(("momentum_k", "face_k", "grad_", "pressure_k", "correction_k", "face_coef_k", "diffusion_k")):
        return "assembly"
    if name.startswith(("restrict", "prolong", "interleave", "deinterleave", "vec_", "scale_", "diag_inverse", "fill_k", "count_diff", "nan_to", "reduce_")):
        return "transfers / scalings"
    if name.startswith(("rows_compact", "narrow_build", "xsort")):
        return "Galerkin + mirrors"
    return "other"

print("per-stream segments of the last iteration (ms from its start): stream: [start-end family (kernel ms, launches)]")
by_stream = collections.defaultdict(list)
for r in it:
    by_stream[r[3]].append(r)
for sid, rs in sorted(by_stream.items(), key=lambda kv: min(r[0] for r in kv[1])):
    segs = []
    for r in sorted(rs):
        f = family2(r[2])
        if segs and segs[-1][2] == f and r[0] - segs[-1][1] < 300000:
            segs[-1][1] = max(segs[-1][1], r[1]); segs[-1][3] += r[1] - r[0]; segs[-1][4] += 1
        else:
            segs.append([r[0], r[1], f, r[1] - r[0], 1])
    txt = ["%.0f-%.0f %s (%.0f ms, %d)" % ((s0 - a) / 1e6, (s1 - a) / 1e6, f, d / 1e6, c) for s0, s1, f, d, c in segs if s1 - s0 >= 1000000]
    print("  stream %s: %s" % (sid, "; ".join(txt)))
