#!/usr/bin/env python3
"""Per-level averages of the counters collected by scripts/gpu_pmc_levels.sh (first dispatch of a level = warm-up, dropped)."""
import collections, csv, glob, sys
root = sys.argv[1]
res = collections.OrderedDict()
dur = {}
for f in sorted(glob.glob(root + '/p*/*/*_counter_collection.csv')):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'spmv' in k and 'EpiStore,' in k:
            d = disp.setdefault(int(r['Dispatch_Id']), {'_t': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
            d[r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(disp)
    per = len(ids) // 4
    for lvl in range(4):
        sel = ids[lvl * per + 1:(lvl + 1) * per]
        for cname in disp[sel[0]]:
            vals = [disp[i][cname] for i in sel]
            key = cname if cname != '_t' else 'duration_us(profiled)'
            res.setdefault(key, {})[lvl] = sum(vals) / len(vals)
print("%-36s" % "counter", "  ".join("%14s" % ("level %d" % l) for l in range(4)))
for c, d in res.items():
    print("%-36s" % c, "  ".join("%14.5g" % d[l] for l in range(4)))
