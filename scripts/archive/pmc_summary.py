#!/usr/bin/env python3
"""Folds the rocprofv3 counter passes of scripts/gpu_pmc.sh (gpurun_out/r01/pmc_fetch, pmc_write, kern) into
profiles/<round>_pmc_fetch_write_per_kernel.csv and profiles/<round>_spmv_pmc.json.  FETCH_SIZE is reported in KB of
32-byte-sector pairs on gfx950: x2 per MI355X_MICROARCH.md, calibrated here on the streaming BiCGSTAB vector kernels,
whose read traffic is known exactly (bicg_s_k reads 2 vectors, bicg_p_k 3, bicg_xr_k 4)."""
import collections
import csv
import glob
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
base = os.path.join(root, "gpurun_out", "r01")
N, NNZ = 10_240_000, 71_372_800


def counters(sub, name):
    f = sorted(glob.glob(os.path.join(base, sub, "*", "*counter_collection.csv")))[-1]
    acc = collections.defaultdict(list)
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != name:
            continue
        per[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])
    for (_, k), v in per.items():
        acc[k.split("(")[0]].append(v)
    return acc


fetch, write = counters("pmc_fetch", "FETCH_SIZE"), counters("pmc_write", "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    fa = sum(fetch.get(k, [0])) / max(len(fetch.get(k, [0])), 1)
    wa = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
    rows.append((k, len(fetch.get(k, write.get(k, []))), fa, wa, int(fa * 2 * 1024), int(wa * 1024)))
out_csv = os.path.join(root, "profiles", tag + "_pmc_fetch_write_per_kernel.csv")
with open(out_csv, "w") as fh:
    fh.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_read_bytes_corrected(x2x1024),hbm_write_bytes(x1024)\n")
    for r in rows:
        fh.write('"%s",%d,%.1f,%.1f,%d,%d\n' % r)
avg = {r[0]: r for r in rows}
calib = {}
for k, vecs in (("orc::bicg_s_k", 2), ("orc::bicg_p_k", 3), ("orc::bicg_xr_k", 4)):
    if k in avg and avg[k][2] > 0:
        calib[k] = vecs * 8.0 * N / (avg[k][2] * 1024.0)
spmv = next(k for k in avg if k.startswith("void orc::spmv_k<orc::EpiStore"))
doc = {
    "workload": "hex channel 400x160x160", "n": N, "nnz": NNZ, "kernel": "spmv_k<EpiStore>",
    "FETCH_SIZE_KB": avg[spmv][2], "WRITE_SIZE_KB": avg[spmv][3], "fetch_correction": 2.0, "fetch_correction_calibration": calib,
    "hbm_bytes_per_launch": avg[spmv][2] * 2048.0 + avg[spmv][3] * 1024.0,
    "algorithmic_bytes_per_launch": 12.0 * NNZ + 20.0 * N,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 scripts/profile_kernels.py ; FETCH_SIZE x2 per "
              "MI355X_MICROARCH.md (calibrated on the streaming kernels bicg_s_k/p_k/xr_k of the same run); scripts/pmc_summary.py",
}
json.dump(doc, open(os.path.join(root, "profiles", tag + "_spmv_pmc.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
