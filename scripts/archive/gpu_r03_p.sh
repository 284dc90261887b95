#!/bin/bash
# where the Galerkin merge spends its time: the kernel cut short after each step (ORC_GALERKIN_STAGES=1), one stream
bash scripts/gpu_profile_seq.sh gals ORC_GALERKIN_STAGES=1 ORC_GALERKIN_BATCHED=1 -- --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/gals.txt 2>&1
grep ms_per_step gpurun_out/gals.txt
python3 - <<PY
import csv
rows = [r for r in csv.DictReader(open("gpurun_out/prof_gals_kernel_stats.csv")) if "galerkin_merge" in r["Name"]]
for r in sorted(rows, key=lambda r: r["Name"]):
    print("   %-50s calls %5s avg %9.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
