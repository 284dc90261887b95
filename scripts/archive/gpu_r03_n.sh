#!/bin/bash
# galerkin merge: wavefronts per CU (ORC_GALERKIN_WAVES) — kernel averages on one stream
for v in 8 16 24 32; do
  bash scripts/gpu_profile_seq.sh galw_$v ORC_GALERKIN_WAVES=$v -- --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/galw_$v.txt 2>&1
  echo "== waves per CU $v"; grep ms_per_step gpurun_out/galw_$v.txt
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_galw_${v}_kernel_stats.csv")):
    if "galerkin_merge" in r["Name"]:
        print("   %-40s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
