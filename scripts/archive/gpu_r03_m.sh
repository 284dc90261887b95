#!/bin/bash
# (record of a measurement: the switches ORC_GALERKIN_ROWS / ORC_GALERKIN_XCD existed in the tree of that run only)
# galerkin merge: fine rows from the SELL image / from the row-contiguous mirror, round-robin / XCD-wise list walk — kernel averages, one stream
for v in "00" "10" "01" "11"; do
  r=${v:0:1}; x=${v:1:1}
  bash scripts/gpu_profile_seq.sh gal_$v ORC_GALERKIN_ROWS=$r ORC_GALERKIN_XCD=$x -- --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/gal_$v.txt 2>&1
  echo "== rows=$r xcd=$x"; grep ms_per_step gpurun_out/gal_$v.txt
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_gal_${v}_kernel_stats.csv")):
    if "galerkin" in r["Name"] or "tail_chase" in r["Name"]:
        print("   %-40s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
