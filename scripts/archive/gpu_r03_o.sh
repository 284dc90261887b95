#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_mixed_mesh.py tests/test_gpu_reference_order.py tests/test_gpu_poly_mesh.py -q -m gpu -x > gpurun_out/o_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/o_tests.log
[ $rc -ne 0 ] && exit 1
for v in 0 1; do
  bash scripts/gpu_profile_seq.sh galb_$v ORC_GALERKIN_BATCHED=$v -- --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/galb_$v.txt 2>&1
  echo "== batched=$v"; grep ms_per_step gpurun_out/galb_$v.txt
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_galb_${v}_kernel_stats.csv")):
    if "galerkin_merge" in r["Name"]:
        print("   %-40s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
