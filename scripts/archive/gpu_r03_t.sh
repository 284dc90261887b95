#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ORC_AMG_CHASE_GROUP=8 timeout -k 10 300 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_mixed_mesh.py -q -m gpu -x > gpurun_out/t_tests.log 2>&1; rc=$?; echo "tests(chase 8) rc=$rc"; tail -2 gpurun_out/t_tests.log; [ $rc -ne 0 ] && exit 1
for v in "ORC_AMG_CHASE_GROUP=8" "ORC_AMG_CHASE_GROUP=16"; do
  tag=$(echo "$v" | tr ' =' '__')
  bash scripts/gpu_profile_seq.sh $tag $v -- --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag.txt 2>&1
  echo "== $v"; grep ms_per_step gpurun_out/$tag.txt
  python3 - <<PY
import csv
tot = 0
for r in csv.DictReader(open("gpurun_out/prof_${tag}_kernel_stats.csv")):
    if "tail_" in r["Name"] or "agg_sweep" in r["Name"] or "chase" in r["Name"]:
        tot += float(r["TotalDurationNs"]) / 1e6
        if float(r["TotalDurationNs"]) > 5e6: print("   %-44s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("   aggregation kernels total %.1f ms" % tot)
PY
done
