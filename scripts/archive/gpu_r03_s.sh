#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "ORC_AMG_EVAL_GROUP=4" "ORC_AMG_EVAL_GROUP=8 ORC_AMG_SWEEP_GROUP=4" "ORC_AMG_EVAL_GROUP=8 ORC_AMG_SWEEP_GROUP=8" "ORC_AMG_EVAL_GROUP=8 ORC_AMG_SWEEP_GROUP=16"; do
  tag=$(echo "$v" | tr ' =' '__')
  bash scripts/gpu_profile_seq.sh $tag $v -- --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag.txt 2>&1
  echo "== $v"; grep ms_per_step gpurun_out/$tag.txt
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_${tag}_kernel_stats.csv")):
    if "tail_eval" in r["Name"] or "agg_sweep" in r["Name"]:
        print("   %-44s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
