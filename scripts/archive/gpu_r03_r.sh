#!/bin/bash
# lock-step round evaluation (tail_eval_k<G>): lanes per row — kernel averages on one stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_mixed_mesh.py -q -m gpu -x > gpurun_out/r_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r_tests.log
[ $rc -ne 0 ] && exit 1
for v in 16 0 8; do
  bash scripts/gpu_profile_seq.sh ev_$v ORC_AMG_EVAL_GROUP=$v -- --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/ev_$v.txt 2>&1
  echo "== eval group $v (0 = default by row length)"; grep ms_per_step gpurun_out/ev_$v.txt
  python3 - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_ev_${v}_kernel_stats.csv")):
    if "tail_eval" in r["Name"]:
        print("   %-44s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
