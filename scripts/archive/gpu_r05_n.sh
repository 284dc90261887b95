#!/bin/bash
# r05 n: a level's two smoothing solves share one inverse diagonal and one set of scaled values (ScaledOperator): parity subset, then the
# bench with the sharing on / off in alternating order on one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_n
O=gpurun_out/r05_n
timeout -k 10 400 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_bench_family.py tests/test_gpu_triple.py -m gpu -x -q --durations=5 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log
[ $rc -eq 0 ] || exit 1
for pass in 1:1 0:2 0:3 1:4; do
  sh=${pass%%:*}; i=${pass##*:}
  ORC_AMG_SHARED_SCALING=$sh timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_shared${sh}_$i.json 2> $O/bench_shared${sh}_$i.err || exit 1
  python3 - $O/bench_shared${sh}_$i.json $sh <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("shared=%s ms_per_step %.1f hbm %.1f GB" % (sys.argv[2], d["ms_per_step"], d.get("hbm_peak_gb", d.get("config", {}).get("hbm_peak_gb", 0)) or 0))
PY
done
