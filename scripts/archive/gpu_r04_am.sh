#!/bin/bash
# r04 call am: experiment — followers start their coarse-level aggregations from the leader's pairing of that level (ORC_AMG_SIBLING_COARSE=1)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_am
O=gpurun_out/r04_am
ORC_AMG_SIBLING_COARSE=1 timeout -k 10 300 python -m pytest tests/test_gpu_triple.py tests/test_gpu_bench_family.py -q -x --timeout=250 > $O/tests.log 2>&1
rc=$?; tail -2 $O/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head; exit $rc; fi
for round in 1 2; do for v in 0 1; do
  ORC_AMG_SIBLING_COARSE=$v timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_${v}_$round.json'));print('coarse warm', $v, round(d['ms_per_step'],1), d['step_ms'], d['status'])"
done; done
ORC_AMG_SIBLING_COARSE=1 ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/trace.json 2> $O/trace.err
grep "amg chase" $O/trace.err | tail -12 | cut -c1-160
