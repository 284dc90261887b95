#!/bin/bash
# r04 call t: adjacent slices per sweep group (ORC_AMG_SWEEP_BLOCK): what the cascades are left with, one-stream trace + default bench per setting
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_t
O=gpurun_out/r04_t
for b in 0 3 8 16 32; do
  ORC_AMG_SWEEP_BLOCK=$b ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/trace_$b.json 2> $O/trace_$b.err || exit 1
  echo "== block $b"; python scripts/amg_phases.py $O/trace_$b.err | grep -E "^all|evaluations"
  ORC_AMG_SWEEP_BLOCK=$b timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$b.json 2> $O/bench_$b.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_$b.json'));print('bench block', $b, d['ms_per_step'], d['step_ms'])"
done
