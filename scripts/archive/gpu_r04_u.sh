#!/bin/bash
# r04 call u: cascades that foresee their successor: pairing tests, then bench + one-stream trace with the prefetch on and off
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_u
O=gpurun_out/r04_u
timeout -k 10 700 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_triple.py tests/test_gpu_reference_order.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py -q -x --timeout=600 > $O/tests.log 2>&1
rc=$?
tail -3 $O/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head -20; exit $rc; fi
for pf in 1 0; do
  ORC_AMG_CHASE_PREFETCH=$pf timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$pf.json 2> $O/bench_$pf.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_$pf.json'));print('bench prefetch', $pf, d['ms_per_step'], d['step_ms'])"
  ORC_AMG_CHASE_PREFETCH=$pf ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/trace_$pf.json 2> $O/trace_$pf.err || exit 1
  python scripts/amg_phases.py $O/trace_$pf.err | grep -E "^all +(casc|all)|^(10240000|5120000|2560000) +casc|evaluations"
done
