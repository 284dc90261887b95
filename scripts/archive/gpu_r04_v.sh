#!/bin/bash
# r04 call v: product kernels with the slice index in scalar registers: exactness tests of the products, then the bench line (per-level table)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_v
O=gpurun_out/r04_v
timeout -k 10 600 python -m pytest tests/test_gpu_linear_algebra.py tests/test_gpu_triple.py tests/test_gpu_full_size.py tests/test_gpu_reference_order.py -q -x --timeout=500 > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head -20; exit $rc; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || exit 1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_v/bench.json'))
r=d['roofline']
print('ms_per_step', d['ms_per_step'], d['step_ms'])
print('level-0 one system', r['avg_launch_ms'], 'three systems', r['three_systems_per_launch']['avg_launch_ms'])
print('levels', [(l['level'], round(l['us_per_product'],1), round(l['frac_of_peak'],3)) for l in d['amg_levels']])
PY
