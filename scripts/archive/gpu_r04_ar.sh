#!/bin/bash
# r04 call ar: preference lists in the sweep and the lock-step evaluations (ORC_AMG_PREFS): pairing tests with the lists, then the iteration with and without
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ar
O=gpurun_out/r04_ar
timeout -k 10 800 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_triple.py tests/test_gpu_reference_order.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py tests/test_gpu_poly_mesh.py -q -x --timeout=700 > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head -20; exit $rc; fi
for round in 1 2; do for v in 1 0; do
  ORC_AMG_PREFS=$v timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_${v}_$round.json'));print('prefs', $v, round(d['ms_per_step'],1), d['step_ms'], d['status'])"
done; done
