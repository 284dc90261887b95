#!/bin/bash
# r04 call p: certification by one verification pass: the multigrid / pairing tests, then the bench with the pass and with r03's round
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_triple.py tests/test_gpu_reference_order.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py -q -x --timeout=600 > gpurun_out/r04p_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r04p_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/r04p_tests.log | head -20; exit $rc; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r04p_bench_on.json 2> gpurun_out/r04p_bench_on.err && python -c "import json;d=json.load(open('gpurun_out/r04p_bench_on.json'));print('verification pass  ', d['ms_per_step'], d['step_ms'])" &&
ORC_AMG_CERTIFY_ROUND=0 timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r04p_bench_off.json 2> gpurun_out/r04p_bench_off.err && python -c "import json;d=json.load(open('gpurun_out/r04p_bench_off.json'));print('second run         ', d['ms_per_step'], d['step_ms'])"
