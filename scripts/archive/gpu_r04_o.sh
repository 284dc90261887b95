#!/bin/bash
# r04 call o: the shared Galerkin pass (one symbolic pass, three value sets for the u / v / w level-0 -> 1 products): triple tests, the
# reference-order tests that pin the Multigrid arm against the oracle, then the bench with the pass on and off
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_triple.py tests/test_gpu_reference_order.py tests/test_gpu_bench_family.py tests/test_gpu_multigrid.py -q -x --timeout=500 > gpurun_out/r04o_tests.log 2>&1
rc=$?
tail -4 gpurun_out/r04o_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/r04o_tests.log | head -20; exit $rc; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r04o_bench_on.json 2> gpurun_out/r04o_bench_on.err && python -c "import json;d=json.load(open('gpurun_out/r04o_bench_on.json'));print('shared pass on ', d['ms_per_step'], d['step_ms'], d['config'].get('hbm_used_gb'))" &&
ORC_AMG_SHARED_GALERKIN=0 timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r04o_bench_off.json 2> gpurun_out/r04o_bench_off.err && python -c "import json;d=json.load(open('gpurun_out/r04o_bench_off.json'));print('shared pass off', d['ms_per_step'], d['step_ms'], d['config'].get('hbm_used_gb'))"
