#!/bin/bash
# r04 call s: cascades with static first shares: pairing tests, bench, one-stream kernel stats
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_s
O=gpurun_out/r04_s
timeout -k 10 700 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_triple.py tests/test_gpu_reference_order.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py -q -x --timeout=600 > $O/tests.log 2>&1
rc=$?
tail -3 $O/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head -20; exit $rc; fi
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err && python -c "import json;d=json.load(open('$O/bench.json'));print('bench', d['ms_per_step'], d['step_ms'])" &&
ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/seq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/seq.log 2>&1
cp $O/seq/*/*kernel_stats.csv $O/seq_kernel_stats.csv; rm -rf $O/seq; grep "tail_chase\|galerkin_bound\|chase_carry" $O/seq_kernel_stats.csv | cut -d, -f1-4 | cut -c1-60,150-
