#!/bin/bash
# r04 call l: the level-0 row mirror of the set-up: exactness test, then same-box pairs of the headline bench with and without it
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_window_fallback.py tests/test_gpu_multigrid.py tests/test_gpu_reference_order.py -q --timeout=600 > gpurun_out/r04l_tests.log 2>&1
rc=$?; grep -E "passed|failed|FAILED" gpurun_out/r04l_tests.log | tail -4
if [ $rc -ne 0 ]; then exit 1; fi
run() { tag=$1; shift; env "$@" timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --spmv-reps 3 > gpurun_out/r04l_$tag.json 2> gpurun_out/r04l_$tag.err || { tail -3 gpurun_out/r04l_$tag.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04l_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), d['step_ms'], d['config']['hbm_used_gb'])"; }
run mirror A=1 || exit 1
run nomirror ORC_AMG_L0_MIRROR=0 || exit 1
run mirror2 A=1 || exit 1
run nomirror2 ORC_AMG_L0_MIRROR=0 || exit 1
ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --spmv-reps 2 > gpurun_out/r04l_trace_mirror.json 2> gpurun_out/r04l_trace_mirror.err
python scripts/amg_phases.py gpurun_out/r04l_trace_mirror.err | tail -30
ORC_AMG_L0_MIRROR=0 ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --spmv-reps 2 > gpurun_out/r04l_trace_nomirror.json 2> gpurun_out/r04l_trace_nomirror.err
python scripts/amg_phases.py gpurun_out/r04l_trace_nomirror.err | tail -30
