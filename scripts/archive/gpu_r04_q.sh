#!/bin/bash
# r04 call q: window-fallback tests after the counter change, then the cascade step budget per launch re-tuned now that a launch no longer
# pays for its shared counters (ORC_AMG_CHASE_STEPS, default 96)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_window_fallback.py -q -x --timeout=250 > gpurun_out/r04q_tests.log 2>&1 || { tail -20 gpurun_out/r04q_tests.log; exit 1; }
tail -2 gpurun_out/r04q_tests.log
for s in 96 48 32 64 160; do
  ORC_AMG_CHASE_STEPS=$s timeout -k 10 200 python bench.py --steps 4 --warmup 1 > gpurun_out/r04q_steps_$s.json 2> gpurun_out/r04q_steps_$s.err || exit 1
  python -c "import json,sys;d=json.load(open('gpurun_out/r04q_steps_$s.json'));print('chase steps', $s, d['ms_per_step'], d['step_ms'])"
done
