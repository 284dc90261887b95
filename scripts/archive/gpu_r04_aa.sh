#!/bin/bash
# r04 call aa: the cascade launch sized to what is resident (ORC_AMG_CHASE_GRID; 68 VGPRs = 7 wavefronts per SIMD = 1792 workgroups on an empty chip)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_aa
O=gpurun_out/r04_aa
for round in 1 2; do for g in 2048 1792 1024 512; do
  ORC_AMG_CHASE_GRID=$g timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${g}_$round.json 2> $O/bench_${g}_$round.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_${g}_$round.json'));print('chase grid', $g, round(d['ms_per_step'],1), d['step_ms'])"
done; done
