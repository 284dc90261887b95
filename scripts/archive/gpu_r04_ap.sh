#!/bin/bash
# r04 call ap: the level-0 row mirror once more, now that the set-up is leaner (ORC_AMG_L0_MIRROR=1: the sweeps and cascades of the fine level walk
# row-contiguous entries: 2 cache lines per row instead of 15)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ap
O=gpurun_out/r04_ap
for round in 1 2; do for v in 0 1; do
  ORC_AMG_L0_MIRROR=$v timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_${v}_$round.json'));print('L0 mirror', $v, round(d['ms_per_step'],1), d['step_ms'], d['config'].get('hbm_used_gb'))"
done; done
