#!/bin/bash
# r04 call aw: with preference lists on the fine level too: is its row mirror still worth its 2.8 GB? (ORC_AMG_L0_MIRROR)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_aw
O=gpurun_out/r04_aw
for round in 1 2 3; do for v in 1 0; do
  ORC_AMG_L0_MIRROR=$v timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_${v}_$round.json'));print('L0 mirror', $v, round(d['ms_per_step'],1), d['step_ms'], d['status'])"
done; done
