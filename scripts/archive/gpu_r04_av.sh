#!/bin/bash
# r04 call av: preference lists from how many entries per row on (ORC_AMG_PREFS_MIN_LEN: 12 = the coarse levels, 5 = the mesh pattern too)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_av
O=gpurun_out/r04_av
for round in 1 2 3; do for v in 12 5; do
  ORC_AMG_PREFS_MIN_LEN=$v timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_${v}_$round.json'));print('prefs from', $v, round(d['ms_per_step'],1), d['step_ms'], d['status'])"
done; done
