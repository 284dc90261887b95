#!/bin/bash
# r04 call as: preference lists on the levels with long rows only (ORC_AMG_PREFS_MIN_LEN, default 12 entries per row)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_as
O=gpurun_out/r04_as
for round in 1 2; do for v in "1 12" "0 12" "1 20"; do set -- $v
  ORC_AMG_PREFS=$1 ORC_AMG_PREFS_MIN_LEN=$2 timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$1_$2_$round.json 2> $O/bench_$1_$2_$round.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_$1_$2_$round.json'));print('prefs $1 min len $2', round(d['ms_per_step'],1), d['step_ms'], d['status'])"
done; done
