#!/bin/bash
# r04 call ab: the cascade grid's new default against 1792 given by hand; the window product's workgroups per CU (ORC_XWIN_WGS_PER_CU, default 8)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ab
O=gpurun_out/r04_ab
run() { # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_$name.json'));print('$name', round(d['ms_per_step'],1), 'levels', [round(l['us_per_product'],1) for l in d['amg_levels']])"
}
for round in 1 2; do
  run default_$round ORC_DUMMY=1
  run grid1792_$round ORC_AMG_CHASE_GRID=1792
  run xwin4_$round ORC_XWIN_WGS_PER_CU=4
  run xwin6_$round ORC_XWIN_WGS_PER_CU=6
done
