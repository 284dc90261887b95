#!/bin/bash
# r05 z4: the window product with four instead of eight entries per chunk (58 VGPRs) now that one workgroup per block comes and goes: bench, alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_z4
O=gpurun_out/r05_z4
i=0
for v in 1 0 0 1; do
  i=$((i+1))
  ORC_XWIN_CHUNK4=$v timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/bench_${v}_$i.json')); print('chunk4=$v ms_per_step %.1f' % d['ms_per_step'], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
