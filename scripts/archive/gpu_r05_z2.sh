#!/bin/bash
# r05 z2: how many blocks a workgroup of the window products should take (entries per workgroup: 0 = one block from 8 000 entries per block on; 12 000; 18 000; 30 000)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_z2
O=gpurun_out/r05_z2
i=0
for e in 0 12000 18000 30000 30000 18000 12000 0; do
  i=$((i+1))
  ORC_XWIN_WG_ENTRIES=$e timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${e}_$i.json 2> $O/bench_${e}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/bench_${e}_$i.json')); print('hex entries per wg=$e ms_per_step %.1f' % d['ms_per_step'], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
i=0
for e in 0 12000 18000 30000 30000 18000 12000 0; do
  i=$((i+1))
  ORC_XWIN_WG_ENTRIES=$e timeout -k 10 200 python3 bench.py --workload config5 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_bench_${e}_$i.json 2> $O/c5_bench_${e}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/c5_bench_${e}_$i.json')); print('config5 entries per wg=$e ms_per_step %.1f' % d['ms_per_step'], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
