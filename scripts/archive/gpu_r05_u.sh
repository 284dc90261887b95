#!/bin/bash
# r05 u: config 5's level-0 products with the non-temporal hint on their matrix streams (their pattern has no narrow column image: ORC_TRACE counts
# the slices that are too wide for one), forced off / default in alternating order
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_u
O=gpurun_out/r05_u
for pass in d:1 0:2 0:3 d:4; do
  m=${pass%%:*}; i=${pass##*:}
  if [ $m = 0 ]; then export ORC_SPMV_NT=0; else unset ORC_SPMV_NT; fi
  ORC_TRACE=$([ $i = 1 ] && echo 1 || echo 0) timeout -k 10 200 python3 bench.py --workload config5 --steps 4 --warmup 1 --no-cpu-baseline --levels-csv $O/c5_levels_nt${m}_$i.csv > $O/c5_bench_nt${m}_$i.json 2> $O/c5_bench_nt${m}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/c5_bench_nt${m}_$i.json')); print('config5 nt=$m ms_per_step %.1f' % d['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],4))"; cut -d, -f1,2,3,5,6,9 $O/c5_levels_nt${m}_$i.csv
done
grep "orc sell" $O/c5_bench_ntd_1.err | head -5
