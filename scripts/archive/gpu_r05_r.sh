#!/bin/bash
# r05 r: config 5's level 1 (17 entries per row, 29 % padding in the SELL image) with the packed mirror + LDS windows (ORC_SPMV_XWIN_MIN_NNZ=16) against the
# default (24: SELL image), alternating on one box; the hex channel's level 1 (15 entries, 5.7 % padding) the same way for reference
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_r
O=gpurun_out/r05_r
for pass in 16:1 24:2 24:3 16:4; do
  m=${pass%%:*}; i=${pass##*:}
  ORC_SPMV_XWIN_MIN_NNZ=$m timeout -k 10 200 python3 bench.py --workload config5 --steps 4 --warmup 1 --no-cpu-baseline --levels-csv $O/c5_levels_min${m}_$i.csv > $O/c5_bench_min${m}_$i.json 2> $O/c5_bench_min${m}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/c5_bench_min${m}_$i.json')); print('config5 min_nnz=$m ms_per_step %.1f' % d['ms_per_step'])"; cut -d, -f1,2,3,5,6,9 $O/c5_levels_min${m}_$i.csv
done
for pass in 14:1 24:2 24:3 14:4; do
  m=${pass%%:*}; i=${pass##*:}
  ORC_SPMV_XWIN_MIN_NNZ=$m timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --levels-csv $O/levels_min${m}_$i.csv > $O/bench_min${m}_$i.json 2> $O/bench_min${m}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/bench_min${m}_$i.json')); print('hex min_nnz=$m ms_per_step %.1f' % d['ms_per_step'])"; cut -d, -f1,2,3,5,6,9 $O/levels_min${m}_$i.csv
done
