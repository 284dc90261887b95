#!/bin/bash
# r05 z6: the in-launch fold without a returning ticket per workgroup: the last-indexed workgroup waits (bounded) for the count and folds — parity subset with it on, then the bench alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_z6
O=gpurun_out/r05_z6
ORC_XWIN_FINISHER=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_triple.py tests/test_gpu_bench_family.py tests/test_gpu_full_size.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
i=0
for v in 1 0 0 1; do
  i=$((i+1))
  ORC_XWIN_FINISHER=$v timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$i.json 2> $O/bench_${v}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/bench_${v}_$i.json')); print('finisher=$v ms_per_step %.1f status %s' % (d['ms_per_step'], d['status']), [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
