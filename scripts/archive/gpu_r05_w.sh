#!/bin/bash
# r05 w: the narrow column image packed four depths per lane (one 8-byte load per lane and four depths): parity subset, the products by level, the bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_w
O=gpurun_out/r05_w
timeout -k 10 600 python3 -m pytest tests/test_gpu_linear_algebra.py tests/test_gpu_multigrid.py tests/test_gpu_triple.py tests/test_gpu_bench_family.py tests/test_gpu_full_size.py tests/test_gpu_grid_switches.py -m gpu -x -q --durations=5 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 scripts/profile_products.py --reps 20 > $O/products.log 2>&1; grep "cells\|level" $O/products.log
for i in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --levels-csv $O/levels_$i.csv > $O/bench_$i.json 2> $O/bench_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/bench_$i.json')); print('hex ms_per_step %.1f frac %.4f bicg %.4f' % (d['ms_per_step'], d['roofline']['frac'], d['roofline']['bicgstab_iteration_ms']))"; cut -d, -f1,6,9 $O/levels_$i.csv
done
