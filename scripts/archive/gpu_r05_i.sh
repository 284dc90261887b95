#!/bin/bash
# r05 i: per-level LDS window share of the window product (XWinDev::cap): tests, the chosen sizes, per-level product times and A/B (ORC_XWIN_LEVEL_CAP=0: 40 KB as before)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_i
O=gpurun_out/r05_i
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_grid_switches.py tests/test_gpu_bench_family.py tests/test_gpu_linear_algebra.py -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc = 0 ] || exit 1
ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --spin-up 1 --no-cpu-baseline --spmv-reps 2 > $O/trace.json 2> $O/trace.err
grep -h "amg windows" $O/trace.err | sort | uniq -c | sort -rn | head -8
r=0; for v in 1 0 0 1; do r=$((r+1))
  ORC_XWIN_LEVEL_CAP=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_cap${v}_$r.json 2> $O/bench_cap${v}_$r.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_cap${v}_$r.json')); print('level cap=$v ms_per_step %.1f' % d['ms_per_step'], 'levels us', [round(l['us_per_product'],1) for l in d['amg_levels']], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
