#!/bin/bash
# r05 z5: window products with about 12 000 entries per workgroup (one to four blocks) on every level with a mirror: parity subset, then the hex channel and config 5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_z5
O=gpurun_out/r05_z5
timeout -k 10 600 python3 -m pytest tests/test_gpu_linear_algebra.py tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_triple.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py tests/test_gpu_full_size.py tests/test_gpu_grid_switches.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
for i in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/bench_$i.json')); print('hex ms_per_step %.1f' % d['ms_per_step'], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
  timeout -k 10 200 python3 bench.py --workload config5 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_bench_$i.json 2> $O/c5_bench_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/c5_bench_$i.json')); print('config5 ms_per_step %.1f' % d['ms_per_step'], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
