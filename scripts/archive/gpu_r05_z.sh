#!/bin/bash
# r05 z: window products with ONE workgroup per 256-row block (dispatched by the hardware) and their sums folded inside the launch, against 2 048 persistent
# workgroups: parity subset, then the bench in alternating order
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_z
O=gpurun_out/r05_z
timeout -k 10 600 python3 -m pytest tests/test_gpu_linear_algebra.py tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_triple.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py tests/test_gpu_full_size.py tests/test_gpu_grid_switches.py -m gpu -x -q --durations=5 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log
[ $rc -eq 0 ] || exit 1
for pass in 1:1 0:2 0:3 1:4; do
  v=${pass%%:*}; i=${pass##*:}
  ORC_XWIN_WG_PER_BLOCK=$v timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_wgpb${v}_$i.json 2> $O/bench_wgpb${v}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/bench_wgpb${v}_$i.json')); print('one wg per block=$v ms_per_step %.1f' % d['ms_per_step'], [round(x) for x in d['step_ms']], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
for pass in 1:1 0:2 0:3 1:4; do
  v=${pass%%:*}; i=${pass##*:}
  ORC_XWIN_WG_PER_BLOCK=$v timeout -k 10 200 python3 bench.py --workload config5 --steps 4 --warmup 1 --no-cpu-baseline > $O/c5_bench_wgpb${v}_$i.json 2> $O/c5_bench_wgpb${v}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/c5_bench_wgpb${v}_$i.json')); print('config5 one wg per block=$v ms_per_step %.1f' % d['ms_per_step'], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
