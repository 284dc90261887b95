#!/bin/bash
# r05 j: the window product's stream in chunks of 4 entries per lane (58 VGPRs: more resident wavefronts where the level's LDS share allows) against 8 (92-94 VGPRs)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_j
O=gpurun_out/r05_j
ORC_XWIN_CHUNK=4 timeout -k 10 300 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "tests (chunk 4) rc=$rc"; tail -2 $O/tests.log
[ $rc = 0 ] || exit 1
r=0; for v in 4 8 8 4; do r=$((r+1))
  ORC_XWIN_CHUNK=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_chunk${v}_$r.json 2> $O/bench_chunk${v}_$r.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_chunk${v}_$r.json')); print('chunk=$v ms_per_step %.1f' % d['ms_per_step'], 'levels us', [round(l['us_per_product'],1) for l in d['amg_levels']], [round(l['frac_of_peak'],3) for l in d['amg_levels']])"
done
