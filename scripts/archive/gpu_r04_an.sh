#!/bin/bash
# r04 call an: level 2 (34 entries per row) on the padded image with direct gathers instead of the window product (ORC_SPMV_XWIN_MIN_NNZ=40)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_an
O=gpurun_out/r04_an
for round in 1 2; do for v in 24 40; do
  ORC_SPMV_XWIN_MIN_NNZ=$v timeout -k 10 250 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_${v}_$round.json 2> $O/bench_${v}_$round.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_${v}_$round.json'));print('xwin min nnz', $v, round(d['ms_per_step'],1), 'levels', [round(l['us_per_product'],1) for l in d['amg_levels']], d['config'].get('hbm_used_gb'))"
done; done
