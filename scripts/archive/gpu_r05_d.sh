#!/bin/bash
# r05 d: (1) the two-rank run with r04's THREE classes kept but the library stream moved into the solve class (the rule of runtime.cpp stream_create):
# does it run now?  (2) the pairing by deferred acceptance: the pairing / Galerkin / golden tests, (3) A/B against r04's machinery in one binary
# (ORC_AMG_DA=0), order new old old new, (4) the proposals' statistics per level.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_d
O=gpurun_out/r05_d
python3 -c "import torch" >/dev/null 2>&1
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_mixed_mesh.py tests/test_gpu_poly_mesh.py tests/test_gpu_bench_family.py tests/test_gpu_triple.py -q -m gpu -x --durations=8 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -14 $O/tests.log
[ $rc = 0 ] || exit 1
r=0; for v in 1 0 0 1; do r=$((r+1))
  ORC_AMG_DA=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_da${v}_$r.json 2> $O/bench_da${v}_$r.err || exit 1
  python - "$O/bench_da${v}_$r.json" "da=$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], 'ms_per_step %.1f' % d['ms_per_step'], 'L0 x1 %.1f us' % (1e3*r['avg_launch_ms']), 'levels', [round(l['us_per_product'],1) for l in d['amg_levels']], 'traj', d['report_trajectory']['timed'][3:])
PY
done
for G in 8 4 32; do
  ORC_AMG_DA_GROUP=$G timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_group$G.json 2> $O/bench_group$G.err || exit 1
  python -c "import json,sys; d=json.load(open('$O/bench_group$G.json')); print('da group $G ms_per_step %.1f' % d['ms_per_step'])"
done
ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --spin-up 1 --no-cpu-baseline --spmv-reps 2 > $O/trace.json 2> $O/trace.err
grep -h "amg da" $O/trace.err | sort | uniq -c | sort -rn | head -30
