#!/bin/bash
# r05 h: the re-applied chain kernel: tests, the longest chain per level (ORC_AMG_TRACE), A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_h
O=gpurun_out/r05_h
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_mixed_mesh.py tests/test_gpu_poly_mesh.py tests/test_gpu_bench_family.py tests/test_gpu_triple.py -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc = 0 ] || exit 1
ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --spin-up 1 --no-cpu-baseline --spmv-reps 2 > $O/trace.json 2> $O/trace.err
grep -h "amg da" $O/trace.err | sort | uniq -c | sort -rn | head -12
r=0; for v in 1 0 0 1; do r=$((r+1))
  ORC_AMG_DA=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_da${v}_$r.json 2> $O/bench_da${v}_$r.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_da${v}_$r.json')); print('da=$v ms_per_step %.1f' % d['ms_per_step'], [round(x) for x in d['step_ms']])"
done
