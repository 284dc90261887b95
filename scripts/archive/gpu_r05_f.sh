#!/bin/bash
# r05 f: deferred acceptance, fourth form (lists from the first pass, two-trip chain steps): tests, A/B against r04's machinery (new old old new), kernel times of one hierarchy
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_f
O=gpurun_out/r05_f
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_mixed_mesh.py tests/test_gpu_poly_mesh.py tests/test_gpu_bench_family.py tests/test_gpu_triple.py -q -m gpu -x --durations=5 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -9 $O/tests.log
[ $rc = 0 ] || exit 1
r=0; for v in 1 0 0 1; do r=$((r+1))
  ORC_AMG_DA=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_da${v}_$r.json 2> $O/bench_da${v}_$r.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_da${v}_$r.json')); print('da=$v ms_per_step %.1f' % d['ms_per_step'], [round(x) for x in d['step_ms']])"
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kern --output-format csv -- python3 scripts/profile_products.py --reps 3 > $O/kern.log 2>&1 || { tail -3 $O/kern.log; exit 1; }
cp $O/kern/*/*kernel_stats.csv $O/products_kernel_stats.csv; cp $O/kern/*/*kernel_trace.csv $O/products_kernel_trace.csv; rm -rf $O/kern
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r05_f/products_kernel_trace.csv")))
for r in sorted(rows, key=lambda r:int(r["Start_Timestamp"])):
    n=r["Kernel_Name"]
    if "da_" in n or "agg_" in n: print("%-28s %9.1f us" % (n.split("(")[0].replace("void ","").replace("orc::","")[:28], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
ORC_AMG_TRACE=1 timeout -k 10 200 python bench.py --steps 1 --warmup 0 --spin-up 1 --no-cpu-baseline --spmv-reps 2 > $O/trace.json 2> $O/trace.err
grep -h "amg da" $O/trace.err | sort | uniq -c | sort -rn | head -12
