#!/bin/bash
# r05 g: lanes per chain (ORC_AMG_DA_GROUP 4 / 8 / 16 / by row length): kernel times of one hierarchy's pairing per level, then tests + A/B of the default
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_g
O=gpurun_out/r05_g
for G in default 4 8; do
  E=""; [ $G != default ] && export ORC_AMG_DA_GROUP=$G
  timeout -k 10 200 rocprofv3 --kernel-trace -d $O/kern$G --output-format csv -- python3 scripts/profile_products.py --reps 2 > $O/kern$G.log 2>&1 || { tail -3 $O/kern$G.log; exit 1; }
  unset ORC_AMG_DA_GROUP
  cp $O/kern$G/*/*kernel_trace.csv $O/trace_$G.csv; rm -rf $O/kern$G
  python3 - $O/trace_$G.csv $G <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
for r in sorted(rows, key=lambda r:int(r["Start_Timestamp"])):
    n=r["Kernel_Name"]
    if "da_first" in n or "da_chase" in n: out.append("%s %.0f" % (n.split("(")[0].replace("void ","").replace("orc::","")[:14], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
print("G =", sys.argv[2], "|", " | ".join(out), "us")
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_mixed_mesh.py tests/test_gpu_poly_mesh.py tests/test_gpu_bench_family.py tests/test_gpu_triple.py -q -m gpu -x > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc = 0 ] || exit 1
r=0; for v in 1 0 0 1; do r=$((r+1))
  ORC_AMG_DA=$v timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_da${v}_$r.json 2> $O/bench_da${v}_$r.err || exit 1
  python -c "import json; d=json.load(open('$O/bench_da${v}_$r.json')); print('da=$v ms_per_step %.1f' % d['ms_per_step'], [round(x) for x in d['step_ms']])"
done
