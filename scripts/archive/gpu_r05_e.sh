#!/bin/bash
# r05 e: where the set-up's time goes now — kernel statistics of ONE hierarchy (scripts/profile_products.py builds a_u's) and of the one-stream bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_e
O=gpurun_out/r05_e
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/kern --output-format csv -- python3 scripts/profile_products.py --reps 3 > $O/kern.log 2>&1 || { tail -3 $O/kern.log; exit 1; }
cp $O/kern/*/*kernel_stats.csv $O/products_kernel_stats.csv; cp $O/kern/*/*kernel_trace.csv $O/products_kernel_trace.csv; rm -rf $O/kern
ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/seq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/seq.log 2>&1
cp $O/seq/*/*kernel_stats.csv $O/bench_sequential_3steps_kernel_stats.csv; rm -rf $O/seq
python3 - <<'PY'
import csv
for f in ("gpurun_out/r05_e/products_kernel_stats.csv", "gpurun_out/r05_e/bench_sequential_3steps_kernel_stats.csv"):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f, "total %.3f s" % (tot/1e9))
    for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:28]:
        print("%6.2f%% %7d calls %9.1f us  %s"%(100*float(r['TotalDurationNs'])/tot,int(r['Calls']),float(r['AverageNs'])/1e3,r['Name'][:100]))
PY
