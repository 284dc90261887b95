#!/bin/bash
# r04 evidence run (part 5, final tree): full GPU suite, smoke, counters + kernel stats of the products and of one hierarchy's set-up, the
# bench line with the per-level table, kernel stats of bench.py in both schedules
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_final_e
O=gpurun_out/r04_final_e
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout=900 -x > $O/suite.log 2>&1
rc=$?; tail -3 $O/suite.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/suite.log | head -20; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log || exit 1
bash scripts/gpu_pmc_r04.sh 6 > $O/pmc.log 2>&1; tail -2 $O/pmc.log
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --levels-csv $O/levels.csv > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-330 $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/conc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/conc.log 2>&1
cp $O/conc/*/*kernel_stats.csv $O/bench_multigrid_concurrent_3steps_kernel_stats.csv; rm -rf $O/conc; echo "concurrent profile done"
ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/seq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/seq.log 2>&1
cp $O/seq/*/*kernel_stats.csv $O/bench_multigrid_sequential_3steps_kernel_stats.csv; rm -rf $O/seq; echo "sequential profile done"
