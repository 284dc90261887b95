#!/bin/bash
# r04 evidence run (part 7, final tree): kernel stats of bench.py in both schedules, counters + kernel stats of the products and of one hierarchy's set-up
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_final_g
O=gpurun_out/r04_final_g
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/conc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/conc.log 2>&1
cp $O/conc/*/*kernel_stats.csv $O/bench_multigrid_concurrent_3steps_kernel_stats.csv; rm -rf $O/conc; echo "concurrent profile done"
ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/seq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/seq.log 2>&1
cp $O/seq/*/*kernel_stats.csv $O/bench_multigrid_sequential_3steps_kernel_stats.csv; rm -rf $O/seq; echo "sequential profile done"
bash scripts/gpu_pmc_r04.sh 6 > $O/pmc.log 2>&1; tail -2 $O/pmc.log
