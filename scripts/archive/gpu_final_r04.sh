#!/bin/bash
# r04 evidence run on the GPU box (part 1): counters + kernel stats of the products and of one hierarchy's set-up, kernel stats of bench.py in the
# concurrent and the one-stream schedule, the bench line with the per-level table, BASELINE configs[2] (bench + kernel stats).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_final
O=gpurun_out/r04_final
bash scripts/gpu_pmc_r04.sh 6 > $O/pmc.log 2>&1; tail -3 $O/pmc.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/conc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/conc.log 2>&1
cp $O/conc/*/*kernel_stats.csv $O/bench_multigrid_concurrent_3steps_kernel_stats.csv; rm -rf $O/conc; echo "concurrent profile rc=$?"
ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/seq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/seq.log 2>&1
cp $O/seq/*/*kernel_stats.csv $O/bench_multigrid_sequential_3steps_kernel_stats.csv; rm -rf $O/seq; echo "sequential profile rc=$?"
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --levels-csv $O/levels.csv > $O/bench.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/c3 --output-format csv -- python3 bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 > $O/config3.log 2>&1
cp $O/c3/*/*kernel_stats.csv $O/config3_kernel_stats.csv; rm -rf $O/c3; echo "config3 rc=$?"; grep '^{"metric' $O/config3.log | cut -c1-300
