#!/bin/bash
# r05 m (evidence 3): the final tree at the benchmark's own sizes in the reference's own mode against the committed oracle trajectories (bit for bit), the bench
# line (10 steps, CPU baseline), kernel statistics of bench.py in the concurrent and the one-stream schedule
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_m
O=gpurun_out/r05_m
timeout -k 10 300 python3 scripts/reference_mode_fullsize.py --out $O/reference_mode_400x160x160.json > $O/reference_mode_hex.log 2>&1; echo "reference mode hex rc=$?"; tail -3 $O/reference_mode_hex.log
timeout -k 10 300 python3 scripts/reference_mode_fullsize.py --workload config5 --nx 252 --ny 100 --nz 72 --oracle profiles/r04_oracle_trajectory_config5_252x100x72_inplace.json --out $O/reference_mode_config5_252x100x72.json > $O/reference_mode_config5.log 2>&1; echo "reference mode config5 rc=$?"; tail -3 $O/reference_mode_config5.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --levels-csv $O/levels.csv > $O/bench_10steps.json 2> $O/bench_10steps.err; echo "bench rc=$?"; cut -c1-400 $O/bench_10steps.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/conc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/conc.log 2>&1
cp $O/conc/*/*kernel_stats.csv $O/bench_multigrid_concurrent_3steps_kernel_stats.csv; rm -rf $O/conc; echo "concurrent profile done"
ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/seq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/seq.log 2>&1
cp $O/seq/*/*kernel_stats.csv $O/bench_multigrid_sequential_3steps_kernel_stats.csv; rm -rf $O/seq; echo "sequential profile done"
