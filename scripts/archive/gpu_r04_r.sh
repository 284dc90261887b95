#!/bin/bash
# r04 call r: kernel stats of bench.py in the one-stream schedule after the set-up changes of the round's second half
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_r
O=gpurun_out/r04_r
ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/seq --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/seq.log 2>&1
cp $O/seq/*/*kernel_stats.csv $O/bench_multigrid_sequential_3steps_kernel_stats_b.csv; rm -rf $O/seq; echo "sequential profile rc=$?"
grep '^{"metric' $O/seq.log | cut -c1-200
