#!/bin/bash
# r05 q: one SIMPLE iteration on ONE stream (the one-stream schedule) under the kernel trace: idle time between kernels, and every kernel's time by launch shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_q
O=gpurun_out/r05_q
ORC_BENCH_SKIP_ROOFLINE=1 ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace -d $O/seq --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > $O/seq.log 2>&1 || { tail -5 $O/seq.log; exit 1; }
T=$(ls $O/seq/*/*kernel_trace.csv | head -1)
python3 scripts/analysis/kernel_gaps.py $T --cut 200 --top 12 > $O/gaps_sequential.txt
python3 scripts/analysis/kernel_by_grid.py $T --top 70 > $O/by_grid_sequential.txt; cat $O/by_grid_sequential.txt
rm -rf $O/seq
