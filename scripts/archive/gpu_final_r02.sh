#!/bin/bash
# Run on the GPU box at the end of a round: the numbers and profiles DESIGN.md / README.md quote.
#   gpurun_out/final_bench3.log, final_bench20.log   bench.py (concurrent schedule), 3 and 20 steps
#   gpurun_out/final_levels.csv                      per-level product table
#   gpurun_out/prof_finalseq_kernel_stats.csv        rocprofv3 --kernel-trace --stats, everything on one stream
#   gpurun_out/prof_finalconc_kernel_stats.csv       the same in the concurrent (production) schedule
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python3 bench.py --steps 3 --warmup 1 --levels-csv gpurun_out/final_levels.csv > gpurun_out/final_bench3.log 2>&1 || exit 1
timeout -k 10 500 python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/final_bench20.log 2>&1 || exit 1
bash scripts/gpu_profile_seq.sh finalseq -- --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final_seq.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_finalconc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_finalconc.log 2>&1 || exit 1
f=$(ls gpurun_out/prof_finalconc/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/prof_finalconc_kernel_stats.csv
rm -rf gpurun_out/prof_finalconc
grep -h "^{" gpurun_out/final_bench3.log gpurun_out/final_bench20.log | cut -c1-200
