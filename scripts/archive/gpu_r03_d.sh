#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python scripts/bench_config5.py --csv gpurun_out/r03_config5_5M.csv > gpurun_out/r03d_config5.log 2>&1
echo "config5 rc=$?"; tail -8 gpurun_out/r03d_config5.log | cut -c1-600
timeout -k 10 900 python -m pytest tests/test_gpu_config5.py tests/test_gpu_bench_family.py tests/test_gpu_triple.py -x -q -m gpu --durations=8 > gpurun_out/r03d_tests.log 2>&1
echo "tests rc=$?"; tail -20 gpurun_out/r03d_tests.log
timeout -k 10 1000 python -m pytest "tests/test_gpu_partition.py::test_two_ranks_converged_default_stack_matches_the_oracle" -x -q -m gpu -s > gpurun_out/r03d_converged.log 2>&1
echo "converged rc=$?"; tail -12 gpurun_out/r03d_converged.log
