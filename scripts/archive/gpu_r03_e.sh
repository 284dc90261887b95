#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( while true; do sleep 60; echo "[alive $(date +%T)]"; done ) &
HB=$!
timeout -k 10 600 python -m pytest "tests/test_gpu_partition.py::test_lane_error_on_one_rank_reaches_every_rank" "tests/test_gpu_partition.py::test_rccl_overlapped_product_on_a_self_loop_communicator" -x -q -m gpu -s > gpurun_out/r03e_part.log 2>&1
echo "partition rc=$?"; tail -25 gpurun_out/r03e_part.log
timeout -k 10 1000 python -m pytest "tests/test_gpu_partition.py::test_two_ranks_converged_default_stack_matches_the_oracle" -x -q -m gpu -s --durations=3 > gpurun_out/r03e_converged.log 2>&1
echo "converged rc=$?"; tail -12 gpurun_out/r03e_converged.log
kill $HB
