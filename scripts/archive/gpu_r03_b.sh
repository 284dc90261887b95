#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_triple.py -x -q -m gpu > gpurun_out/r03b_triple.log 2>&1
echo "triple rc=$?"; tail -15 gpurun_out/r03b_triple.log
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_poly_mesh.py -x -q -m gpu > gpurun_out/r03b_mg.log 2>&1
echo "mg rc=$?"; tail -8 gpurun_out/r03b_mg.log
timeout -k 10 300 python scripts/probe_bench_family.py > gpurun_out/r03b_probe.log 2>&1
echo "probe rc=$?"; cat gpurun_out/r03b_probe.log | tail -30
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03b_bench_triple.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/r03b_bench_triple.log | cut -c1-700
ORC_TRIPLE_MOMENTUM=0 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03b_bench_lanes.log 2>&1
echo "bench(lanes) rc=$?"; tail -1 gpurun_out/r03b_bench_lanes.log | cut -c1-700
