#!/bin/bash
# r03 first GPU call: the new parity tests, then the config-3 bench line with its kernel-trace profile.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bench_family.py tests/test_gpu_poly_mesh.py "tests/test_gpu_partition.py::test_two_ranks_converged_default_stack_matches_the_oracle" -x -q -m gpu -s > gpurun_out/r03a_tests.log 2>&1
echo "tests rc=$?"; tail -25 gpurun_out/r03a_tests.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_config3 --output-format csv -- python bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 5 --warmup 1 > gpurun_out/r03a_config3.log 2>&1
echo "config3 rc=$?"; tail -1 gpurun_out/r03a_config3.log | cut -c1-1500
f=$(ls gpurun_out/prof_config3/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -20 "$f" | cut -c1-200; fi
