#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/gpu_variants.sh "A=1" "ORC_AMG_CHASE_GRID=1024" "ORC_AMG_CHASE_GRID=512" "ORC_AMG_CHASE_GRID=256" "ORC_AMG_CHASE_STEPS=256" "ORC_GALERKIN_WAVES=8" "ORC_AMG_TAIL_GRID=256" "ORC_AMG_CHASE_GRID=512 ORC_AMG_CHASE_STEPS=256"
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/tl --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/tl.log 2>&1
f=$(ls gpurun_out/tl/*/*kernel_trace.csv | head -1)
python3 scripts/timeline.py "$f"
rm -rf gpurun_out/tl
timeout -k 10 120 python -m pytest "tests/test_gpu_partition.py::test_rccl_overlapped_product_on_a_self_loop_communicator" -x -q -m gpu -s 2>&1 | grep "against the plain\|passed\|failed" | cut -c1-300
