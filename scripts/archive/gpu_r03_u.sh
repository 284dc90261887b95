#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_mixed_mesh.py tests/test_gpu_reference_order.py tests/test_gpu_poly_mesh.py tests/test_gpu_triple.py -q -m gpu -x > gpurun_out/u_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/u_tests.log
[ $rc -ne 0 ] && exit 1
OLD="ORC_AMG_EVAL_GROUP=16 ORC_AMG_SWEEP_GROUP=16 ORC_AMG_CHASE_GROUP=-1 ORC_GALERKIN_WAVES=16"
bash scripts/gpu_variants.sh "$OLD" "ORC_NOP=1" "$OLD" "ORC_NOP=1" "$OLD" "ORC_NOP=1"
