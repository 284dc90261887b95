#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_linear_algebra.py tests/test_gpu_multigrid.py tests/test_gpu_triple.py tests/test_gpu_reference_order.py -q -m gpu -x > gpurun_out/y_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/y_tests.log
[ $rc -ne 0 ] && exit 1
bash scripts/gpu_variants.sh "ORC_NOP=1" "ORC_NOP=2" "ORC_NOP=3"
