#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_linear_algebra.py -q -m gpu -x > gpurun_out/k_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/k_tests.log
ORC_XWIN_EARLY=1 timeout -k 10 300 python -m pytest tests/test_gpu_multigrid.py -q -m gpu -x > gpurun_out/k_tests_early.log 2>&1; echo "tests(early) rc=$?"; tail -2 gpurun_out/k_tests_early.log
bash scripts/gpu_variants.sh "ORC_XWIN_EARLY=0" "ORC_XWIN_EARLY=1" "ORC_XWIN_EARLY=0" "ORC_XWIN_EARLY=1"
