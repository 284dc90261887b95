#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_linear_algebra.py tests/test_gpu_multigrid.py tests/test_gpu_reference_order.py tests/test_gpu_triple.py tests/test_gpu_assembly.py tests/test_gpu_full_size.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
bash scripts/gpu_variants.sh "A=1" "ORC_SPMV_NARROW_COLS=0" "A=2" "ORC_SPMV_NARROW_COLS=0"
