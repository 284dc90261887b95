#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_reference_order.py tests/test_gpu_mixed_mesh.py tests/test_gpu_poly_mesh.py tests/test_gpu_triple.py -x -q -m gpu 2>&1 | tail -12 | cut -c1-300
bash scripts/gpu_variants.sh "A=1" "ORC_SPMV_XSORT=0" "A=2" "ORC_SPMV_XSORT=0"
