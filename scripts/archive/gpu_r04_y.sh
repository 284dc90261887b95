#!/bin/bash
# r04 call y: exactness tests of the window product with the new build, then A/B against the previous build (scripts/gpu_r04_w.sh)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_y
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_linear_algebra.py tests/test_gpu_full_size.py tests/test_gpu_reference_order.py -q -x --timeout=500 > gpurun_out/r04_y/tests.log 2>&1
rc=$?; tail -3 gpurun_out/r04_y/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/r04_y/tests.log | head -20; exit $rc; fi
bash scripts/gpu_r04_w.sh
