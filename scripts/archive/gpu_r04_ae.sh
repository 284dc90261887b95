#!/bin/bash
# r04 call ae: window descriptions built in two passes (small bitmap first): window tests, then A/B against the previous build
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ae
timeout -k 10 800 python -m pytest tests/test_gpu_window_fallback.py tests/test_gpu_multigrid.py tests/test_gpu_mixed_mesh.py tests/test_gpu_config5.py -q -x --timeout=700 > gpurun_out/r04_ae/tests.log 2>&1
rc=$?; tail -3 gpurun_out/r04_ae/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/r04_ae/tests.log | head -20; exit $rc; fi
bash scripts/gpu_r04_w.sh
