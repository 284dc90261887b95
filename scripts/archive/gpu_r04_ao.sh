#!/bin/bash
# r04 call ao: cascades at eight wavefronts per SIMD (64 VGPRs, 16 bytes of scratch per lane; the launch grows to the 2 048 resident workgroups):
# pairing tests, then A/B against the previous build
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ao
timeout -k 10 500 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_bench_family.py -q -x --timeout=400 > gpurun_out/r04_ao/tests.log 2>&1
rc=$?; tail -2 gpurun_out/r04_ao/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/r04_ao/tests.log | head; exit $rc; fi
bash scripts/gpu_r04_w.sh
