#!/bin/bash
# r04 call ad: the three-system product on resident workgroups (virtual grid): bit-identity tests, then A/B against the previous build
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ad
timeout -k 10 800 python -m pytest tests/test_gpu_triple.py tests/test_gpu_full_size.py tests/test_gpu_reference_order.py tests/test_gpu_bench_family.py tests/test_gpu_solve_steady.py -q -x --timeout=700 > gpurun_out/r04_ad/tests.log 2>&1
rc=$?; tail -3 gpurun_out/r04_ad/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/r04_ad/tests.log | head -20; exit $rc; fi
bash scripts/gpu_r04_w.sh
