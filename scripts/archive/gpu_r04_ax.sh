#!/bin/bash
# r04 call ax: the certification pass from the preference lists too: pairing tests, then A/B against the previous build
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ax
timeout -k 10 800 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_triple.py tests/test_gpu_reference_order.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py tests/test_gpu_poly_mesh.py -q -x --timeout=700 > gpurun_out/r04_ax/tests.log 2>&1
rc=$?; tail -2 gpurun_out/r04_ax/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/r04_ax/tests.log | head -20; exit $rc; fi
bash scripts/gpu_r04_w.sh
