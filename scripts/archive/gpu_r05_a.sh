#!/bin/bash
# r05 a: the new frozen-assembly / frozen-iteration pins
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_a
O=gpurun_out/r05_a
timeout -k 10 500 python -m pytest tests/test_gpu_bench_family.py -q -m gpu -x --durations=10 -k "frozen" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
