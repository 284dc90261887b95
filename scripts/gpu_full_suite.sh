#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tee gpurun_out/full_suite.log | tail -40
