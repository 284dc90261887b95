#!/bin/bash
# Run on the GPU box: kernel trace (no stats) of a short bench.py run in the concurrent schedule + scripts/timeline.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/tl --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/tl.log 2>&1
f=$(ls gpurun_out/tl/*/*kernel_trace.csv | head -1)
python3 scripts/timeline.py "$f"
rm -rf gpurun_out/tl
