#!/bin/bash
# Run on the GPU box: kernel-trace stats of one bench.py run.  usage: scripts/gpu_profile.sh <tag> <bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag --output-format csv -- python bench.py "$@" > gpurun_out/prof_$tag.log 2>&1
echo "rc=$?"
tail -1 gpurun_out/prof_$tag.log | cut -c1-300
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -16 "$f" | cut -c1-180; fi
