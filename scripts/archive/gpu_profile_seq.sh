#!/bin/bash
# Run on the GPU box: kernel-trace stats of a bench.py run with every solve on ONE stream (no lanes, no side stream, no early
# p' hierarchy), so that kernel durations are not stretched by sharing.  usage: scripts/gpu_profile_seq.sh <tag> [env=val ...] -- <bench args>
tag=$1; shift
while [ "$1" != "--" ] && [ -n "$1" ]; do export "$1"; shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag --output-format csv -- python3 bench.py "$@" > gpurun_out/prof_$tag.log 2>&1
echo "rc=$?"
grep -o "\"ms_per_step\": [0-9.]*" gpurun_out/prof_$tag.log
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/prof_${tag}_kernel_stats.csv; head -24 "$f" | cut -c1-60,200-; fi
rm -rf gpurun_out/prof_$tag
