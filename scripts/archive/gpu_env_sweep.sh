#!/bin/bash
# usage: scripts/gpu_env_sweep.sh VAR v1 v2 ... : bench.py once per value of the environment variable (STEPS, default 3)
cd "$GRAFT_REPO_ROOT" || exit 1
var=$1; shift
for v in "$@"; do
  echo -n "$var=$v  "
  env $var=$v timeout -k 10 400 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' || exit 1
done
