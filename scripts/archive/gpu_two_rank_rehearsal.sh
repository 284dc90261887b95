#!/bin/bash
# Two ranks sharing ONE GPU through the host-staged transport (rehearsal of the partitioned path; the numbers are not
# multi-GPU numbers).  usage: scripts/gpu_two_rank_rehearsal.sh [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
ORC_BENCH_HOST_TRANSPORT=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
