#!/bin/bash
# bench.py with two ranks sharing one GPU (host-staged transport): does the N > 1 bench path still run end to end on this tree?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
ORC_BENCH_HOST_TRANSPORT=1 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/two_rank_bench.log 2> gpurun_out/two_rank_bench.err
echo "rc=$?"; tail -1 gpurun_out/two_rank_bench.log | cut -c1-1500; tail -5 gpurun_out/two_rank_bench.err | cut -c1-300
