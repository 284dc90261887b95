#!/bin/bash
# r04 evidence run (part 8, final tree): the N-rank entries rehearsed with two ranks on one GPU (host-staged transport: a launch rehearsal, not a
# measurement), and BASELINE configs[4] / configs[2] through bench.py on one GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_final_h
O=gpurun_out/r04_final_h
ORC_BENCH_HOST_TRANSPORT=1 timeout -k 10 400 python3 bench.py --workload config5 --gpus 2 --nx 60 --ny 30 --nz 30 --steps 2 --warmup 1 --no-cpu-baseline > $O/config5_2ranks_host.json 2> $O/config5_2ranks.err; echo "config5 2 ranks rc=$?"; cut -c1-300 $O/config5_2ranks_host.json
ORC_BENCH_HOST_TRANSPORT=1 timeout -k 10 400 python3 bench.py --gpus 2 --nx 100 --ny 40 --nz 40 --steps 2 --warmup 1 --no-cpu-baseline > $O/hex_2ranks_host.json 2> $O/hex_2ranks.err; echo "hex 2 ranks rc=$?"; cut -c1-300 $O/hex_2ranks_host.json
timeout -k 10 700 python3 bench.py --workload config5 --steps 5 --warmup 1 > $O/config5_bench.json 2> $O/config5.err; echo "config5 rc=$?"; cut -c1-300 $O/config5_bench.json
timeout -k 10 300 python3 bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 > $O/config3_bench.json 2> $O/config3.err; echo "config3 rc=$?"; cut -c1-300 $O/config3_bench.json
