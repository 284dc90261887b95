#!/bin/bash
# r04 evidence run (part 4, final tree): counters + kernel stats of the products (scripts/gpu_pmc_r04.sh), then the bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_final_d
O=gpurun_out/r04_final_d
bash scripts/gpu_pmc_r04.sh 6 > $O/pmc.log 2>&1; tail -3 $O/pmc.log
timeout -k 10 400 python3 bench.py --steps 10 --warmup 3 --levels-csv $O/levels.csv > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-330 $O/bench.json
