#!/bin/bash
# r04 evidence run (part 6): the final tree at the benchmark's own sizes, in the reference's own mode, against the committed oracle trajectories
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_final_f
O=gpurun_out/r04_final_f
timeout -k 10 500 python scripts/reference_mode_fullsize.py --nx 400 --ny 160 --nz 160 --iterations 3 --oracle profiles/r03_oracle_trajectory_400x160x160_inplace.json --out $O/reference_mode_400x160x160.json > $O/hex.log 2>&1
echo "hex rc=$?"; tail -3 $O/hex.log | cut -c1-400
timeout -k 10 500 python scripts/reference_mode_fullsize.py --workload config5 --nx 252 --ny 100 --nz 72 --iterations 3 --oracle profiles/r04_oracle_trajectory_config5_252x100x72_inplace.json --out $O/reference_mode_config5_252x100x72.json > $O/c5.log 2>&1
echo "config5 rc=$?"; tail -3 $O/c5.log | cut -c1-400
