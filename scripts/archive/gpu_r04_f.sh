#!/bin/bash
# r04 call f: CU masks for the set-up streams (measurement), then the BASELINE configs[3] text literally: AMG with a GS smoother (bench + kernel stats)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
run() { tag=$1; shift; env "$@" timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --spmv-reps 3 > gpurun_out/r04f_sw_$tag.json 2> gpurun_out/r04f_sw_$tag.err || { tail -3 gpurun_out/r04f_sw_$tag.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04f_sw_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), d['step_ms'])"; }
run base A=1 || exit 1
run half ORC_SETUP_CU_MASK=55555555 || exit 1
run quarter ORC_SETUP_CU_MASK=11111111 || exit 1
run threeq ORC_SETUP_CU_MASK=77777777 || exit 1
run halfblk ORC_SETUP_CU_MASK=0F0F0F0F || exit 1
run all ORC_SETUP_CU_MASK=FFFFFFFF || exit 1
run base2 A=1 || exit 1
timeout -k 10 400 python bench.py --solver multigrid_gs --steps 5 --warmup 1 > gpurun_out/r04f_config4_gs_bench.json 2> gpurun_out/r04f_config4_gs.err || { tail -5 gpurun_out/r04f_config4_gs.err; exit 1; }
cut -c1-400 gpurun_out/r04f_config4_gs_bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/r04f_gs --output-format csv -- python3 bench.py --solver multigrid_gs --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04f_gs_prof.log 2>&1
cp gpurun_out/r04f_gs/*/*kernel_stats.csv gpurun_out/r04f_config4_gs_kernel_stats.csv; rm -rf gpurun_out/r04f_gs
head -14 gpurun_out/r04f_config4_gs_kernel_stats.csv | cut -c1-100,220-
