#!/bin/bash
# r05 s: the padding-aware mirror rule (a ragged level of >= 12 entries per row takes the packed mirror + windows): parity on the mixed meshes, then config 5 and the hex channel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_s
O=gpurun_out/r05_s
timeout -k 10 600 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py tests/test_gpu_config5.py tests/test_gpu_full_size.py -m gpu -x -q --durations=5 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python3 bench.py --workload config5 --steps 4 --warmup 1 --no-cpu-baseline --levels-csv $O/c5_levels.csv > $O/c5_bench.json 2> $O/c5_bench.err || exit 1
python3 -c "import json,sys; d=json.load(open('$O/c5_bench.json')); print('config5 ms_per_step %.1f' % d['ms_per_step'])"; cut -d, -f1,2,3,5,6,9 $O/c5_levels.csv
timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --levels-csv $O/levels.csv > $O/bench.json 2> $O/bench.err || exit 1
python3 -c "import json,sys; d=json.load(open('$O/bench.json')); print('hex ms_per_step %.1f' % d['ms_per_step'])"; cut -d, -f1,2,3,5,6,9 $O/levels.csv
timeout -k 10 300 python3 scripts/reference_mode_fullsize.py --workload config5 --nx 252 --ny 100 --nz 72 --oracle profiles/r04_oracle_trajectory_config5_252x100x72_inplace.json --out $O/reference_mode_config5_252x100x72.json > $O/reference_mode_config5.log 2>&1; echo "reference mode config5 rc=$?"; tail -1 $O/reference_mode_config5.log
