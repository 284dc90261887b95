#!/bin/bash
# r04 call d: full GPU suite (prints of the partition tests kept), config-3 bench with the slot-space GS solver, kernel stats + timeline of the headline bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout=900 -rP -k "not test_two_ranks_converged" > gpurun_out/r04d_tests.log 2>&1
rc=$?
grep -E "passed|failed" gpurun_out/r04d_tests.log | tail -3
grep -E "lock-step partitioned|mixed slabs|FAILED" gpurun_out/r04d_tests.log | head -20
if [ $rc -gt 1 ]; then echo "pytest ended with $rc: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 > gpurun_out/r04d_config3.json 2> gpurun_out/r04d_config3.err
rc2=$?
cut -c 1-300 gpurun_out/r04d_config3.json; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04d_config3.json").read().strip().splitlines()[-1])
    print("config3 ms_per_step", d["ms_per_step"], "gs", json.dumps(d["roofline"]["gauss_seidel_sweep"])[:600])
except Exception as e:
    print("config3 parse failed", e)
PY
if [ $rc2 -gt 1 ]; then exit $rc2; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r04d_c3 --output-format csv -- python3 bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04d_config3_prof.log 2>&1
cp gpurun_out/r04d_c3/*/*kernel_stats.csv gpurun_out/r04d_config3_kernel_stats.csv; rm -rf gpurun_out/r04d_c3
head -12 gpurun_out/r04d_config3_kernel_stats.csv | cut -c1-90,200-
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r04d_conc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04d_conc.log 2>&1
cp gpurun_out/r04d_conc/*/*kernel_stats.csv gpurun_out/r04d_bench_multigrid_concurrent_3steps_kernel_stats.csv
f=$(ls gpurun_out/r04d_conc/*/*kernel_trace.csv | head -1)
python3 scripts/timeline.py "$f" > gpurun_out/r04d_timeline.txt 2>&1
rm -rf gpurun_out/r04d_conc
tail -40 gpurun_out/r04d_timeline.txt
