#!/bin/bash
# r04 call e: GS / triple / partition tests after the vector-kernel rewrite, config-3 bench, per-stream timeline, scheduling switches
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_gauss_seidel.py tests/test_gpu_config3.py tests/test_gpu_triple.py tests/test_gpu_multigrid.py "tests/test_gpu_partition.py::test_two_ranks_on_rank_local_mixed_poly_slabs_match_single_rank" "tests/test_gpu_partition.py::test_lock_step_momentum_solve_on_a_partitioned_mesh" "tests/test_gpu_partition.py::test_rccl_overlapped_product_on_a_self_loop_communicator" -q --timeout=600 > gpurun_out/r04e_tests.log 2>&1
rc=$?
grep -E "passed|failed|FAILED" gpurun_out/r04e_tests.log | tail -5
if [ $rc -gt 1 ]; then echo "pytest ended with $rc: stopping"; exit $rc; fi
timeout -k 10 200 python bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04e_config3.json 2> gpurun_out/r04e_config3.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04e_config3.json").read().strip().splitlines()[-1])
g = d["roofline"]["gauss_seidel_sweep"]
print("config3 ms_per_step %.2f  sweep3 %.1f us frac %.3f  sweep1 %.1f us frac %.3f" % (d["ms_per_step"], g["three_systems"]["avg_sweep_ms"] * 1e3, g["three_systems"]["frac"], g["one_system"]["avg_sweep_ms"] * 1e3, g["one_system"]["frac"]))
PY
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/r04e_tl --output-format csv -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --spmv-reps 3 > gpurun_out/r04e_tl.log 2>&1 || exit 1
f=$(ls gpurun_out/r04e_tl/*/*kernel_trace.csv | head -1)
python3 scripts/timeline.py "$f" > gpurun_out/r04e_timeline.txt 2>&1
rm -rf gpurun_out/r04e_tl
tail -12 gpurun_out/r04e_timeline.txt | cut -c1-1500
run() { tag=$1; shift; env "$@" timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --spmv-reps 3 > gpurun_out/r04e_sw_$tag.json 2> gpurun_out/r04e_sw_$tag.err || return 1; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04e_sw_$tag.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],1), d['step_ms'])"; }
run base A=1 || exit 1
run prio0 ORC_STREAM_PRIORITIES=0 || exit 1
run prio2 ORC_STREAM_PRIORITIES=2 || exit 1
run pearly ORC_P_HIERARCHY_LATE=0 || exit 1
run steps256 ORC_AMG_CHASE_STEPS=256 || exit 1
run steps1M ORC_AMG_CHASE_STEPS=1000000 || exit 1
run batch4 ORC_AMG_CHASE_BATCH=4 || exit 1
run base2 A=1 || exit 1
