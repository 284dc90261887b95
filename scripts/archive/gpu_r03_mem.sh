#!/bin/bash
# companion arena for the transient row mirrors: multigrid / triple / partition tests, memory and wall time of the bench, window statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_triple.py tests/test_gpu_solve_steady.py tests/test_gpu_bench_family.py -q -m gpu -x > gpurun_out/mem_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/mem_tests.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/mem_bench.log 2> gpurun_out/mem_bench.err && echo "bench rc=$?" && python - <<'PY'
import json
d = json.loads(open("gpurun_out/mem_bench.log").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "hbm", d["config"].get("hbm_used_gb"), "inloop", d["roofline"]["frac"])
PY
ORC_AMG_MIRROR_ARENA=0 timeout -k 10 300 python bench.py --steps 5 --warmup 2 > gpurun_out/mem_bench_off.log 2> gpurun_out/mem_bench_off.err && python - <<'PY'
import json
d = json.loads(open("gpurun_out/mem_bench_off.log").read().strip().splitlines()[-1])
print("[mirror arena off] ms_per_step", d["ms_per_step"], "hbm", d["config"].get("hbm_used_gb"))
PY
ORC_XWIN_STATS=1 timeout -k 10 300 python bench.py --steps 1 --warmup 0 > gpurun_out/xwin_stats.log 2> gpurun_out/xwin_stats.err; grep "orc xwin" gpurun_out/xwin_stats.err | sort | uniq -c | sort -rn | head -20
