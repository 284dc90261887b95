#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_mixed_mesh.py tests/test_gpu_reference_order.py tests/test_gpu_poly_mesh.py -q -m gpu -x > gpurun_out/v_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/v_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/v_conc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/v_conc.log 2>&1
grep -o '"ms_per_step": [0-9.]*' gpurun_out/v_conc.log
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/v_conc/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "scan" in r["Name"] or "slice_sizes" in r["Name"] or "galerkin_bound" in r["Name"]:
        print("   %-44s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf gpurun_out/v_conc
