#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_linear_algebra.py tests/test_gpu_multigrid.py tests/test_gpu_triple.py tests/test_gpu_reference_order.py tests/test_gpu_config3.py -q -m gpu -x > gpurun_out/z_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/z_tests.log
[ $rc -ne 0 ] && exit 1
bash scripts/gpu_variants.sh "ORC_NOP=1" "ORC_SPMV_NT=0" "ORC_SPMV_NT=1"
for v in "ORC_NOP=1" "ORC_SPMV_NT=1"; do
  env $v timeout -k 10 200 python bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('config3 [$v] ms_per_step %.2f inloop frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"
done
