#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_mixed_mesh.py tests/test_gpu_poly_mesh.py tests/test_gpu_triple.py -x -q -m gpu 2>&1 | tail -3
bash scripts/gpu_variants.sh "A=1" "ORC_AMG_SWEEP_GROUP=0" "A=2" "ORC_AMG_SWEEP_GROUP=0"
for v in "A=1" "ORC_AMG_SWEEP_GROUP=0"; do
  env $v ORC_AMG_TRACE=1 ORC_CONCURRENT_MOMENTUM=0 ORC_EARLY_P_HIERARCHY=0 ORC_TWO_STREAM_MULTIGRID=0 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --spin-up 2 --no-cpu-baseline > gpurun_out/trace_seq.log 2> gpurun_out/trace_seq.err
  echo "== $v"; python scripts/amg_phases.py gpurun_out/trace_seq.err | grep "bulk sweeps\|all        all\|cascades"
done
