#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/gpu_variants.sh "A=1" "ORC_AMG_BULK=2" "ORC_AMG_BULK=3" "A=2" "ORC_AMG_BULK=2"
for v in "A=1" "ORC_AMG_BULK=2" "ORC_AMG_BULK=3"; do
  env $v ORC_AMG_TRACE=1 ORC_CONCURRENT_MOMENTUM=0 ORC_EARLY_P_HIERARCHY=0 ORC_TWO_STREAM_MULTIGRID=0 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --spin-up 2 --no-cpu-baseline > gpurun_out/trace_seq.log 2> gpurun_out/trace_seq.err
  echo "== $v"; python scripts/amg_phases.py gpurun_out/trace_seq.err | grep "all  \|evaluations"
done
