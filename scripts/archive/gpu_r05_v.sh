#!/bin/bash
# r05 v: the set-up's long-running kernels (chains of proposals: all 32 wave slots of every CU; Galerkin merges: 24) at a fraction of their launch width, so that
# the products beside them keep their wave slots: the bench at 100 / 50 / 25 / 12 percent, alternating
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_v
O=gpurun_out/r05_v
i=0
for sh in 100 50 25 12 12 25 50 100; do
  i=$((i+1))
  ORC_AMG_SETUP_SHARE=$sh timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_share${sh}_$i.json 2> $O/bench_share${sh}_$i.err || exit 1
  python3 -c "import json,sys; d=json.load(open('$O/bench_share${sh}_$i.json')); print('share=$sh ms_per_step %.1f' % d['ms_per_step'], [round(x) for x in d['step_ms']])"
done
