#!/bin/bash
# r04 call ac: the three-system product's launch against its residency (82 VGPRs: five wavefronts per SIMD = 1280 workgroups; the launch has 2048):
# ORC_SPMV_GRID caps every product's grid — only the three-system figure is read here
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ac
O=gpurun_out/r04_ac
run() { local name=$1; shift
  env "$@" timeout -k 10 250 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err || exit 1
  python -c "import json;d=json.load(open('$O/bench_$name.json'));r=d['roofline'];print('$name', round(d['ms_per_step'],1), 'L0 x1 %.1f us x3 %.1f us' % (1e3*r['avg_launch_ms'], 1e3*r['three_systems_per_launch']['avg_launch_ms']))"
}
for round in 1 2; do
  run default_$round ORC_DUMMY=1
  run g1280_$round ORC_SPMV_GRID=1280
  run g1024_$round ORC_SPMV_GRID=1024
done
