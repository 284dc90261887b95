#!/bin/bash
# r04 call x: the multicolour-GS kernels with scalar slice descriptors: their tests, then BASELINE configs[2] (bench)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_x
O=gpurun_out/r04_x
timeout -k 10 600 python -m pytest tests/test_gpu_gauss_seidel.py tests/test_gpu_config3.py -q -x --timeout=500 > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/tests.log | head -20; exit $rc; fi
for r in 1 2; do
timeout -k 10 300 python3 bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 --no-cpu-baseline > $O/config3_$r.json 2> $O/config3_$r.err || exit 1
python -c "import json;d=json.load(open('$O/config3_$r.json'));print('config3', d['ms_per_step'], d['roofline'].get('gauss_seidel_sweep',{}).get('three_systems',{}))" | cut -c1-400
done
