#!/bin/bash
# r04 call ak: tier lists in row order: coarse-operator tests, kernel time of the merges in both builds, then A/B of the iteration
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_ak
timeout -k 10 600 python -m pytest tests/test_gpu_multigrid.py tests/test_gpu_triple.py tests/test_gpu_mixed_mesh.py -q -x --timeout=500 > gpurun_out/r04_ak/tests.log 2>&1
rc=$?; tail -2 gpurun_out/r04_ak/tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" gpurun_out/r04_ak/tests.log | head -20; exit $rc; fi
bash scripts/gpu_r04_af.sh
python3 - <<'PY'
import csv
for v in ('new','old'):
    tot=0
    for r in csv.DictReader(open(f'gpurun_out/r04_af/stats_{v}.csv')):
        if 'galerkin_merge' in r['Name'] or 'galerkin_bound' in r['Name']:
            tot+=int(r['TotalDurationNs']); print(v, r['Calls'], '%.2f ms' % (int(r['TotalDurationNs'])/1e6), r['Name'][:48])
    print(v, 'total %.2f ms' % (tot/1e6))
PY
bash scripts/gpu_r04_w.sh
