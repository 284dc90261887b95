#!/bin/bash
# r04 call k: full GPU suite after the overlapped lock-step products and the two-queue rehearsal setting; the two-rank hex rehearsal that did not finish; smoke
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu --timeout=900 -rP > gpurun_out/r04k_tests.log 2>&1
rc=$?
grep -E "passed|failed" gpurun_out/r04k_tests.log | tail -3
grep -E "lock-step partitioned|mixed slabs|FAILED" gpurun_out/r04k_tests.log | head -20
if [ $rc -gt 1 ]; then echo "pytest ended with $rc: stopping"; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04k_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r04k_smoke.log
ORC_BENCH_HOST_TRANSPORT=1 ORC_BENCH_WATCHDOG=250 timeout -k 10 300 python3 bench.py --gpus 2 --nx 100 --ny 40 --nz 40 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04k_hex_2ranks_host.json 2> gpurun_out/r04k_hex_2ranks.err; echo "hex 2 ranks rc=$?"; cut -c1-300 gpurun_out/r04k_hex_2ranks_host.json
