#!/bin/bash
# r04 call m: GS tests after frozen-system skipping, the mixed mid-size golden test, the three-rank lock-step test, config-3 bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest "tests/test_gpu_partition.py::test_lock_step_momentum_solve_on_three_ranks" -q --timeout=600 -rP > gpurun_out/r04m_tests.log 2>&1
rc=$?
grep -E "passed|failed|FAILED" gpurun_out/r04m_tests.log | tail -5; grep -E "lock-step partitioned" gpurun_out/r04m_tests.log | head
if [ $rc -gt 1 ]; then exit $rc; fi
grep "three or more" gpurun_out/r04m_tests.log
timeout -k 10 600 python scripts/reference_mode_fullsize.py --workload config5 --nx 252 --ny 100 --nz 72 --iterations 3 --oracle profiles/r04_oracle_trajectory_config5_252x100x72_inplace.json --out gpurun_out/r04_reference_mode_config5_252x100x72.json > gpurun_out/r04m_refmode_c5.log 2>&1
echo "reference mode config5 rc=$?"; tail -4 gpurun_out/r04m_refmode_c5.log | cut -c1-700
