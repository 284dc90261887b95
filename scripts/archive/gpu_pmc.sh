#!/bin/bash
# Run on the GPU box (gpurun -- ./scripts/gpu_pmc.sh): kernel stats + FETCH_SIZE + WRITE_SIZE passes of
# scripts/profile_kernels.py, each counter in its own pass (MI355X_MICROARCH.md, HBM section).  Stops at the first failure.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r01
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01/kern --output-format csv -- python3 scripts/profile_kernels.py > gpurun_out/r01/kern.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r01/pmc_fetch --output-format csv -- python3 scripts/profile_kernels.py > gpurun_out/r01/pmc_fetch.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r01/pmc_write --output-format csv -- python3 scripts/profile_kernels.py > gpurun_out/r01/pmc_write.log 2>&1
echo "rc=$?"
tail -1 gpurun_out/r01/kern.log
