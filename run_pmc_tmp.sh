#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r01/kern --output-format csv -- python3 scripts/profile_kernels.py > gpurun_out/r01/kern.log 2>&1; echo "stats rc=$?"; tail -1 gpurun_out/r01/kern.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/r01/pmc_fetch --output-format csv -- python3 scripts/profile_kernels.py > gpurun_out/r01/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/r01/pmc_write --output-format csv -- python3 scripts/profile_kernels.py > gpurun_out/r01/pmc_write.log 2>&1; echo "write rc=$?"
ls gpurun_out/r01/*/* | head -20
