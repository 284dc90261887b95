#!/bin/bash
# Run on the GPU box: kernel stats + memory-side request counters of scripts/profile_kernels.py (assembly kernels, the
# product, one BiCGSTAB iteration body at 10.24 M cells).  One rocprofv3 --pmc pass per counter group, no trace domains
# beside --pmc.  HBM read bytes per kernel = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x the rest (request-size resolved: valid
# for gather kernels too, where FETCH_SIZE x 2 over-counts); writes = 64 x WRREQ_64B + 32 x the rest.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r02_pmc
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/kern --output-format csv -- python3 scripts/profile_kernels.py > $OUT/kern.log 2>&1 || { echo "kernel-trace pass failed"; tail -3 $OUT/kern.log; exit 1; }
i=0
for G in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $G -d $OUT/p$i --output-format csv -- python3 scripts/profile_kernels.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $G"; grep -m3 "error code" $OUT/p$i.log; continue; }
  echo "pass $i ok: $G"
done
cp $OUT/kern/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/kern
tail -1 $OUT/kern.log
