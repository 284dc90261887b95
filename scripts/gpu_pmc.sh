#!/bin/bash
# Run on the GPU box: kernel stats + counters of scripts/profile_products.py — the products bench.py's roofline is about, as the solver launches them
# (level 0: spmv_uniform_k / spmv3_uniform_k, coarse levels: spmv_uniform_k / spmv_xwin_k) and ONE hierarchy's set-up.  One rocprofv3 --pmc pass per
# counter group, no trace domain beside --pmc.  Folded by scripts/pmc_summary.py into profiles/<round>_*.
#   usage: gpu_pmc.sh <out tag, e.g. r05_pmc> <reps> <"full" | "bytes" (FETCH_SIZE and WRITE_SIZE only)> [profile_products.py arguments ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05_pmc}; REPS=${2:-6}; WHAT=${3:-full}; shift 3
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kern --output-format csv -- python3 scripts/profile_products.py --reps $REPS "$@" > $OUT/kern.log 2>&1 || { echo "kernel-trace pass failed"; tail -3 $OUT/kern.log; exit 1; }
cp $OUT/kern/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cp $OUT/kern/*/*kernel_trace.csv $OUT/kernel_trace.csv 2>/dev/null
rm -rf $OUT/kern
grep "cells\|level" $OUT/kern.log
PASSES_BYTES=("FETCH_SIZE" "WRITE_SIZE")
PASSES_FULL=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
         "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
         "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum"
         "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY"
         "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU")
if [ "$WHAT" = bytes ]; then PASSES=("${PASSES_BYTES[@]}"); else PASSES=("${PASSES_FULL[@]}"); fi
i=0
for G in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $G -d $OUT/p$i --output-format csv -- python3 scripts/profile_products.py --reps $REPS "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $G"; grep -m2 -i "error\|invalid" $OUT/p$i.log | cut -c1-200; continue; }
  echo "pass $i ok: $G"
  find $OUT/p$i -type f ! -name "*counter_collection.csv" -delete  # keep the counter csv only
done
du -sh $OUT
