#!/bin/bash
# Run on the GPU box: counter passes over the per-level products (scripts/profile_levels.py), one rocprofv3 --pmc pass per
# counter group (no trace domains beside --pmc: the pool refuses that combination).  usage: gpu_pmc_levels.sh <variant> <tag>
V=${1:-0}; TAG=${2:-v$V}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_levels_$TAG
mkdir -p $OUT
i=0
for G in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
         "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $G -d $OUT/p$i --output-format csv -- python3 scripts/profile_levels.py --variant $V > $OUT/p$i.log 2>&1 || { echo "pass $i failed: $G"; grep -m3 "error code\|Error" $OUT/p$i.log; continue; }
  echo "pass $i ok: $G"
done
# keep only the counter csv files (small)
find $OUT -name "*.csv" | head -20
