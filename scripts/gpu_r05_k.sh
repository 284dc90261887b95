#!/bin/bash
# r05 k (evidence 1): counters + kernel statistics of the products and of ONE hierarchy's set-up at the benchmark's size -> gpurun_out/r05_pmc (scripts/pmc_summary.py folds it)
bash "$GRAFT_REPO_ROOT/scripts/gpu_pmc.sh" r05_pmc 6 full
