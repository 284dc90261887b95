#!/bin/bash
# r05 z3: the uniform products (levels 0-1, plain product without partial sums) with more workgroups than are resident: slices per wavefront 0 (2 048 persistent workgroups), 1, 2, 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_z3
O=gpurun_out/r05_z3
for v in 0 1 2 4 4 2 1 0; do
  ORC_SPMV_WG_SLICES=$v timeout -k 10 300 python3 scripts/profile_products.py --reps 30 > $O/products_$v.log 2>&1 || exit 1
  echo "slices per wavefront = $v: $(grep -o 'plain [0-9.]* us' $O/products_$v.log) $(grep 'level 1' $O/products_$v.log | grep -o '[0-9.]* us')"
done
