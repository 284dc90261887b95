#!/bin/bash
# r04 call af: kernel time of the window builds in the two builds (one-stream schedule, rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_af
O=gpurun_out/r04_af
cp orc_amd/liborc_amd.so $O/new.so; cp orc_amd/liborc_amd_alt.so $O/old.so
for v in new old; do
  cp $O/$v.so orc_amd/liborc_amd.so
  ORC_CONCURRENT_MOMENTUM=0 ORC_TWO_STREAM_MULTIGRID=0 ORC_EARLY_P_HIERARCHY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_$v --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/$v.log 2>&1
  cp $O/p_$v/*/*kernel_stats.csv $O/stats_$v.csv; rm -rf $O/p_$v
  echo "== $v"; grep "xwin_build" $O/stats_$v.csv | cut -d'"' -f2,3 | cut -c1-60,120-
done
rm -f $O/new.so $O/old.so
