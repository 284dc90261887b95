#!/bin/bash
# r05 o: how the LDS x windows of the bench hierarchies (hex channel and config 5) are made up: runs of consecutive columns, and what bridging small gaps buys
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_o
O=gpurun_out/r05_o
ORC_XWIN_STATS=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/hex.json 2> $O/hex.err || exit 1
grep "orc xwin" $O/hex.err | sort | uniq -c | sort -k1,1nr > $O/hex_windows.txt; head -64 $O/hex_windows.txt
ORC_XWIN_STATS=1 timeout -k 10 300 python3 bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline > $O/config5.json 2> $O/config5.err || exit 1
grep "orc xwin" $O/config5.err | sort | uniq -c | sort -k1,1nr > $O/config5_windows.txt; head -64 $O/config5_windows.txt
