#!/bin/bash
# r05 p: LDS x windows described by runs of consecutive columns (XWinDev::desc) against the column list: parity subset, window statistics, then the
# bench with runs on / off in alternating order on one box (level products from the levels csv), hex channel and config 5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_p
O=gpurun_out/r05_p
timeout -k 10 500 python3 -m pytest tests/test_gpu_multigrid.py tests/test_gpu_window_fallback.py tests/test_gpu_bench_family.py tests/test_gpu_mixed_mesh.py tests/test_gpu_full_size.py -m gpu -x -q --durations=5 > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 $O/tests.log
[ $rc -eq 0 ] || exit 1
ORC_XWIN_STATS=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/stats.json 2> $O/stats.err || exit 1
grep "orc xwin" $O/stats.err | sort | uniq -c
show() {
python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
lv = d.get("levels") or d.get("roofline", {}).get("levels") or []
print(sys.argv[2], "ms_per_step %.1f" % d["ms_per_step"], json.dumps(lv)[:300])
PY
}
for pass in 1:1 0:2 0:3 1:4; do
  r=${pass%%:*}; i=${pass##*:}
  ORC_XWIN_RUNS=$r timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --levels-csv $O/levels_runs${r}_$i.csv > $O/bench_runs${r}_$i.json 2> $O/bench_runs${r}_$i.err || exit 1
  show $O/bench_runs${r}_$i.json "hex runs=$r"; cat $O/levels_runs${r}_$i.csv
done
for pass in 1:1 0:2 0:3 1:4; do
  r=${pass%%:*}; i=${pass##*:}
  ORC_XWIN_RUNS=$r timeout -k 10 200 python3 bench.py --workload config5 --steps 4 --warmup 1 --no-cpu-baseline --levels-csv $O/c5_levels_runs${r}_$i.csv > $O/c5_bench_runs${r}_$i.json 2> $O/c5_bench_runs${r}_$i.err || exit 1
  show $O/c5_bench_runs${r}_$i.json "config5 runs=$r"; cat $O/c5_levels_runs${r}_$i.csv
done
