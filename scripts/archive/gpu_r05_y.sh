#!/bin/bash
# r05 y: where an iteration of configs[3]-as-written (--solver multigrid_gs: the multicolour Gauss-Seidel smoother, an extension) and of configs[2]
# (bicgstab_gs) spends its kernel time
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_y
O=gpurun_out/r05_y
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/gs --output-format csv -- python3 bench.py --solver multigrid_gs --steps 2 --warmup 1 --no-cpu-baseline > $O/gs.log 2>&1 || { tail -5 $O/gs.log; exit 1; }
cp $O/gs/*/*kernel_stats.csv $O/multigrid_gs_kernel_stats.csv; rm -rf $O/gs
python3 - $O/multigrid_gs_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:22]:
    print("%-64s %7d %9.1f ms %5.1f%% avg %8.1f us" % (r["Name"].replace("void ", "").replace("orc::", "")[:64], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]), float(r["AverageNs"]) / 1e3))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/c3 --output-format csv -- python3 bench.py --nx 512 --ny 2016 --nz 1 --momentum quick --solver bicgstab_gs --steps 10 --warmup 2 --no-cpu-baseline > $O/c3.log 2>&1 || { tail -5 $O/c3.log; exit 1; }
cp $O/c3/*/*kernel_stats.csv $O/config3_kernel_stats.csv; rm -rf $O/c3
python3 - $O/config3_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f" % (tot / 1e6))
for r in rows[:22]:
    print("%-64s %7d %9.1f ms %5.1f%% avg %8.1f us" % (r["Name"].replace("void ", "").replace("orc::", "")[:64], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]), float(r["AverageNs"]) / 1e3))
PY
tail -2 $O/c3.log | cut -c1-300
