#!/bin/bash
# r05 t: the concurrent schedule of one SIMPLE iteration under the kernel trace: time with solve kernels resident, with set-up kernels only, idle
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05_t
O=gpurun_out/r05_t
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/conc --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/conc.log 2>&1 || { tail -5 $O/conc.log; exit 1; }
T=$(ls $O/conc/*/*kernel_trace.csv | head -1)
python3 scripts/analysis/kernel_timeline.py $T > $O/timeline_concurrent.txt; cat $O/timeline_concurrent.txt
python3 - $T $O/trace_small.csv <<'PY'
import csv, sys
# a compact copy of the trace for further analysis off the box: start, end (us from the first kernel), queue, grid, short name
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
with open(sys.argv[2], "w") as f:
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("orc::", "")[:48]
        f.write("%.1f,%.1f,%s,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Queue_Id"], r["Grid_Size_X"], n))
PY
rm -rf $O/conc; ls -la $O
