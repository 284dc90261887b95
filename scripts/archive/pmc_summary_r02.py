#!/usr/bin/env python3
"""Folds the passes of scripts/gpu_pmc_r02.sh (gpurun_out/r02_pmc) into profiles/r02_pmc_traffic_per_kernel.csv and
profiles/r02_spmv_pmc.json.  Read bytes are resolved by request size (TCC_EA0_RDREQ_{32B,64B,128B}; the remainder of
TCC_EA0_RDREQ is counted at 32 B), which holds for the gather kernels as well; FETCH_SIZE x 2 (the streaming-kernel rule of
MI355X_MICROARCH.md) is listed beside it for comparison."""
import collections, csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(root, "gpurun_out", "r02_pmc")
N, NNZ = 10_240_000, 71_372_800
per = collections.defaultdict(lambda: collections.defaultdict(list))  # kernel -> counter -> per-dispatch values
for f in sorted(glob.glob(os.path.join(base, "p*", "*", "*counter_collection.csv"))):
    disp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        disp[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, k, c), v in disp.items():
        per[k][c].append(v)
dur = {}
ks = os.path.join(base, "kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        dur[r["Name"].split("(")[0]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
rows = []
for k in sorted(per):
    c = {n: sum(v) / len(v) for n, v in per[k].items()}
    rd, r32, r64, r128 = (c.get(x, 0.) for x in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
    wr, w64 = c.get("TCC_EA0_WRREQ_sum", 0.), c.get("TCC_EA0_WRREQ_64B_sum", 0.)
    read_b = 128. * r128 + 64. * r64 + 32. * max(rd - r128 - r64, 0.)
    write_b = 64. * w64 + 32. * max(wr - w64, 0.)
    calls, us = dur.get(k, (0, 0.))
    rows.append((k, calls, us, rd, r32, r64, r128, wr, w64, read_b, write_b, c.get("FETCH_SIZE", 0.) * 2048., c.get("WRITE_SIZE", 0.) * 1024.,
                 (read_b + write_b) / (us * 1e-6) / 1e12 if us else 0.))
with open(os.path.join(root, "profiles", "r02_pmc_traffic_per_kernel.csv"), "w") as fh:
    fh.write("kernel,calls,avg_us,RDREQ,RDREQ_32B,RDREQ_64B,RDREQ_128B,WRREQ,WRREQ_64B,hbm_read_bytes,hbm_write_bytes,FETCH_SIZE_x2_bytes,WRITE_SIZE_bytes,implied_TB_per_s\n")
    for r in rows:
        fh.write('"%s",%d,%.1f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.2f\n' % r)
spmv = next(r for r in rows if r[0].startswith("void orc::spmv_k<orc::EpiStore,"))
doc = {"workload": "hex channel 400x160x160", "n": N, "nnz": NNZ, "kernel": spmv[0], "avg_us": spmv[2],
       "hbm_read_bytes_per_launch": spmv[9], "hbm_write_bytes_per_launch": spmv[10], "hbm_bytes_per_launch": spmv[9] + spmv[10],
       "algorithmic_bytes_per_launch": 12.0 * NNZ + 20.0 * N, "FETCH_SIZE_x2_bytes": spmv[11], "WRITE_SIZE_bytes": spmv[12],
       "source": "rocprofv3 --pmc TCC_EA0_RDREQ_{sum,32B,64B,128B} / TCC_EA0_WRREQ_{sum,64B} (separate passes) -- python3 scripts/profile_kernels.py; "
                 "bytes = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x rest, writes = 64 x WRREQ_64B + 32 x rest; scripts/gpu_pmc_r02.sh + scripts/pmc_summary_r02.py"}
json.dump(doc, open(os.path.join(root, "profiles", "r02_spmv_pmc.json"), "w"), indent=1)
print(json.dumps(doc, indent=1))
for r in rows:
    print("%-44s %8.1f us  read %8.1f MB  write %8.1f MB  (FETCHx2 %8.1f MB)  %.2f TB/s" % (r[0][-44:], r[2], r[9] / 1e6, r[10] / 1e6, r[11] / 1e6, r[13]))
