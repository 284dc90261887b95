#!/usr/bin/env python3
"""Folds the passes of scripts/gpu_pmc.sh (gpurun_out/<tag>) into
  profiles/<prefix>_pmc_products.csv   one row per product kernel (and per level for spmv_xwin_k): time, memory-side bytes, cache and
                                  pipe counters per launch;
  profiles/<prefix>_spmv_pmc.json      the traffic of the kernel bench.py's roofline names — bench.py quotes it only if the kernel
                                  name and the matrix (n, nnz) are the ones it has just timed;
  profiles/<prefix>_products_kernel_stats.csv, <prefix>_setup.csv   the rocprofv3 --kernel-trace --stats summary of the same program.
Memory-side bytes follow MI355X_MICROARCH.md (HBM): reads = FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes),
writes = WRITE_SIZE, both in KiB, each from its own --pmc pass; the request-size-resolved figure (128 x RDREQ_128B + 64 x
RDREQ_64B + 32 x the rest) is listed beside it."""
import argparse, collections, csv, glob, json, os, shutil, sys
ap = argparse.ArgumentParser()
ap.add_argument("--tag", default="r05_pmc", help="directory under gpurun_out/ that scripts/gpu_pmc.sh wrote")
ap.add_argument("--prefix", default="r05", help="profiles/<prefix>_*.csv / .json")
ap.add_argument("--n", type=int, default=10_240_000)
ap.add_argument("--nnz", type=int, default=71_372_800)
ap.add_argument("--workload", default="hex channel 400x160x160, a_u through two Jacobi scalings (scripts/profile_products.py)")
ap.add_argument("--xwin-levels", default="2,3", help="the levels whose products are spmv_xwin_k launches, in the order profile_products.py runs them")
args = ap.parse_args()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(root, "gpurun_out", args.tag)
N, NNZ = args.n, args.nnz
PRE = args.prefix


def short(name):
    return name.split("(")[0].replace("void ", "").replace("orc::", "")


# per kernel: counter -> list of (dispatch id, value)
per = collections.defaultdict(lambda: collections.defaultdict(list))
# gpurun MERGES what a call wrote into gpurun_out/: files of an earlier run of the script stay beside the new ones — per pass only the
# newest counter file counts
passes = collections.defaultdict(list)
for f in glob.glob(os.path.join(base, "p*", "*", "*counter_collection.csv")):
    passes[os.path.dirname(f)].append(f)
for f in sorted(max(v, key=os.path.getmtime) for v in passes.values()):
    disp = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        disp[(int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in sorted(disp.items()):
        per[k][c].append((d, v))
# durations per dispatch from the kernel trace (same program, same order of launches)
trace = collections.defaultdict(list)
kt = os.path.join(base, "kernel_trace.csv")
if os.path.exists(kt):
    for r in sorted(csv.DictReader(open(kt)), key=lambda r: int(r["Start_Timestamp"])):
        trace[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


XWIN_LEVELS = [int(x) for x in args.xwin_levels.split(",")]


def split_levels(k, values):
    """spmv_xwin_k runs for one level after the other under the same kernel name (the hex channel: levels 2 and 3; config 5 since r05: 1, 2, 3):
    equal shares of the dispatches, in order."""
    m = len(XWIN_LEVELS)
    if not k.startswith("spmv_xwin_k") or len(values) < m:
        return {"": values}
    h = len(values) // m
    return {" [level %d]" % lv: values[q * h:(q + 1) * h if q + 1 < m else len(values)] for q, lv in enumerate(XWIN_LEVELS)}


rows = []
want = ("spmv_uniform_k", "spmv3_uniform_k", "spmv_xwin_k", "spmv_k")
for k in sorted(per):
    if not k.startswith(want):
        continue
    parts = collections.defaultdict(dict)
    for c, dv in per[k].items():
        for tag, vals in split_levels(k, [v for _, v in dv]).items():
            parts[tag][c] = sum(vals) / len(vals)
    durs = split_levels(k, trace.get(k, []))
    for tag, c in sorted(parts.items()):
        d = durs.get(tag, [])
        us = sum(d) / len(d) if d else 0.
        rd, r64, r128 = (c.get(x, 0.) for x in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
        wr, w64 = c.get("TCC_EA0_WRREQ_sum", 0.), c.get("TCC_EA0_WRREQ_64B_sum", 0.)
        read_req = 128. * r128 + 64. * r64 + 32. * max(rd - r128 - r64, 0.)
        write_req = 64. * w64 + 32. * max(wr - w64, 0.)
        read_b, write_b = c.get("FETCH_SIZE", 0.) * 2. * 1024., c.get("WRITE_SIZE", 0.) * 1024.
        rows.append(dict(kernel=k + tag, launches=len(d), avg_us=us, hbm_read_bytes=read_b, hbm_write_bytes=write_b, read_bytes_by_request_size=read_req,
                         write_bytes_by_request_size=write_req, TB_per_s=(read_b + write_b) / (us * 1e-6) / 1e12 if us else 0., counters=c))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
extra = ["TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TCP_TOTAL_CACHE_ACCESSES_sum",
         "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum",
         "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_ANY",
         "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU"]
with open(os.path.join(root, "profiles", PRE + "_pmc_products.csv"), "w") as fh:
    fh.write("kernel,launches,avg_us,hbm_read_bytes(FETCH_SIZEx2),hbm_write_bytes(WRITE_SIZE),read_bytes_by_request_size,write_bytes_by_request_size,TB_per_s," + ",".join(extra) + "\n")
    for r in rows:
        fh.write('"%s",%d,%.1f,%.0f,%.0f,%.0f,%.0f,%.2f,' % (r["kernel"], r["launches"], r["avg_us"], r["hbm_read_bytes"], r["hbm_write_bytes"],
                                                              r["read_bytes_by_request_size"], r["write_bytes_by_request_size"], r["TB_per_s"]))
        fh.write(",".join("%.0f" % r["counters"].get(x, float("nan")) for x in extra) + "\n")
ks = os.path.join(base, "kernel_stats.csv")
if os.path.exists(ks):
    shutil.copyfile(ks, os.path.join(root, "profiles", PRE + "_products_kernel_stats.csv"))
# the kernel bench.py's roofline names: the two in-loop launches of one system, averaged (one of each per BiCGSTAB iteration)
# the in-loop launches of one system on level 0: spmv_uniform_k<EpiStoreSum | EpiTs, false, true, narrow, scaled>
pair = [r for r in rows if r["kernel"].startswith(("spmv_uniform_k<EpiStoreSum, false, true", "spmv_uniform_k<EpiTs, false, true"))]
if len(pair) == 2:
    names = [r["kernel"] for r in sorted(pair, key=lambda r: "EpiTs" in r["kernel"])]
    kernel_name = names[0] + " / " + names[1].replace("spmv_uniform_k", "")
    avg = lambda key: sum(r[key] for r in pair) / 2.  # noqa: E731
    doc = {"workload": args.workload, "n": N, "nnz": NNZ,
           "kernel": kernel_name, "avg_us": avg("avg_us"),
           "hbm_read_bytes_per_launch": avg("hbm_read_bytes"), "hbm_write_bytes_per_launch": avg("hbm_write_bytes"),
           "hbm_bytes_per_launch": avg("hbm_read_bytes") + avg("hbm_write_bytes"),
           "bytes_by_request_size_per_launch": avg("read_bytes_by_request_size") + avg("write_bytes_by_request_size"),
           "algorithmic_bytes_per_launch": 12.0 * NNZ + 20.0 * N,
           "bytes_the_launch_must_move_with_4_byte_columns": 12.0 * NNZ + 20.0 * N + 4.0 * N,  # EpiTs also re-reads s (8 n): 4 n on average
           "per_epilogue": {r["kernel"]: {"avg_us": r["avg_us"], "hbm_bytes": r["hbm_read_bytes"] + r["hbm_write_bytes"]} for r in pair},
           "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB; reads doubled: gfx950 tallies 128-byte requests at 64 bytes, "
                     "MI355X_MICROARCH.md HBM); scripts/gpu_pmc.sh + scripts/pmc_summary.py"}
    json.dump(doc, open(os.path.join(root, "profiles", PRE + "_spmv_pmc.json"), "w"), indent=1)
    print(json.dumps(doc, indent=1))
for r in rows:
    c = r["counters"]
    print("%-52s %3d x %7.1f us  read %8.1f MB  write %7.1f MB  %.2f TB/s   TA busy %4.0f%%  L2 hit %4.0f%%  LDS conflicts/inst %.2f" % (
        r["kernel"][-52:], r["launches"], r["avg_us"], r["hbm_read_bytes"] / 1e6, r["hbm_write_bytes"] / 1e6, r["TB_per_s"],
        # TA_TA_BUSY_sum adds the 256 texture addressers' busy cycles, GRBM_GUI_ACTIVE the 8 XCDs' active cycles: per addresser busy / (active / 8)
        100. * c.get("TA_TA_BUSY_sum", 0.) / max(c.get("GRBM_GUI_ACTIVE", 0.) * 32, 1.) if c.get("GRBM_GUI_ACTIVE") else float("nan"),
        100. * c.get("TCC_HIT_sum", 0.) / max(c.get("TCC_HIT_sum", 0.) + c.get("TCC_MISS_sum", 0.), 1.),
        c.get("SQ_LDS_BANK_CONFLICT", 0.) / max(c.get("SQ_INSTS_LDS", 0.), 1.)))


# ---- [r04] the set-up of ONE hierarchy (a_u: three aggregations + three Galerkin products, built by orc_bench_amg_levels inside the same program):
# per phase the launches, the kernel time and the memory-side bytes (FETCH_SIZE x 2 + WRITE_SIZE per launch, summed) — VERDICT r03, weak #3:
# "0.25 s of the 0.82 s iteration is hierarchy set-up with no byte model"
PHASES = [("aggregation [r05]: deferred acceptance (first pass, chains, pairing, verification)", ("da_first_k", "da_chase_k", "da_finish_k", "agg_reset_k", "agg_verify_k")),
          ("aggregation [r04 fallback]: slice sweeps", ("agg_init_k", "agg_init_prefs_k", "agg_scatter_k", "agg_sweep_group_k", "agg_sweep_k", "agg_rotate_k")),
          ("aggregation [r04 fallback]: lock-step rounds", ("tail_eval_k", "tail_commit_k", "tail_update_k", "tail_rotate_k", "tail_seed_k", "tail_small_k")),
          ("aggregation [r04 fallback]: cascades", ("tail_chase_k", "chase_carry_k", "chase_rotate_k", "chase_convert_k")),
          ("aggregation: chooser table", ("chooser_k",)),
          ("Galerkin: bounds + scans", ("galerkin_bound_k", "scan_i64_k")),
          ("Galerkin: merge", ("galerkin_merge_k", "galerkin_wave_k")),
          ("Galerkin: pack (SELL image + packed mirror)", ("slice_sizes_k", "scan2_i64_k", "galerkin_pack_fused_k")),
          ("mirrors: row-contiguous, LDS windows, narrow columns", ("rows_compact_k", "xwin_build_k", "narrow_build_k", "xsort_build_k")),
          ("scaled values + diagonals", ("scale_values_k", "scale_packed_k", "scale_values3_k", "diag_inverse_k", "diag_inverse3_k"))]
setup_rows = []
for phase, names in PHASES:
    launches, us, rd, wr = 0, 0., 0., 0.
    for k in per:
        if not k.split("<")[0] in names:
            continue
        d = trace.get(k, [])
        launches += len(d)
        us += sum(d)
        rd += sum(v for _, v in per[k].get("FETCH_SIZE", [])) * 2. * 1024.
        wr += sum(v for _, v in per[k].get("WRITE_SIZE", [])) * 1024.
    setup_rows.append((phase, launches, us / 1e3, rd / 1e9, wr / 1e9, (rd + wr) / (us * 1e-6) / 1e12 if us else 0.))
with open(os.path.join(root, "profiles", PRE + "_setup.csv"), "w") as fh:
    fh.write("phase (one hierarchy of a_u, %d rows: levels 0->1, 1->2, 2->3; one stream; no sibling pairing),launches" % N + ",kernel_ms,read_GB(FETCH_SIZEx2),written_GB(WRITE_SIZE),TB_per_s\n")
    for r in setup_rows:
        fh.write('"%s",%d,%.2f,%.2f,%.2f,%.2f\n' % r)
    tot = [sum(r[i] for r in setup_rows) for i in (1, 2, 3, 4)]
    fh.write('"total",%d,%.2f,%.2f,%.2f,%.2f\n' % (tot[0], tot[1], tot[2], tot[3], (tot[2] + tot[3]) / max(tot[1], 1e-9)))
print()
for r in setup_rows:
    print("%-52s %5d launches %8.2f ms  read %6.2f GB  written %6.2f GB  %.2f TB/s" % r)
