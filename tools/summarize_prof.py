#!/usr/bin/env python3
"""Condenses a tools/rocprof_passes.sh output directory into the files kept under profiles/:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, fmx kernels only
   <tag>_counters.csv       per kernel, per counter: dispatches, mean value
   <tag>_summary.md         durations + HBM traffic with the gfx950 FETCH_SIZE correction
Usage: tools/summarize_prof.py gpurun_out/prof_r1 profiles/r01_c3
"""
import collections
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)


def short(name):
    return name.split("(")[0].replace("fmx::", "")


stats = []
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "fmx::" in r["Name"]:
            stats.append(r)
with open(dst + "_kernel_stats.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
    for r in stats:
        w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"],
                    r["StdDev"]])

# fmx_prepare calibrates the search kernel with a few 60-us launches of k_search4 itself (its residency census, round 5): they
# are the dispatches of that kernel BEFORE the workload generates its patterns (k_lf_walk) and are left out of every figure
# of k_search4 below -- counted, and said so in the summary.
def is_calibration(rows_of_file, r):
    if "k_search4" not in r["Kernel_Name"]:
        return False
    walks = [int(x["Dispatch_Id"]) for x in rows_of_file if "k_lf_walk" in x["Kernel_Name"]]
    return bool(walks) and int(r["Dispatch_Id"]) < min(walks)


agg = collections.defaultdict(list)
meta = {}
n_calib = 0
for f in glob.glob(os.path.join(src, "pmc*", "*", "*_counter_collection.csv")):
    rows_f = list(csv.DictReader(open(f)))
    for r in rows_f:
        if "fmx::" not in r["Kernel_Name"]:
            continue
        if is_calibration(rows_f, r):
            continue
        k = short(r["Kernel_Name"])
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
# the search kernel's durations without the calibration launches, from the kernel trace of pass 0
search_ns = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")):
    rows_f = list(csv.DictReader(open(f)))
    for r in rows_f:
        if "k_search4" in r["Kernel_Name"]:
            if is_calibration(rows_f, r):
                n_calib += 1
            else:
                search_ns[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(dst + "_counters.csv", "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["Kernel", "Counter", "Dispatches", "Mean"])
    for (k, c), v in sorted(agg.items()):
        w.writerow([k, c, len(v), "%.6g" % (sum(v) / len(v))])


def mean(k, c):
    v = agg.get((k, c))
    return sum(v) / len(v) if v else None


# Calibration of FETCH_SIZE for this access pattern: the workload's k_occ launch reads a known byte count.
import re
calib_bytes = None
for f in glob.glob(os.path.join(src, "*.log")):
    m = re.search(r"CALIB kernel=k_occ queries=(\d+) line_bytes=(\d+) read_bytes=(\d+)", open(f, errors="replace").read())
    if m:
        calib_bytes = int(m.group(3))
occ_key = next((k for k, c in agg if c == "FETCH_SIZE" and "k_occ" in k), None)
bytes_per_kib = 1024.0            # uncalibrated: FETCH_SIZE taken at face value
if calib_bytes and occ_key:
    bytes_per_kib = calib_bytes / mean(occ_key, "FETCH_SIZE")
calls = None          # regex workloads: one fmx_regex_batch_match = several k_frontier dispatches
for f in glob.glob(os.path.join(src, "*.log")):
    m = re.search(r"frontier: (\d+) calls;", open(f, errors="replace").read())
    if m:
        calls = int(m.group(1)) + 2          # prof_workload.py makes two untimed calls first
src_sha = None        # the source hash the profiled build was made from (prof_workload.py prints bench.source_hash())
for f in glob.glob(os.path.join(src, "*.log")):
    m = re.search(r"SRC_SHA16=([0-9a-f]{16})", open(f, errors="replace").read())
    if m:
        src_sha = m.group(1)
with open(dst + "_counters.csv", "a", newline="") as fo:
    csv.writer(fo).writerow(["(calibration)", "FETCH_BYTES_PER_KIB", 1, "%.6g" % bytes_per_kib])
    if src_sha:
        csv.writer(fo).writerow(["(source)", "SRC_SHA16", 1, src_sha])
    if calls:
        csv.writer(fo).writerow(["(workload)", "CALLS", 1, calls])


with open(dst + "_summary.md", "w") as fo:
    fo.write("# rocprofv3 summary (%s)\n\n" % os.path.basename(dst))
    fo.write("Source: `tools/rocprof_passes.sh` (pass 0 `--kernel-trace --stats`; one `--pmc` group per further pass), "
             "workload `tools/prof_workload.py`.\n\n")
    fo.write("| kernel | calls | avg ms | min ms | max ms | VGPR | SGPR | LDS B |\n|---|---|---|---|---|---|---|---|\n")
    for r in stats:
        k = short(r["Name"])
        m = meta.get(k, ("", "", "", "", ""))
        fo.write("| %s | %s | %.4f | %.4f | %.4f | %s | %s | %s |\n" % (
            k, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, m[0], m[1], m[2]))
    for k, v in sorted(search_ns.items()):
        fo.write("| %s WITHOUT fmx_prepare's %d calibration launches (60-us launches of this kernel that count its resident "
                 "workgroups: not searches) | %d | %.4f | %.4f | %.4f | | | |\n" % (k, n_calib, len(v), sum(v) / len(v) / 1e6, min(v) / 1e6, max(v) / 1e6))
    if search_ns:
        fo.write("\nEvery counter of `k_search4` below is a mean over its SEARCH launches only (the calibration launches, which precede the "
                 "workload's pattern generation, are left out); the workload rotates a ring of distinct batches through them.\n")
    fo.write("\nHBM traffic per dispatch.  FETCH_SIZE / WRITE_SIZE are in KiB.  FETCH_SIZE is calibrated on the "
             "workload's k_occ launch, whose read bytes are known (%s B: one rank-dictionary block per random query plus "
             "the streamed (c, i) inputs): one KiB of FETCH_SIZE stands for %.0f bytes in this access pattern "
             "(MI355X_MICROARCH.md, HBM: wide reads are tallied at half size on gfx950, other widths need calibrating).\n\n"
             % (calib_bytes, bytes_per_kib))
    fo.write("| kernel | FETCH_SIZE KiB | read GB (calibrated) | WRITE_SIZE KiB | write GB | RDREQ | RDREQ_32B | L2 hit rate |\n"
             "|---|---|---|---|---|---|---|---|\n")
    for k in sorted({k for k, _ in agg}):
        fs, ws = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
        hit, miss = mean(k, "TCC_HIT_sum"), mean(k, "TCC_MISS_sum")
        rq, rq32 = mean(k, "TCC_EA0_RDREQ_sum"), mean(k, "TCC_EA0_RDREQ_32B_sum")
        fo.write("| %s | %s | %s | %s | %s | %s | %s | %s |\n" % (
            k, "%.0f" % fs if fs else "", "%.3f" % (fs * bytes_per_kib / 1e9) if fs else "",
            "%.0f" % ws if ws else "", "%.3f" % (ws * 1024 / 1e9) if ws else "",
            "%.4g" % rq if rq else "", "%.4g" % rq32 if rq32 is not None else "",
            "%.3f" % (hit / (hit + miss)) if hit is not None and miss else ""))
    fo.write("\nSQ counters (quad-cycles, summed over waves): see `%s_counters.csv`.\n" % os.path.basename(dst))
    for k in sorted({k for k, _ in agg}):
        wc = mean(k, "SQ_WAVE_CYCLES")
        if wc:
            fo.write("- %s: WAIT_ANY %.0f%%, WAIT_INST_ANY %.0f%%, ACTIVE_INST_ANY %.0f%% of SQ_WAVE_CYCLES; "
                     "VALU insts %.4g, VMEM reads %.4g, waves %d\n" % (
                         k, 100 * mean(k, "SQ_WAIT_ANY") / wc, 100 * mean(k, "SQ_WAIT_INST_ANY") / wc,
                         100 * mean(k, "SQ_ACTIVE_INST_ANY") / wc, mean(k, "SQ_INSTS_VALU"),
                         mean(k, "SQ_INSTS_VMEM_RD"), mean(k, "SQ_WAVES")))
print("wrote", dst + "_{kernel_stats.csv,counters.csv,summary.md}")
