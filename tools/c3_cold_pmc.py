#!/usr/bin/env python3
"""Labels the k_search4 dispatches of `rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/c3_cold.py` runs with the
phases that script printed (first / replay / ring / replay2 / flushed / s_ring / s_replay) and prints, per phase, the mean of
every counter and of the dispatch duration, plus the derived figures the cold-launch question needs:

    mean L2->fabric read latency = TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum      (cycles a read request is outstanding)
    read requests per lookup     = TCC_EA0_RDREQ_sum / requests per launch
    translation misses per lookup = TCP_UTCL1_TRANSLATION_MISS_sum / requests per launch

    python tools/c3_cold_pmc.py <dir with pass*/ subdirectories and pass*.log> [out.csv]
"""
import collections
import csv
import glob
import json
import os
import sys

base = sys.argv[1]
out_csv = sys.argv[2] if len(sys.argv) > 2 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))      # phase -> counter -> values
dur = collections.defaultdict(list)
reqs = None
for log in sorted(glob.glob(os.path.join(base, "pass*.log"))):
    tag = os.path.basename(log)[:-4]
    phases = None
    for ln in open(log, errors="replace"):
        if ln.startswith("PHASES "):
            phases = json.loads(ln[7:])
        if ln.startswith("SUMMARY "):
            reqs = json.loads(ln[8:])["requests_per_launch"]
    if phases is None:
        print("no PHASES line in", log)
        continue
    rows = []
    for f in glob.glob(os.path.join(base, tag, "*", "*_counter_collection.csv")):
        rows += list(csv.DictReader(open(f)))
    if not rows:
        print("no counters for", tag)
        continue
    last_walk = max([int(r["Dispatch_Id"]) for r in rows if "k_lf_walk" in r["Kernel_Name"]] or [0])
    disp = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        if "k_search4" in r["Kernel_Name"] and int(r["Dispatch_Id"]) > last_walk:
            disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                disp[int(r["Dispatch_Id"])]["__ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            if r.get("SGPR_Count"):
                disp[int(r["Dispatch_Id"])]["__sgpr"] = float(r["SGPR_Count"])
                disp[int(r["Dispatch_Id"])]["__vgpr"] = float(r.get("VGPR_Count") or 0)
    ids = list(disp)
    if len(ids) != len(phases):
        print("%s: %d k_search4 dispatches after the pattern generation, %d phases printed -- not labelled" % (tag, len(ids), len(phases)))
        continue
    seen = collections.Counter()
    for did, (ph, b) in zip(ids, phases):
        seen[ph] += 1
        label = ph
        if ph in ("replay", "replay2") and seen[ph] == 1:
            label = ph + ":1st"
        for c, v in disp[did].items():
            agg[label][c].append(v)
order = ["first", "replay:1st", "replay", "ring", "replay2:1st", "replay2", "fl32M", "fl128M", "fl512M", "flushed", "rd512M", "rd1G", "idle20ms", "s_ring", "s_replay"]
counters = sorted({c for ph in agg for c in agg[ph]})
w = csv.writer(open(out_csv, "w", newline="")) if out_csv else None
if w:
    w.writerow(["Phase", "Counter", "Dispatches", "Mean"])
print("requests per launch:", reqs)
for ph in order:
    if ph not in agg:
        continue
    m = {c: sum(v) / len(v) for c, v in agg[ph].items()}
    line = ["%-12s" % ph]
    for c in counters:
        if c in m:
            if w:
                w.writerow([ph, c, len(agg[ph][c]), "%.6g" % m[c]])
            if not c.startswith("__"):
                line.append("%s=%.4g" % (c.replace("_sum", ""), m[c]))
    if "__ns" in m:
        line.append("us(under pmc)=%.1f" % (m["__ns"] / 1e3))
    if "TCC_EA0_RDREQ_sum" in m and "TCC_EA0_RDREQ_LEVEL_sum" in m and m["TCC_EA0_RDREQ_sum"]:
        line.append("LAT=%.0f cyc" % (m["TCC_EA0_RDREQ_LEVEL_sum"] / m["TCC_EA0_RDREQ_sum"]))
    if reqs and "TCC_EA0_RDREQ_sum" in m:
        line.append("rd/lookup=%.2f" % (m["TCC_EA0_RDREQ_sum"] / reqs))
    if reqs and "TCP_UTCL1_TRANSLATION_MISS_sum" in m:
        line.append("tlbmiss/lookup=%.2f" % (m["TCP_UTCL1_TRANSLATION_MISS_sum"] / reqs))
    print("  ".join(line))
