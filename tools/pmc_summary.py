#!/usr/bin/env python3
"""Summarise rocprofv3 outputs per kernel: average of every counter over the dispatches of each kernel (counter
collection CSVs of one or more --pmc passes) and average duration (kernel-trace CSVs).
    pmc_summary.py DIR [DIR ...] [--match SUBSTR] [--json OUT]
Counter values are summed over the dimension rows of a dispatch (rocprofv3 writes one row per counter instance)."""
import csv, glob, json, os, sys, collections

dirs, match, jout = [], None, None
a = sys.argv[1:]
while a:
    v = a.pop(0)
    if v == "--match": match = a.pop(0)
    elif v == "--json": jout = a.pop(0)
    else: dirs.append(v)

cnt = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))  # kernel -> counter -> dispatch -> value
dur = collections.defaultdict(list)
meta = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if match and match not in k: continue
            cnt[k][r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
            meta[k] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if match and match not in k: continue
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)

out = {}
for k in sorted(set(cnt) | set(dur)):
    short = k if len(k) < 110 else k[:107] + "..."
    print(short)
    o = out.setdefault(k, {})
    if k in meta: print("   ", meta[k]); o["meta"] = meta[k]
    if dur[k]:
        v = dur[k]
        print(f"    dispatches {len(v)}  avg {sum(v)/len(v):.2f} us  min {min(v):.2f}  max {max(v):.2f}")
        o["avg_us"], o["n"], o["min_us"], o["max_us"] = sum(v) / len(v), len(v), min(v), max(v)
    wc = None
    if "SQ_WAVE_CYCLES" in cnt[k]:
        vals = list(cnt[k]["SQ_WAVE_CYCLES"].values()); wc = sum(vals) / len(vals)
    for c in sorted(cnt[k]):
        vals = list(cnt[k][c].values())
        avg = sum(vals) / len(vals)
        o[c] = avg
        extra = f"   {avg/wc:.3f} of SQ_WAVE_CYCLES" if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else ""
        print(f"    {c:28s} {avg:16.1f}  ({len(vals)} dispatches){extra}")
if jout:
    json.dump(out, open(jout, "w"), indent=1)
