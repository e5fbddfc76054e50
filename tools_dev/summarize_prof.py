#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace / PMC) into a small text table.
usage: summarize_prof.py <dir> <out.txt> [kernel-name-substring]"""
import csv, glob, os, sys
from collections import defaultdict
d, out = sys.argv[1], sys.argv[2]
filt = sys.argv[3] if len(sys.argv) > 3 else ""
lines = []
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    lines.append(f"== {os.path.relpath(f, d)}")
    lines += [l.rstrip() for l in open(f)][:40]
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    agg = defaultdict(lambda: [0, 0.0, 0, 0, 0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        a = agg[k]
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a[2] = r.get("VGPR_Count", ""); a[3] = r.get("Accum_VGPR_Count", ""); a[4] = r.get("SGPR_Count", ""); a[5] = r.get("LDS_Block_Size", "")
    lines.append(f"== {os.path.relpath(f, d)}  (calls, total_us, avg_us, vgpr, agpr, sgpr, lds)")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{a[0]:6d} {a[1]:12.1f} {a[1]/a[0]:10.1f} {a[2]:>4} {a[3]:>4} {a[4]:>4} {a[5]:>7}  {k}")
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if filt and filt not in k: continue
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    lines.append(f"== {os.path.relpath(f, d)}  (kernel, counter, dispatches, sum, per-dispatch)")
    for k, cs in agg.items():
        for c, a in sorted(cs.items()):
            lines.append(f"{k:60s} {c:28s} {a[0]:5d} {a[1]:18.1f} {a[1]/a[0]:16.1f}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:80]))
