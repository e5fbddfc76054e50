#!/usr/bin/env python3
"""Print the kernel timeline of the LAST search in a rocprofv3 kernel-trace CSV (start offset, duration, gap).
usage: trace_timeline.py <dir>"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# a search starts with init_search_kernel, or -- when the prep launch does the resets itself (option fuse) -- with the prep kernel
def is_start(i):
    nm = rows[i]["Kernel_Name"]
    if "init_search_kernel" in nm: return True
    return ("prep_q16" in nm or "prep_q8" in nm) and not (i > 0 and "init_search_kernel" in rows[i - 1]["Kernel_Name"])
starts = [i for i in range(len(rows)) if is_start(i)]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-n], starts[-n + 1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {r.get('Grid_Size', '?'):>8}  {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"total {(prev_end - t0) / 1e3:.1f} us")
