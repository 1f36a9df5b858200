#!/usr/bin/env python3
"""Developer aid (GPU box): the kernel timeline of the LAST multi-level navigation (lmi_nav_order) + search in a rocprofv3 kernel trace.
  python3 tools/nb_timeline.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nav_pop" in r["Kernel_Name"]]
print(len(rows), "kernels,", len(idx), "nav_pop launches")
last = idx[-1]
start = last
while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 400_000:
    start -= 1
t0 = int(rows[start]["Start_Timestamp"])
prev = t0
for r in rows[start:last + 40]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s - prev > 2_000_000:
        break
    print("%-64s gap %7.1f run %7.1f  at %8.1f us" % (r["Kernel_Name"][:64], (s - prev) / 1e3, (e - s) / 1e3, (s - t0) / 1e3))
    prev = e
