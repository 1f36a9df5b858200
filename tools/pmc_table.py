#!/usr/bin/env python3
"""Per-launch averages of every counter under out_dir/<variant>/<pass>/ (written by tools/pmc_variants.sh)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "pass2_kernel<false"
for var in sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))):
    acc = defaultdict(list)
    dur = []
    for f in glob.glob(os.path.join(root, var, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(root, var, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    # a counter row exists per dispatch (and possibly per dimension): average over the dispatches
    print(f"== {var}: {len(dur)} launches, mean {sum(dur)/max(1,len(dur)):.3f} ms")
    for k in sorted(acc):
        v = acc[k]
        print(f"   {k:28s} {sum(v)/len(v):18.1f}   (n={len(v)})")
