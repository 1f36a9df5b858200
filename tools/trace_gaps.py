#!/usr/bin/env python3
"""Timeline of one search step out of a rocprofv3 kernel trace: per kernel (in launch order on the busiest queue) the average
duration and the average idle gap before it (start minus the latest end of anything earlier on that queue).

  python3 tools/trace_gaps.py gpurun_out/trace_<tag>/trace [anchor-kernel-substring]

A step is cut at each occurrence of the anchor kernel (default: route_kernel when the trace has it, else pack_queries16_kernel)."""
import collections
import csv
import glob
import sys


def main(src, anchor=None):
    f = glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    if anchor is None:   # the first kernel of a scan: route_kernel (round 5), else the old preparation's pack kernel
        anchor = "route_kernel" if any("route_kernel" in r["Kernel_Name"] for r in rows) else "pack_queries16_kernel"
    by_q = collections.defaultdict(list)
    for r in rows:
        by_q[r["Queue_Id"]].append(r)
    # every queue that carries a scan (the pipelined loop and the resident loop run on different streams)
    for q in sorted(by_q):
        if any("pass2_" in r["Kernel_Name"] for r in by_q[q]):
            one_queue(q, sorted(by_q[q], key=lambda r: int(r["Start_Timestamp"])), anchor)


def one_queue(q, rs, anchor):
    steps, cur = [], []
    for r in rs:
        if anchor in r["Kernel_Name"] and cur:
            steps.append(cur)
            cur = []
        cur.append(r)
    steps.append(cur)
    # every step shape seen at least 5 times (the pipelined loop and the resident loop differ; warm-up / set-up sequences drop out)
    shapes = collections.Counter(tuple(r["Kernel_Name"] for r in s) for s in steps)
    print(f"queue {q}: {len(steps)} steps cut at '{anchor}'")
    for shape, cnt in shapes.most_common():
        if cnt < 5:
            continue
        keep = [s for s in steps if tuple(r["Kernel_Name"] for r in s) == shape]
        n = len(keep)
        print(f"--- {n} steps of {len(shape)} launches")
        tot_d = tot_g = 0.0
        for i, name in enumerate(shape):
            d = sum(int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"]) for s in keep) / n / 1e3
            g = 0.0
            if i:
                g = sum(int(s[i]["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in s[:i]) for s in keep) / n / 1e3
            tot_d += d
            tot_g += g
            short = name.split("(")[0].replace("lmi::", "").replace("void ", "")[:56]
            print(f"{i:3d} {short:56s} gap {g:7.1f} us   run {d:8.1f} us")
        span = sum(int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in keep) / n / 1e3
        print(f"sum of runs {tot_d:.1f} us, sum of gaps {tot_g:.1f} us, first start -> last end {span:.1f} us")

if __name__ == "__main__":
    main(*sys.argv[1:])
