#!/usr/bin/env python3
"""Condenses rocprofv3 output (gpurun_out/prof_rNN/) into the small files committed under profiles/.

  python profiles/summarize.py gpurun_out/prof_r01 r01 [dominant-kernel-substring]

Writes profiles/<tag>_kernel_stats.csv (top kernels by total time, names shortened),
profiles/<tag>_scan_pmc.json (per-launch FETCH_SIZE / WRITE_SIZE of lmi::scan_kernel and the HBM
bytes derived as /opt/skills/guides/MI355X_MICROARCH.md section "HBM" prescribes: both counters are
in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16-B-per-lane coalesced streaming reads --
which is what both operand streams of the scan kernel are -- so it is doubled; WRITE_SIZE is exact).
"""
import csv
import glob
import json
import os
import sys


def one(pattern):
    files = glob.glob(pattern, recursive=True)   # gpurun merges new files into the directory: leftovers of an earlier pass may lie beside them
    return max(files, key=os.path.getmtime) if files else None


DOMINANT = "lmi::scan_kernel("


def is_dom(name):
    return DOMINANT in name


def main(src, tag, dominant=None):
    global DOMINANT
    if dominant:
        DOMINANT = dominant
    out_dir = os.path.dirname(os.path.abspath(__file__))
    stats = one(os.path.join(src, "trace", "**", "*_kernel_stats.csv"))
    if stats:
        rows = list(csv.DictReader(open(stats)))
        with open(os.path.join(out_dir, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows[:25]:
                w.writerow([r["Name"][:96], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"]])
    pmc = {}
    # the dominant kernel's launches in the kernel trace: the stats file averages the redo launch of pass 2 (returns at once, same
    # name) in with the real ones, so the real ones (>= 20 % of the longest) are averaged here from the trace itself
    trace = one(os.path.join(src, "trace", "**", "*_kernel_trace.csv"))
    if trace:
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(trace)) if is_dom(r["Kernel_Name"])]
        real = [v for v in durs if v >= 0.2 * max(durs)] if durs else []
        if real:
            pmc["kernel_trace"] = {"launches_all": len(durs), "launches_real": len(real), "avg_ms_real": sum(real) / len(real) / 1e6,
                                   "min_ms_real": min(real) / 1e6, "max_ms_real": max(real) / 1e6,
                                   "avg_ms_steady": sum(sorted(real)[: max(1, len(real) * 3 // 4)]) / max(1, len(real) * 3 // 4) / 1e6,
                                   "what": "rocprofv3 --kernel-trace of the bench command; real = launches >= 20 % of the longest (the redo launch "
                                           "returns at once); steady = the fastest three quarters of them (the first launches run while the clock settles)"}
    for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        f = one(os.path.join(src, sub, "**", "*_counter_collection.csv"))
        if not f:
            continue
        vals, durs, meta = [], [], None
        rows_f = [r for r in csv.DictReader(open(f)) if is_dom(r["Kernel_Name"]) and r["Counter_Name"] == name]
        longest = max([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows_f] or [0])
        for r in rows_f:
            if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 0.2 * longest:   # (the redo launch of pass 2 returns at once)
                vals.append(float(r["Counter_Value"]))
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                          "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")}
        if vals:
            pmc[name] = {"launches": len(vals), "avg_kib": sum(vals) / len(vals), "min_kib": min(vals),
                         "max_kib": max(vals), "avg_duration_ms_profiled": sum(durs) / len(durs) / 1e6}
            pmc["dispatch"] = meta
    extra = {}
    for sub in ("pmc_sq", "pmc_tcc"):
        f = one(os.path.join(src, sub, "**", "*_counter_collection.csv"))
        if not f:
            continue
        acc = {}
        rows_f = [r for r in csv.DictReader(open(f)) if is_dom(r["Kernel_Name"])]
        longest = max([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows_f] or [0])
        for r in rows_f:
            if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 0.2 * longest:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            extra[k] = sum(v) / len(v)
    if extra:
        pmc["per_launch_avg"] = extra
        if "SQ_VALU_MFMA_BUSY_CYCLES" in extra and "GRBM_GUI_ACTIVE" in extra:
            # 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs
            pmc["mfma_pipe_busy_frac"] = extra["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (extra["GRBM_GUI_ACTIVE"] / 8.0)
        if "TCC_HIT" in extra and "TCC_MISS" in extra:
            pmc["l2_hit_rate"] = extra["TCC_HIT"] / (extra["TCC_HIT"] + extra["TCC_MISS"])
    if "FETCH_SIZE" in pmc:
        fetch = pmc["FETCH_SIZE"]["avg_kib"] * 1024.0 * 2.0  # gfx950: half-counted 16-B/lane streams
        write = pmc.get("WRITE_SIZE", {}).get("avg_kib", 0.0) * 1024.0
        pmc["hbm_bytes_per_launch"] = fetch + write
        pmc["note"] = ("(2 * FETCH_SIZE + WRITE_SIZE) KiB -> bytes; FETCH_SIZE counts L2 misses, Infinity-Cache "
                       "hits included, so this is an upper bound on true HBM reads")
    pmc["kernel"] = DOMINANT
    # provenance: what the PROFILED bench run itself reported about the library it had loaded (path, sha256/16 of the .so, the
    # source hash baked into it by csrc/build.sh) and when it ran -- copied, never recomputed here: this script runs later and in
    # another container than the collection (tools/profile_round.sh).  bench.py replays hbm_bytes_per_launch / mfma_pipe_busy_frac
    # into a bench line only when the library IT loaded reports the same built-from hash.
    import subprocess
    bj = one(os.path.join(src, "*bench_under_rocprof.json"))
    pmc["lib"], pmc["collected_utc"] = None, None
    if bj:
        for ln in open(bj):
            ln = ln.strip()
            if ln.startswith("{"):
                try:
                    j = json.loads(ln)
                except ValueError:
                    continue
                pmc["lib"], pmc["collected_utc"] = j.get("lib"), j.get("collected_utc")
    if pmc["lib"] is None:
        print("no provenance: the profiled bench line carries no `lib` field; bench.py will not replay this summary", file=sys.stderr)
    try:
        pmc["commit"] = subprocess.run(["git", "-C", os.path.dirname(out_dir), "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:  # noqa: BLE001
        pmc["commit"] = None
    pmc["commit_is"] = "HEAD when this summary was written (the collection's own stamp is `lib` / `collected_utc`)"
    with open(os.path.join(out_dir, f"{tag}_scan_pmc.json"), "w") as fh:
        json.dump(pmc, fh, indent=1)
    print(json.dumps(pmc, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
