#!/usr/bin/env bash
# Build container: condense the merged gpurun_out/prof_<tag> of tools/profile_round.sh into profiles/ and install the per-leg replays.
set -e
python profiles/summarize.py gpurun_out/prof_r05 r05 "pass2_kernel<false" > /dev/null
python profiles/summarize.py gpurun_out/prof_r05hard r05hard "pass2_kernel<false" > /dev/null
python profiles/summarize.py gpurun_out/prof_r05x r05x "scan_kernel" > /dev/null
python profiles/summarize.py gpurun_out/prof_r05c5 r05c5 "pass2_small_kernel<3, false" > /dev/null
python profiles/summarize.py gpurun_out/prof_r05c1 r05c1 "pass2_kernel<false" > /dev/null
cp profiles/r05_scan_pmc.json profiles/scan_pmc_c2.json
cp profiles/r05c5_scan_pmc.json profiles/scan_pmc_c5.json
cp profiles/r05c1_scan_pmc.json profiles/scan_pmc_c1.json
cp profiles/r05hard_scan_pmc.json profiles/scan_pmc_hard.json
cp profiles/r05x_scan_pmc.json profiles/scan_pmc_c2_exact.json
for t in r05 r05c5 r05c1 r05hard r05x; do cp gpurun_out/prof_$t/${t}_bench_under_rocprof.json profiles/${t}_bench_under_rocprof.json; done
python3 - <<'PY'
import json, subprocess
cur = subprocess.check_output(["python3", "learnedmetricindex_amd/_srchash.py"]).decode().strip()
for t in ("c2", "c5", "c1", "hard", "c2_exact"):
    j = json.load(open(f"profiles/scan_pmc_{t}.json"))
    print(t, j["lib"]["built_from_source_sha16"], "== tree" if j["lib"]["built_from_source_sha16"] == cur else "!= tree " + cur,
          "kernel avg ms", round(j["kernel_trace"]["avg_ms_real"], 4), "busy", round(j.get("mfma_pipe_busy_frac", 0), 3), "GB", round(j.get("hbm_bytes_per_launch", 0) / 1e9, 2))
PY
