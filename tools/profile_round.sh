#!/usr/bin/env bash
# GPU box: the rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats of the bench, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE, SQ, TCC) restricted to the dominant kernel, condensed by profiles/summarize.py.
#   bash tools/profile_round.sh r02 [dominant-kernel-regex] [extra bench flags, e.g. "--config c5"]
set -uo pipefail
export TMPDIR=/tmp
tag="$1"; K="${2:-pass2_kernel<false}"; EXTRA="${3:-}"
root="$PWD"; out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
B="python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-recall --no-hard-leg --no-other-configs --no-exact-leg $EXTRA"
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B > "$out/bench_under_rocprof.json" 2> "$out/trace.err" || { echo "trace pass failed"; tail -5 "$out/trace.err"; exit 1; }
echo "trace done"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "$K" --output-format csv -d "$out/pmc_fetch" -- $B > /dev/null 2> "$out/fetch.err" || { echo "fetch pass failed"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "$K" --output-format csv -d "$out/pmc_write" -- $B > /dev/null 2> "$out/write.err" || { echo "write pass failed"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-include-regex "$K" --output-format csv -d "$out/pmc_sq" -- $B > /dev/null 2> "$out/sq.err" || { echo "sq pass failed"; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ --kernel-include-regex "$K" --output-format csv -d "$out/pmc_tcc" -- $B > /dev/null 2> "$out/tcc.err" || { echo "tcc pass failed"; exit 1; }
cd "$root"
# (profiles/summarize.py runs in the build container afterwards, on the merged gpurun_out/prof_<tag>: it stamps the commit)
cp "$out/bench_under_rocprof.json" "$out/${tag}_bench_under_rocprof.json"
echo "profiles done"
