#!/usr/bin/env bash
# Dumps the gfx950 ISA + resource usage of the library's kernels into /tmp/lmi_isa (developer aid).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${1:-/tmp/lmi_isa}"
mkdir -p "$out" && cd "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I"${here}/../../include" -I"${here}" \
  ${LMI_EXTRA_FLAGS:-} -c "${here}/lmi_hip.hip" -o x.o -save-temps -Rpass-analysis=kernel-resource-usage 2> usage.txt || { cat usage.txt; exit 1; }
S=lmi_hip-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_ZN3lmi11scan_kernelENS_10ScanParamsE:/,/s_endpgm/' $S > scan.s
grep -A12 "Function Name: _ZN3lmi11scan_kernel" usage.txt | grep -E "VGPRs:|Spill|Occupancy|LDS Size|SGPRs:" | sed 's/remark: [^ ]* //'
echo "mfma $(grep -c v_mfma scan.s)  glds $(grep -c global_load_lds scan.s)  scratch $(grep -c scratch_ scan.s)  lines $(wc -l < scan.s)"
