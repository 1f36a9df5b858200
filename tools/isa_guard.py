#!/usr/bin/env python3
"""Build-time guard of pass2_kernel's register reservation (lmi_pass2.h: v[232:255] are written by inline-asm
`global_load_dwordx4` and read by inline-asm `v_mfma_f32_16x16x32_f16`, and kept away from hipcc only by
`__attribute__((amdgpu_num_vgpr(116)))`).  That attribute's meaning is not documented: this tool disassembles the
BUILT library and fails when the compiler's own code comes near the reserved registers.

For `pass2_kernel<true>` and `pass2_kernel<false>` of the gfx950 code object inside liblmi_hip.so:
  (a) the kernel descriptor says 256 VGPRs and 0 AGPRs;
  (b) every instruction that names v232..v255 is a `global_load_dwordx4` whose destination is a reserved set, or a
      `v_mfma_f32_16x16x32_f16` whose A operand is a reserved pair of registers (and whose other operands are not);
  (c) no `scratch_` instruction sits inside a tile body (a run of MFMAs less than 400 instructions apart).

  python3 tools/isa_guard.py [path/to/liblmi_hip.so]     exit code 0 = all checks pass
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_SO = os.path.join(ROOT, "learnedmetricindex_amd", "liblmi_hip.so")
KERNELS = {"pass2_kernel<true>": "_ZN3lmi12pass2_kernelILb1EEEvNS_15PrefilterParamsE",
           "pass2_kernel<false>": "_ZN3lmi12pass2_kernelILb0EEEvNS_15PrefilterParamsE"}
RES_LO, RES_HI = 232, 255
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def extract_code_object(so_path: str, workdir: str) -> str:
    fat = os.path.join(workdir, "fat.bin")
    co = os.path.join(workdir, "gfx950.co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", so_path, os.devnull])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"])
    return co


def kernel_notes(co: str) -> dict:
    """{mangled name: {".vgpr_count": int, ".agpr_count": int, ...}} from the code object's metadata note."""
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        s = line.strip()
        if s.startswith("- .agpr_count:") or s.startswith("- .args:"):
            cur = {}
            s = s[2:]
        if cur is None or ":" not in s:
            continue
        key, _, val = s.partition(":")
        val = val.strip()
        if key.strip() == ".name":
            out[val] = cur
        elif re.fullmatch(r"-?\d+", val):
            cur[key.strip()] = int(val)
    return out


def kernel_body(disasm_lines, mangled: str):
    start = None
    for i, l in enumerate(disasm_lines):
        if l.endswith(f"<{mangled}>:"):
            start = i + 1
        elif start is not None and re.match(r"^[0-9a-f]+ <.*>:$", l):
            return disasm_lines[start:i]
    assert start is not None, f"{mangled} not found in the disassembly"
    return disasm_lines[start:]


def regs_of(operand: str):
    """VGPR numbers an operand names (v7 -> {7}, v[232:235] -> {232..235})."""
    got = set()
    for m in _REG.finditer(operand):
        if m.group(1) is not None:
            got.add(int(m.group(1)))
        else:
            got.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return got


def check_kernel(name: str, body, notes: dict) -> list:
    errs = []
    if notes.get(".vgpr_count") != 256:
        errs.append(f"{name}: .vgpr_count = {notes.get('.vgpr_count')} (expected 256: 232 compiler + 24 reserved)")
    if notes.get(".agpr_count") != 0:
        errs.append(f"{name}: .agpr_count = {notes.get('.agpr_count')} (expected 0: no VGPR/AGPR split)")
    reserved = set(range(RES_LO, RES_HI + 1))
    n_load = n_mfma = 0
    mfma_at, scratch_at = [], []
    for i, raw in enumerate(body):
        ins = raw.split("//")[0].strip()
        if not ins:
            continue
        op, _, rest = ins.partition(" ")
        if op.startswith("v_mfma"):
            mfma_at.append(i)
        if op.startswith("scratch_"):
            scratch_at.append(i)
        ops = [o.strip() for o in rest.split(",")]
        touched = [regs_of(o) & reserved for o in ops]
        if not any(touched):
            continue
        if op == "global_load_dwordx4":
            dst = regs_of(ops[0])
            if dst and dst <= reserved and not any(touched[1:]):
                n_load += 1
                continue
        elif op == "v_mfma_f32_16x16x32_f16" and len(ops) >= 4:
            a = regs_of(ops[1])
            if a and a <= reserved and not touched[0] and not touched[2] and not touched[3]:
                n_mfma += 1
                continue
        errs.append(f"{name}: instruction outside the asm blocks names v{RES_LO}..v{RES_HI}: `{ins}`")
        if len(errs) > 20:
            break
    if n_load == 0 or n_mfma == 0:
        errs.append(f"{name}: expected reserved-register loads and MFMAs, found {n_load} / {n_mfma}")
    # tile bodies = runs of MFMAs closer than 400 instructions
    runs, s = [], None
    for a, b in zip(mfma_at, mfma_at[1:] + [None]):
        if s is None:
            s = a
        if b is None or b - a > 400:
            runs.append((s, a))
            s = None
    inside = [x for x in scratch_at if any(lo <= x <= hi for lo, hi in runs)]
    if inside:
        errs.append(f"{name}: {len(inside)} scratch_ instruction(s) inside a tile body, first: `{body[inside[0]].split('//')[0].strip()}`")
    return errs, {"reserved_loads": n_load, "reserved_mfmas": n_mfma, "mfma": len(mfma_at), "scratch": len(scratch_at), "tile_bodies": len(runs)}


def check(so_path: str = DEFAULT_SO):
    """-> (errors, per-kernel summary)"""
    with tempfile.TemporaryDirectory() as wd:
        co = extract_code_object(so_path, wd)
        notes = kernel_notes(co)
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout.splitlines()
    errors, summary = [], {}
    for name, mangled in KERNELS.items():
        if mangled not in notes:
            errors.append(f"{name}: no kernel descriptor for {mangled}")
            continue
        e, s = check_kernel(name, kernel_body(dis, mangled), notes[mangled])
        errors += e
        summary[name] = dict(s, vgpr_count=notes[mangled].get(".vgpr_count"), agpr_count=notes[mangled].get(".agpr_count"),
                             vgpr_spill_count=notes[mangled].get(".vgpr_spill_count"), sgpr_spill_count=notes[mangled].get(".sgpr_spill_count"))
    return errors, summary


if __name__ == "__main__":
    errs, summ = check(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_SO)
    for k, v in summ.items():
        print(k, v)
    for e in errs:
        print("FAIL", e)
    sys.exit(1 if errs else 0)
