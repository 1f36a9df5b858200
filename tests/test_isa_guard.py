"""CPU (`-m "not gpu"`): the register reservation of pass2_kernel holds in the BUILT library (tools/isa_guard.py).

pass2_kernel keeps its vector fragments in v[232:255], registers that only inline asm names and that hipcc is kept away from by
`amdgpu_num_vgpr(116)` -- undocumented behaviour a toolchain upgrade could change silently.  The guard disassembles the gfx950
code object inside liblmi_hip.so; csrc/build.sh runs it after every build as well."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_built_library_keeps_the_reserved_registers():
    import isa_guard
    from learnedmetricindex_amd import _capi

    assert os.path.exists(_capi.LIB_PATH), "build liblmi_hip.so first (__graft_entry__.build())"
    errors, summary = isa_guard.check(_capi.LIB_PATH)
    assert not errors, "\n".join(errors)
    for name in ("pass2_kernel<true>", "pass2_kernel<false>"):
        s = summary[name]
        assert s["vgpr_count"] == 256 and s["agpr_count"] == 0
        assert s["reserved_mfmas"] == s["mfma"] > 0      # every MFMA of the kernel reads its A operand from a reserved set
        assert s["reserved_loads"] > 0


def test_guard_flags_a_compiler_use_of_a_reserved_register():
    """The checker itself: a body in which the compiler touches v240, spills inside a tile body, or a descriptor with an
    AGPR split / fewer VGPRs must fail (what a build with the attribute edited to 128 would look like)."""
    import isa_guard

    good = (["\tglobal_load_dwordx4 v[232:235], v[4:5], off    // 0: x", "\tv_mfma_f32_16x16x32_f16 v[0:3], v[232:235], v[8:11], v[0:3]"] * 3)
    notes = {".vgpr_count": 256, ".agpr_count": 0}
    errs, s = isa_guard.check_kernel("k", good, notes)
    assert not errs and s["reserved_mfmas"] == 3
    # the compiler allocates a reserved register
    errs, _ = isa_guard.check_kernel("k", good + ["\tv_add_f32_e32 v240, v1, v2"], notes)
    assert any("v_add_f32_e32 v240" in e for e in errs)
    # a reserved register as an MFMA accumulator or B operand
    errs, _ = isa_guard.check_kernel("k", good + ["\tv_mfma_f32_16x16x32_f16 v[236:239], v[232:235], v[8:11], v[236:239]"], notes)
    assert errs
    # a spill reload of a reserved range
    errs, _ = isa_guard.check_kernel("k", good + ["\tscratch_load_dwordx4 v[252:255], off, off offset:16"], notes)
    assert errs
    # a spill between two MFMAs of one tile body
    body = [good[0], good[1], "\tscratch_store_dword off, v3, off offset:4", good[1]]
    errs, _ = isa_guard.check_kernel("k", body, notes)
    assert any("inside a tile body" in e for e in errs)
    # descriptor: the attribute read as a whole-file count (128 VGPRs + an AGPR half)
    errs, _ = isa_guard.check_kernel("k", good, {".vgpr_count": 128, ".agpr_count": 128})
    assert len(errs) == 2
