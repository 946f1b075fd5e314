"""Checks on the COMPILED device code that need no GPU (hipcc cross-compiles gfx950)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
STRICT_FILES = ("conv0_zmarch.hip", "deconv11_splitf16.hip", "deconv9_splitf16.hip", "conv_s2_splitf16.hip", "conv11_prob_zfused.hip", "conv2d_k5s2_splitf16.hip", "fnet_conv0_mm.hip")   # kernels written with NO floating-point work inside their matrix phases


def _lint_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_hazard_lint", os.path.join(ROOT, "tools", "mfma_hazard_lint.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    return tool


@pytest.mark.skipif(not os.path.isfile(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_floating_point_work_inside_matrix_phases_is_pinned_for_every_f16_kernel():
    """DESIGN.md 2.0, second hazard rule: floating-point vector work that does not depend on the matrix results, scheduled between a wave's own f16 matrix
    instructions, corrupted values at two workgroups per CU.  No functional test on the CPU can see a schedule; tools/mfma_hazard_lint.py reads it from the
    compiler's assembly.  For EVERY kernel of the production library with f16 / bf16 matrix instructions the table {opcode: count} of floating-point vector
    instructions inside its matrix phases is pinned by tests/golden/mfma_phase_fp_instructions.json - the schedules that
    test_split_f16_kernels_are_bit_stable_at_full_occupancy validated on the MI355X.  A compiler update or an edit that moves one more floating-point instruction
    into a phase fails here, on the CPU box; the way out is to re-run the bit-stability test on the GPU and regenerate the file
    (python tools/mfma_hazard_lint.py --json > tests/golden/mfma_phase_fp_instructions.json).  The kernels of STRICT_FILES must have none at all."""
    import json
    tool = _lint_tool()
    got = tool.table()
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "mfma_phase_fp_instructions.json")))
    assert set(got) == set(want), (sorted(set(got) ^ set(want)), "kernel set changed: regenerate the golden file after the GPU bit-stability test")
    worse = {k: (want[k], got[k]) for k in got if any(n > want[k].get(op, 0) for op, n in got[k].items())}
    assert not worse, worse
    for name, flagged in got.items():
        if name.startswith(tool.STRICT):
            assert not flagged, (name, flagged)
    assert sum(1 for name in got if name.startswith(tool.STRICT)) >= 14


@pytest.mark.skipif(not os.path.isfile(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_split_f16_kernels_do_not_spill_and_fit_two_waves_per_simd():
    """tools/kernel_resources.py (the compiler's metadata): every split-f16 kernel without scratch
    and within 256 vector registers (two waves per SIMD = the two workgroups per CU their LDS sizes are chosen for; conv_ci_sf_kernel<32, 32> / <64, 64> own
    their CU: 512)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    files = STRICT_FILES + ("conv0_splitf16.hip", "conv_ci_splitf16.hip", "conv2d_ci_splitf16.hip", "fpn_fused_sf.hip", "prob_regress.hip")
    with ThreadPoolExecutor(max_workers=8) as pool:
        results = list(pool.map(lambda f: tool.resources(os.path.join(ROOT, "casmvsnet_pl_amd", "csrc", f)), files))
    seen = 0
    for f, rows in zip(files, results):
        for name, vgpr, sgpr, scratch, lds in rows:
            if "probe" in name:
                continue
            seen += 1
            assert scratch == 0, (f, name, scratch)
            one_per_cu = name.startswith(("conv_ci_sf_kernel<32, 32", "conv_ci_sf_kernel<64, 64"))   # their lane images leave room for one workgroup
            assert vgpr <= (512 if one_per_cu else 256), (f, name, vgpr)
    assert seen >= 30


def test_no_wide_store_is_followed_by_a_write_of_its_data_registers():
    """tools/store_hazard_lint.py over every kernel of the library: no vector-memory store of 12 / 16 bytes per lane has a VALU write of one of its data
    registers within the next two issue slots.  The MI355X stores the NEW value in some lanes then (round 5: a depth hypothesis in channels 12-15 of the
    16-plane homo_warp kernel, different lanes every run); the ISA lists the pair as needing wait states, and LLVM pads it except where the store's soffset
    operand is a scalar register - the form of the plane sweep's volume stores, which therefore carry their own `s_nop 1` tied to the data registers
    (csrc/costvol_lds.hip: store_plane_transposed).  Whether a kernel is hit is a matter of register allocation: this pins it for every build."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("store_hazard_lint", os.path.join(ROOT, "tools", "store_hazard_lint.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    report = tool.lint(window=2)
    assert not report, report
    # the scan follows control flow (round-5 advisor finding: it used to stop at any label or branch): a wide store at the end of a loop body is checked against
    # the head of the loop its back edge lands on, an unconditional branch against its target, a label is fallen through; s_nop N pads N + 1 slots
    st = "buffer_store_dwordx4 v[4:7], v9, s[0:3], s8 offen"
    assert not tool.find(["LABEL .L0", "v_mov_b32 v1, v2", st, "s_cbranch_scc1 .L0", "s_endpgm"], 2)
    assert tool.find(["LABEL .L0", "v_mov_b32 v5, v2", st, "s_cbranch_scc1 .L0", "s_endpgm"], 2)[0][2] == "v_mov_b32 v5, v2"      # through the back edge
    assert tool.find([st, "s_branch .L1", "v_mov_b32 v4, 0", "LABEL .L1", "v_mov_b32 v6, 0"], 2)[0][2] == "v_mov_b32 v6, 0"          # at the branch target only
    assert tool.find([st, "LABEL .L1", "v_mov_b32 v6, 0"], 2) and not tool.find([st, "s_nop 1", "LABEL .L1", "v_mov_b32 v6, 0"], 2)   # labels take no slot
    assert tool.find([st, "s_cbranch_vccz .L1", "s_nop 0", "LABEL .L1", "v_mov_b32 v7, 0"], 2)                                      # taken side of a conditional branch
    assert not tool.find([st, "s_cbranch_vccz .L1", "s_nop 0", "v_mov_b32 v7, 0", "LABEL .L1", "s_endpgm"], 2)                       # 3 slots on the fall-through side


def test_packed_float32_rewrite_exchanges_the_sources_of_the_faulty_class_only():
    """casmvsnet_pl_amd/build.py rewrite_unsafe_packed on assembly lines: the faulty class of the MI355X (tools/probes/pk_fma_opsel_repro.hip: low result from
    src0's low half and a VECTOR src1's high half) gets src0 / src1 and every per-source modifier bit exchanged - the same arithmetic, a clean form; everything
    else passes through untouched."""
    sys.path.insert(0, ROOT)
    from casmvsnet_pl_amd import build
    cases = {
        "\tv_pk_fma_f32 v[12:13], v[14:15], v[92:93], v[94:95] op_sel:[0,1,1]": "\tv_pk_fma_f32 v[12:13], v[92:93], v[14:15], v[94:95] op_sel:[1,0,1]",
        "\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]": "\tv_pk_mul_f32 v[2:3], v[6:7], v[4:5] op_sel:[1,0] op_sel_hi:[0,1]",
        "\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]": "\tv_pk_add_f32 v[2:3], v[6:7], v[4:5] op_sel:[1,0]",
        "\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0] clamp":
            "\tv_pk_fma_f32 v[2:3], v[6:7], v[4:5], v[8:9] op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0] clamp",
    }
    untouched = ["\tv_pk_fma_f32 v[0:1], v[0:1], v[2:3], 0.5 op_sel_hi:[1,1,0]", "\tv_pk_mul_f32 v[20:21], s[74:75], v[12:13]",
                 "\tv_pk_fma_f32 v[2:3], v[4:5], s[6:7], v[8:9] op_sel:[0,1,0]",            # src1 in scalar registers: measured clean
                 "\tv_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,0] op_sel_hi:[0,1]", "\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[1,1,1]",
                 "\tv_pk_mov_b32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]", "\tv_fma_mix_f32 v1, v2, v3, -v4 op_sel:[0,1,0] op_sel_hi:[0,1,0]", "\ts_nop 0"]
    for before, after in cases.items():
        assert build.packed_f32_is_unsafe(before) and not build.packed_f32_is_unsafe(after)
        assert build.rewrite_unsafe_packed(before) == (after, 1)
    for line in untouched:
        assert not build.packed_f32_is_unsafe(line)
        assert build.rewrite_unsafe_packed(line) == (line, 0)
    text = "\n".join(list(cases) + untouched)
    fixed, n = build.rewrite_unsafe_packed(text)
    assert n == len(cases) and fixed == "\n".join(list(cases.values()) + untouched)


@pytest.mark.skipif(not os.path.isfile(HIPCC), reason="needs hipcc and llvm-objdump")
def test_the_built_library_contains_no_packed_float32_instruction_of_the_faulty_class():
    """The SHIPPED artefact: the gfx950 code objects inside casmvsnet_pl_amd/libcasmvs_hip.so (built here if it is not), disassembled - thousands of packed
    float32 instructions, none with op_sel:[0,1,..] on a vector src1.  That is what lets kernels with f16 matrix instructions run beside the float32
    kernels of other streams (casmvsnet_pl_amd/streams.py: the guard is off for a library that says casmvs_packed_opsel_safe() == 1)."""
    import importlib.util
    sys.path.insert(0, ROOT)
    from casmvsnet_pl_amd import build
    lib = build.build_library()
    spec = importlib.util.spec_from_file_location("packed_opsel_lint", os.path.join(ROOT, "tools", "packed_opsel_lint.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    objects, packed, unsafe = tool.lint_library(lib)
    assert objects >= 20 and packed > 10000, (objects, packed)
    assert not unsafe, unsafe[:10]
