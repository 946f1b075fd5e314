"""Checks on the COMPILED device code that need no GPU (hipcc cross-compiles gfx950)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
UNMEASURED = ("conv0_zmarch.hip", "fnet_conv0_fused.hip", "deconv11_splitf16.hip", "deconv9_splitf16.hip", "conv11_prob_fused.hip")


@pytest.mark.skipif(not os.path.isfile(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_unmeasured_kernels_keep_floating_point_work_out_of_their_matrix_phases():
    """DESIGN.md 2.0, second hazard rule: floating-point vector work that does not depend on the matrix results, scheduled between a wave's own f16 matrix
    instructions, corrupted values at two workgroups per CU.  The emulation cannot see a schedule; tools/mfma_hazard_lint.py reads it from the compiler's
    assembly.  The kernels no GPU has run yet must have NO non-integer vector instruction inside a matrix phase (their sched_barriers hold)."""
    def lint(name):
        return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_hazard_lint.py"), os.path.join(ROOT, "casmvsnet_pl_amd", "csrc", name)],
                              capture_output=True, text=True, timeout=900)
    with ThreadPoolExecutor(max_workers=len(UNMEASURED)) as pool:
        results = list(pool.map(lint, UNMEASURED))
    for name, res in zip(UNMEASURED, results):
        assert res.returncode == 0 and "matrix instructions" in res.stdout and "FLAGGED" not in res.stdout, (name, res.stdout[-1500:], res.stderr[-500:])


@pytest.mark.skipif(not os.path.isfile(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")
def test_split_f16_kernels_do_not_spill_and_fit_two_waves_per_simd():
    """tools/kernel_resources.py (the compiler's metadata): every split-f16 kernel - the production ones and the five no GPU has run yet - without scratch
    and within 256 vector registers (two waves per SIMD = the two workgroups per CU their LDS sizes are chosen for; conv_ci_sf_kernel<32, 32> / <64, 64> own
    their CU: 512)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    files = UNMEASURED + ("conv0_splitf16.hip", "conv_ci_splitf16.hip", "conv2d_ci_splitf16.hip", "fpn_fused_sf.hip", "prob_regress.hip")
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
