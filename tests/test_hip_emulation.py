"""The kernels' own source on the CPU: tests/hipemu/hip/hip_runtime.h stands in for the HIP runtime (one std::thread per GPU thread, wave collectives -
the f16 and float32 MFMAs with the lane layouts the MI355X self-tests verified, DPP exchanges and wave shifts, ballots, readlane - as rendezvous of a wave's
64 threads, raw buffer addressing with range checking, LDS / global atomics), the drivers include the .hip files as C++ and run them against float64:

  run_kernels   conv0 split-f16 (tiled: validates the emulator; z-march, shifted grids), FeatureNet.conv0 fused, deconv9 / deconv11 split-f16
  run_kernels2  conv_ci_sf / conv2d_ci_sf (production)                 run_kernels3  conv_s2_sf: the stride-2 layers on the f16 cores (round 4)
  run_kernels4  prob z-walk head (production)                         run_kernels5  prob weight gradient, both fusion kernels (production)
  run_kernels6  FPN tail split-f16 (production)                       run_kernels7  LDS-staged plane sweep / variance volume (production)
  run_kernels8  training: conv_wgrad (all kinds, both LDS layouts), channel sums, variance-volume backward
  run_kernels9  the float32 matrix-core layers of conv3d_mfma.hip (3D stride 1 / 2 / transposed + skip, 2D k3 / k5 s2)
  run_kernels10 conv11 + prob + softmax regression as one depth-walking kernel (round 4; 512-thread workgroups)
  run_kernels11 conv2d_k5s2_sf: FeatureNet's stride-2 layers (conv1.0 / conv2.0) on the f16 cores (round 4)
  run_kernels12 fnet_conv0_mm: FeatureNet.conv0 (conv0.0 + conv0.1) as one kernel, both layers on the f16 cores (round 6)

The kernels written at the end of round 3 without access to a GPU RUN here for the first time - ragged shapes, persistent workgroups that walk several
items, z segments.  The same sources under ThreadSanitizer (a missing barrier is a reported race) and under an LDS bank-conflict / cache-line profile built
from the compiler's memory-access hooks (tools/lds_bank_profile.py), whose conflict ratios match the MI355X's counters.  What the emulation cannot show:
timing, the co-residency hazard of DESIGN.md 2.0 (tools/mfma_hazard_lint.py reads the schedules for it), the device compiler's code.  CPU only; needs
ROCm's clang++ (host target)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get("HIPEMU_CXX") or "/opt/rocm/lib/llvm/bin/clang++"


DRIVERS = ("run_kernels", "run_kernels2", "run_kernels3", "run_kernels4", "run_kernels5", "run_kernels6", "run_kernels7", "run_kernels8", "run_kernels9", "run_kernels10", "run_kernels11", "run_kernels12")
PROFILED = ("run_kernels", "run_kernels3", "run_kernels11", "run_kernels12")
# costvol_lds.hip instantiates 30 kernels: its ThreadSanitizer build alone takes 80 s - part of the suite only with HIPEMU_FULL=1 (clean when it was added)
# (run_kernels8: the weight-gradient cases take a minute under the sanitizer; run_kernels9: conv3d_mfma.hip is 2500 lines of templates - clean when added)
TSAN_DRIVERS = DRIVERS if os.environ.get("HIPEMU_FULL") == "1" else tuple(d for d in DRIVERS if d not in ("run_kernels7", "run_kernels8", "run_kernels9"))


def _profile_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_bank_profile", os.path.join(ROOT, "tools", "lds_bank_profile.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    return tool


def _compile(workdir, source, extra=(), suffix=""):
    exe, obj = os.path.join(workdir, source + suffix), os.path.join(workdir, source + suffix + ".o")
    build = subprocess.run([CLANG, "-std=c++20", "-O1", "-pthread", *extra, "-DCASMVS_SPLIT_NOASM", "-I" + os.path.join(ROOT, "tests", "hipemu"),
                            "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "casmvsnet_pl_amd", "csrc"), "-x", "c++", "-c",
                            os.path.join(ROOT, "tests", "hipemu", source + ".cpp"), "-o", obj], capture_output=True, text=True, timeout=900)
    assert build.returncode == 0, build.stderr[-3000:]
    stubs = _profile_tool().link_stubs(obj, workdir, source + suffix)   # (run_kernels9: the kernel families conv3d_mfma.hip's engine functions call)
    link = subprocess.run([CLANG, "-pthread", *[e for e in extra if e.startswith("-fsanitize")], obj, *stubs, "-o", exe], capture_output=True, text=True, timeout=300)
    assert link.returncode == 0, link.stderr[-3000:]
    return exe


def _has_tsan(workdir):
    src = os.path.join(workdir, "t.cpp")
    with open(src, "w") as f:
        f.write("int main() { return 0; }\n")
    return subprocess.run([CLANG, "-fsanitize=thread", src, "-o", os.path.join(workdir, "t")], capture_output=True).returncode == 0


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    """Every executable of this module - plain, ThreadSanitizer and LDS-profile builds of the five drivers - compiled concurrently, once."""
    from concurrent.futures import ThreadPoolExecutor
    workdir = str(tmp_path_factory.mktemp("hipemu"))
    tsan = _has_tsan(workdir)
    tool = _profile_tool()
    jobs = {}
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        for source in DRIVERS:
            jobs[(source, "plain")] = pool.submit(_compile, workdir, source)
            if tsan and source in TSAN_DRIVERS:
                jobs[(source, "tsan")] = pool.submit(_compile, workdir, source, ("-g", "-fsanitize=thread"), "_tsan")
        if tsan:
            for source in PROFILED:
                jobs[(source, "profile")] = pool.submit(tool.build, source, workdir)
    return {"workdir": workdir, "tsan": tsan, **{k: f.result() for k, f in jobs.items()}}


def _run(exe, names, mode="quick"):
    env = dict(os.environ, TSAN_OPTIONS="exitcode=0")   # the sanitizer test judges the reports itself
    out = subprocess.run([exe, mode], capture_output=True, text=True, timeout=1800, env=env)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-2000:] + out.stderr[-1000:]
    for name in names:
        assert name in out.stdout
    return out


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_kernels_run_on_the_cpu_against_float64(built):
    _run(built[("run_kernels", "plain")], ("conv0_sf", "conv0_zm", "conv0_zw", "deconv11", "deconv9"))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_production_channel_inner_kernels_run_on_the_cpu(built):
    """conv_ci_sf_kernel (CostRegNet conv2 / conv4 / conv6) and conv2d_ci_sf_kernel (FeatureNet, with its pixel-major second output): the device code of
    two GPU-validated production kernels as a regression test that needs no GPU."""
    _run(built[("run_kernels2", "plain")], ("conv_ci", "conv2d_ci"))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_stride2_z_march_kernel_runs_on_the_cpu(built):
    """conv_s2_sf_kernel (CostRegNet conv1 / conv3 on the f16 matrix cores, input-stationary along z; written in round 4 with this run as its first test):
    8 -> 16 and 16 -> 32 on ragged volumes against the layer in float64 (the driver's `all` mode: odd sizes along every axis, three z segments, rows of 4)."""
    _run(built[("run_kernels3", "plain")], ("conv_s2    8 -> 16", "conv_s2    16 -> 32"))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_fused_regulariser_tail_runs_on_the_cpu(built):
    """conv11_prob_zfused_kernel (conv11 + skip, `prob`, softmax regression walking the depth axis; written in round 4 with this run as its first test): cost,
    depth, confidence and index against the three layers in float64 - image borders inside a tile, two tiles in x (the driver's `all` mode: several tiles in
    y, an odd number of input planes, the smallest volume)."""
    _run(built[("run_kernels10", "plain")], ("conv11_prob_zfused",))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_featurenet_stride2_kernel_runs_on_the_cpu(built):
    """conv2d_k5s2_sf_kernel (FeatureNet conv1.0 8 -> 16 and conv2.0 16 -> 32, Conv2d k5 s2 p2 on the f16 matrix cores; written in round 4 with this run as
    its first test): against the layer in float64 - borders inside a tile, two tiles in y and x, one and two channel chunks (the driver's `all` mode: three
    images, 6x8 and 2x4 images, persistent workgroups walking several units)."""
    _run(built[("run_kernels11", "plain")], ("conv2d_k5s2 8 -> 16", "conv2d_k5s2 16 -> 32"))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_fused_featurenet_conv0_runs_on_the_cpu(built):
    """fnet_conv0_mm_kernel (FeatureNet.conv0 = ConvBnReLU 3 -> 8 -> 8 as ONE kernel, both layers on the f16 matrix cores, the 8-channel map between them in
    LDS only; round 6): against the two layers in float64 - borders inside a tile, two tiles in y, three in x (the last one 4 pixels wide), images smaller
    than a tile (the driver's `all` mode: three images of 44 x 92, a 2 x 2 image, amplitudes 1e-20 and 3e4)."""
    _run(built[("run_kernels12", "plain")], ("fnet_conv0_mm N=1 22x64", "fnet_conv0_mm N=2 6x8"))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_production_prob_head_runs_on_the_cpu(built):
    """prob_zwalk_kernel (Conv3d 8 -> 1 walking the depth axis, regression fused or chunked; the production head): cost, depth, confidence and index against
    float64 - a GPU-free regression test of the kernel the fused tail was derived from."""
    _run(built[("run_kernels4", "plain")], ("prob_zwalk",))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_prob_weight_gradient_and_fusion_kernels_run_on_the_cpu(built):
    """prob_wgrad_kernel + its reduction against a float64 loop; fuse_view_paired_kernel against fuse_view_kernel, all eight outputs bit-equal."""
    _run(built[("run_kernels5", "plain")], ("prob_wgrad", "fusion"))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_production_fpn_tail_runs_on_the_cpu(built):
    """fpn_tail0_sf_kernel (FeatureNet's full-resolution tail on the f16 matrix cores, the bilinear interpolation inside its staging; 0.45 ms of the step)
    against lat0 / upsample-add / smooth0 in float64 at the GPU test's bound, both output layouts."""
    _run(built[("run_kernels6", "plain")], ("fpn_tail0",))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_production_plane_sweep_runs_on_the_cpu(built):
    """costvol_lds_kernel (the fused homo_warp + variance cost volume with the source boxes staged in LDS: the second largest kernel of the step) for
    C = 8 and C = 16 on ragged tiles against mvsnet.py:147-167 per voxel, tap positions from the shared float32 routine, sums in float64."""
    _run(built[("run_kernels7", "plain")], ("costvol_lds",))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_training_weight_gradient_kernel_runs_on_the_cpu(built):
    """conv_wgrad_kernel (csrc/train.hip: the weight gradient of every convolution kind as a GEMM over the positions on v_mfma_f32_16x16x4_f32 + the
    fixed-order reduction) for Conv3d s1 and Conv2d k5 s2 (every kind in the driver's `all` mode) against the definition in float64, twice (the fixed-order
    reduction must reproduce the bits); channel_sums_kernel against float64 sums; train-mode ABN with the per-channel epilogue inside the elementwise kernels
    (abn_train_apply_kernel / abn_bwd_apply_stats_kernel: values, constants, running statistics, gradients; 16-byte and scalar paths); costvol_var_bwd_kernel (the scatter transpose of the plane sweep through a
    64-bit fixed-point LDS image, ds_add_u64) against the derivative of the variance through the bilinear weights in float64 - also with source views 1e4 x the
    reference view, where the workgroups leave the fixed-point range and repeat the pass with float atomics (`all`: zero reference features, 1e-30 gradients),
    and its group-wise correlation form."""
    out = _run(built[("run_kernels8", "plain")], ("wgrad S1", "wgrad K5S2", "channel_sums", "pack_gather_batch", "abn_fused", "var_backward", "gwc_backward"))   # (`all`: every kind, profiles/r03_hip_emulation_all.txt)
    assert "DIFFERENT" not in out.stdout and out.stdout.count("bit-identical") >= 2


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
def test_float32_matrix_core_layers_run_on_the_cpu(built):
    """csrc/conv3d_mfma.hip through casmvs_conv3d_forward_f32: Conv3d k3 s2 (conv1) and ConvTranspose3d k3 s2 + skip (conv11) on v_mfma_f32_16x16x4_f32
    against the layers in float64 (the driver's `all` mode: also the stride-1 PX / CI forms, conv3, conv9 and the one-channel tile kernel)."""
    _run(built[("run_kernels9", "plain")], ("conv3d_f32 S2", "conv3d_f32 T2"))


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
@pytest.mark.parametrize("source,names", [("run_kernels", ("conv0_zm", "conv0_zw", "deconv11", "deconv9")), ("run_kernels2", ("conv_ci", "conv2d_ci")), ("run_kernels3", ("conv_s2",)), ("run_kernels10", ("conv11_prob_zfused",)), ("run_kernels11", ("conv2d_k5s2",)), ("run_kernels12", ("fnet_conv0_mm",)),
                                          ("run_kernels4", ("prob_zwalk",)), ("run_kernels5", ("prob_wgrad", "fusion")), ("run_kernels6", ("fpn_tail0",)), ("run_kernels7", ("costvol_lds",)), ("run_kernels8", ("wgrad",)), ("run_kernels9", ("conv3d_f32",))])
def test_no_lds_race_under_thread_sanitizer(built, source, names):
    """A missing __syncthreads() rarely shows in the results of an emulated run (the threads happen to be scheduled kindly): ThreadSanitizer sees it anyway.
    LDS is plain memory shared by the workgroup's std::threads and the barrier is the only synchronisation between waves (the wave collectives synchronise
    one wave's 64 threads, as the hardware's lock step does), so a write and a read of the same LDS word by different waves without a barrier between them is
    reported as a data race - as are two workgroups storing to the same output element.  Checked to work in round 3: the (since removed) fused tail kernel with its
    slot-release barrier removed passed the value check and produced 64 reports."""
    if not built["tsan"]:
        pytest.skip("this clang++ has no ThreadSanitizer runtime")
    if source not in TSAN_DRIVERS:
        pytest.skip("over a minute of compile / run time: HIPEMU_FULL=1")
    out = _run(built[(source, "tsan")], names)
    reports = out.stderr.split("WARNING: ThreadSanitizer")[1:]
    # the one intended same-address access: prob_wgrad_kernel's staging rounds past the last item all WRITE the dummy word box[DUMMY], which nobody reads
    benign = [r for r in reports if "Write of size 4" in r and "Previous write of size 4" in r and
              [ln.split("prob_wgrad.hip:")[1].split(":")[0] for ln in r.splitlines() if ln.lstrip().startswith("#0 ") and "prob_wgrad.hip:" in ln] == ["99", "99"]]
    assert len(reports) == len(benign), reports[0][:3000]


@pytest.mark.skipif(not (os.path.isfile(CLANG) or shutil.which(CLANG)), reason="needs a clang++ with ext_vector_type / _Float16 (ROCm's)")
@pytest.mark.parametrize("source", ["run_kernels", "run_kernels3", "run_kernels11", "run_kernels12"])
def test_lds_bank_profile_of_the_split_f16_kernels(built, source):
    """tools/lds_bank_profile.py: the compiler's memory-access hooks (-fsanitize=thread, linked against tests/hipemu/lds_profile.cpp instead of the sanitizer)
    record every LDS access of the emulated run; the accesses of a wave are regrouped into wave-instructions and priced with the bank rules of
    MI355X_MICROARCH.md.  The model reproduces what the GPU's counters said about the tuned production kernels (prob_zwalk_kernel, conv_ci_sf_kernel,
    conv2d_ci_sf_kernel: conflict-free, profiles/r03_lds_bank_model.txt).  Asserted for the kernels no GPU has timed yet: the operand reads of their matrix
    phases are conflict-free, and the staging stores stay within 1.4x of their floor (a 16-byte store costs 13 cycles of register transfer anyway)."""
    if not built["tsan"]:   # the instrumentation pass comes with the same option
        pytest.skip("this clang++ has no -fsanitize=thread")
    tool = _profile_tool()
    totals = tool.per_kernel(tool.profile(source, "quick", workdir=built["workdir"], exe=built[(source, "profile")]))
    want = {"run_kernels": ("conv0_sf_kernel", "conv0_zw_kernel", "deconv11_sf_kernel", "deconv9_sf_kernel"), "run_kernels3": ("conv_s2_sf_kernel",), "run_kernels11": ("conv2d_k5s2_sf_kernel",), "run_kernels12": ("fnet_conv0_mm_kernel",)}[source]
    for name in want:
        kernels = [k for k in totals if k.startswith(name)]
        assert kernels, (name, list(totals))
        for k in kernels:
            n, cyc, ideal, _, _ = totals[k]["R"]
            assert n > 0 and cyc <= 1.0 * ideal, (k, "reads", cyc, ideal)
            n, _, _, eff, eff_ideal = totals[k]["W"]
            # fnet_conv0_mm_kernel: layer 1's results leave the accumulators as one dword (a channel pair) per pixel and lane - pixels 32 bytes apart, so
            # lanes j, j + 4, j + 8, j + 12 share a bank (4-way, 24 stores per lane and tile: ~3 % of a tile's time; the unit layout is what keeps layer 2's
            # operand READS conflict-free, which is what matters)
            assert n > 0 and eff <= (1.6 if name == "fnet_conv0_mm_kernel" else 1.4) * eff_ideal, (k, "writes", eff, eff_ideal)
    if source == "run_kernels":   # the model against the hardware: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of conv0_sf_kernel<8, 3> on the MI355X = 0.184
        t = totals[next(k for k in totals if k.startswith("conv0_sf_kernel<8"))]   # (profiles/r03_pmc_conv0_split_kernels.txt)
        cycles, ideal = t["R"][1] + t["W"][1], t["R"][2] + t["W"][2]
        assert 0.17 < (cycles - ideal) / cycles < 0.20, (cycles, ideal)
