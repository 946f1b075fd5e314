"""GPU parity tests (run with -m gpu on a real MI355X): every HIP kernel, called through the C ABI,
against the oracle (oracle/cpu_restatement.py, pinned to the real reference by tests/test_oracle.py)
on the same seeded inputs, plus end-to-end parity against the committed golden fixtures.

Tolerances.  north_star's bar: depth within 1e-3 relative, depth index = clamp(trunc(sum_k p_k k)) identical.
What the tests assert is tighter - at most ~10x what was MEASURED on the MI355X (profiles/*parity_report.json),
so that a regression of one order of magnitude fails although it would still meet the bar:
  depth 1e-4 relative (measured 3e-6..8e-6); homo_warp 1.5e-4 abs (1.3e-5); cost volume 5e-5 of its range
  (4.6e-6); conv layers 1.2e-5 of the range (1.1e-6); hypotheses 3e-6 rel (2.7e-7); confidence 5e-4 abs on
  pixels with equal index (4.5e-5).
Depth index: every pixel whose index differs from the oracle's must have an expected index e = sum_k p_k k
within 1e-3 of an integer - trunc() is discontinuous there and the oracle's own 1-thread vs 8-thread runs
already differ by 1e-5 in e - and each such pixel is listed in the parity report with its
distance to the boundary.  Anywhere else a different index fails the test.  The end-to-end tests also bound the
NOISE of e itself (max |e_gpu - e_oracle| over all pixels, from the engine's own cost volume in float64) and the
cost volumes' error: a flip can only happen within that noise of a boundary.  One workload is ill-conditioned by
construction - group-wise correlation with G = 8 has ONE channel per group at level 0, its softmax is one-hot
almost everywhere (96 000 of 327 680 pixels within 1e-3 of an integer e) and amplifies a 1e-6 cost error to 2.5e-3
in e: there the boundary distance is bounded by 1e-2 and the cost error carries the parity claim.
The three cost-volume kernel families (NCHW gather, pixel-major gather, LDS-staged) must agree BIT FOR BIT.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_restatement as R
from util import GOLDEN_CASES, Golden, max_abs, rel_err, scaled_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _ops():
    from casmvsnet_pl_amd import ops
    return ops


def test_mfma_lane_mapping_selftest(report):
    rc, dump, msg = _ops().selftest_mfma()
    report("mfma_selftest", rc=rc, msg=msg, abid5_reg0=dump[1, 0, :8].tolist(), nobcast_reg0=dump[3, 0, :8].tolist())
    assert rc == 0, msg


def _proj_like(B, V, H, W, seed, geometry="dtu", level=0):
    from casmvsnet_pl_amd.synthetic import make_inputs
    # make_inputs builds level matrices for a (H*2**level, W*2**level) image
    _, proj, dmin, dint = make_inputs(B, V, H * 2 ** level, W * 2 ** level, seed=seed, geometry=geometry)
    return proj[:, :, level].contiguous(), dmin, dint


@pytest.mark.parametrize("B,C,H,W,D,geometry", [(1, 8, 32, 40, 8, "dtu"), (2, 32, 24, 56, 5, "dtu"),
                                                (1, 3, 17, 23, 4, "random"), (1, 16, 64, 80, 48, "dtu"),
                                                (2, 32, 40, 72, 16, "dtu"), (1, 8, 20, 30, 8, "random")])
def test_homo_warp_matches_oracle(dev, report, B, C, H, W, D, geometry):
    g = torch.Generator().manual_seed(B * 1000 + C * 10 + D)
    src = torch.randn(B, C, H, W, generator=g)
    proj, dmin, dint = _proj_like(B, 2, H, W, seed=C, geometry=geometry)
    proj = proj[:, 0]
    depth = dmin + torch.rand(B, D, H, W, generator=g) * 500.0
    if geometry == "random":
        depth[:, 0] = 1e-9      # tiny positive depth: huge coordinates
        depth[:, 1] = -50.0     # negative depth
    want = R.homo_warp(src, proj, depth)
    got = _ops().homo_warp(src.to(dev), proj.to(dev), depth.to(dev), impl="gather").cpu()
    assert torch.isfinite(got).all()
    err = max_abs(got, want)
    report("homo_warp", shape=[B, C, H, W, D], geometry=geometry, max_abs=err, nonzero_frac=float((want != 0).float().mean()))
    assert err < 1.5e-4  # measured 1.3e-5: bilinear is continuous, |d value| <= |grad| * coordinate noise (~1e-5 px)
    assert float(((got != 0) != (want != 0)).float().mean()) < 1e-3  # same in/out-of-bounds pattern
    from casmvsnet_pl_amd import _lib
    if C in (8, 16, 32) and _lib.load().casmvs_homo_warp_lds_supported(C, W, D):  # the LDS-staged forms: same bits
        assert torch.equal(_ops().homo_warp(src.to(dev), proj.to(dev), depth.to(dev), impl="lds").cpu(), got)        # box staged from the channel planes
        assert torch.equal(_ops().homo_warp(src.to(dev), proj.to(dev), depth.to(dev), impl="lds_copy").cpu(), got)   # pixel-major copy + the same sweep


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), float("-inf")])
def test_homo_warp_degenerate_x_row_gives_zeros_in_every_implementation(dev, bad):
    """A projection matrix whose x row is NaN / inf while y and z stay finite (round-5 advisor finding): bounds-checked taps contribute nothing, the warped
    volume is 0.  The LDS form that stages its box from the channel planes rounds the box down to a quad of x (bx0 down to -4): its NaN clamp must stay left
    of THAT box, else the lane reads a staged column with NaN weights."""
    from casmvsnet_pl_amd import _lib
    g = torch.Generator().manual_seed(5)
    B, C, H, W, D = 2, 16, 32, 48, 8
    src = torch.randn(B, C, H, W, generator=g)
    proj, dmin, dint = _proj_like(B, 2, H, W, seed=3)
    proj = proj[:, 0].clone()
    depth = dmin + torch.rand(B, D, H, W, generator=g) * 300.0
    proj[1, 0, 3] = bad          # sample 1: x = R p + T / d is NaN / inf everywhere, y finite and inside the image
    # (torch's CPU grid_sample, the oracle, propagates the NaN weights of such a sample; its GPU kernel - what the reference runs on - bounds-checks every
    # tap on the truncated integer and writes 0, and so do this library's kernels: the sample is compared between implementations, not with the oracle)
    want = R.homo_warp(src[:1], proj[:1], depth[:1])
    got = _ops().homo_warp(src.to(dev), proj.to(dev), depth.to(dev), impl="gather").cpu()
    assert not got[1].any() and want.any() and max_abs(got[:1], want) < 1.5e-4
    assert _lib.load().casmvs_homo_warp_lds_supported(C, W, D)
    for impl in ("lds", "lds_copy"):
        assert torch.equal(_ops().homo_warp(src.to(dev), proj.to(dev), depth.to(dev), impl=impl).cpu(), got), impl


@pytest.mark.parametrize("B,V,C,G,h,w,D,geometry", [
    (1, 3, 8, 1, 32, 40, 8, "dtu"), (1, 3, 16, 1, 32, 48, 32, "dtu"), (2, 5, 32, 1, 16, 24, 48, "dtu"),
    (1, 2, 4, 1, 20, 28, 3, "dtu"), (1, 3, 6, 1, 20, 28, 3, "random"), (1, 7, 8, 1, 24, 32, 8, "dtu"),
    (1, 3, 32, 8, 16, 24, 48, "dtu"), (1, 3, 16, 8, 32, 48, 32, "dtu"), (1, 3, 8, 8, 32, 40, 8, "dtu"),
    (2, 4, 16, 4, 16, 24, 8, "random"), (1, 3, 32, 2, 16, 24, 8, "dtu"),
    # LDS-staged kernel: partial tiles (w % 32 != 0, h % 8 != 0), w % 4 != 0 (scalar stores), 64-wide tiles, 7 views
    (1, 3, 16, 1, 44, 72, 16, "dtu"), (1, 3, 8, 1, 21, 30, 8, "dtu"), (2, 3, 16, 1, 16, 128, 8, "dtu"),
    (1, 7, 8, 1, 24, 64, 8, "dtu"), (1, 9, 8, 1, 16, 32, 8, "dtu"), (1, 3, 32, 1, 24, 40, 8, "random"),
    (1, 5, 16, 8, 24, 64, 16, "dtu")])
def test_costvol_matches_oracle(dev, report, B, V, C, G, h, w, D, geometry):
    g = torch.Generator().manual_seed(V * 100 + C + G)
    feats = torch.randn(B, V, C, h, w, generator=g)
    proj, dmin, dint = _proj_like(B, V, h, w, seed=V + C, geometry=geometry)
    depth = dmin + torch.rand(B, 1, h, w, generator=g) * 300.0 + torch.arange(D).view(1, D, 1, 1) * dint * 2
    want = R.cost_volume(feats, proj, depth, G)
    got = _ops().costvol(feats.to(dev), proj.to(dev), depth.to(dev), G).cpu()
    err = max_abs(got, want)
    report("costvol", shape=[B, V, C, G, h, w, D], geometry=geometry, max_abs=err, ref_absmax=float(want.abs().max()))
    assert got.shape == want.shape and torch.isfinite(got).all()
    assert err < 5e-5 * max(1.0, float(want.abs().max()))  # measured 4.6e-6
    if C in (8, 16, 32):  # the channel-last kernels do the same arithmetic in the same order: bit-identical
        nhwc = _ops().nchw_to_nhwc(feats.reshape(B * V, C, h, w).to(dev))
        assert torch.equal(nhwc.cpu(), feats.reshape(B * V, C, h, w).permute(0, 2, 3, 1).contiguous())
        got2 = _ops().costvol(nhwc.view(B, V, h, w, C), proj.to(dev), depth.to(dev), G, channels_last=True, impl="gather").cpu()
        assert torch.equal(got2, got)
        from casmvsnet_pl_amd import _lib
        if _lib.load().casmvs_costvol_lds_supported(C, w, D, V - 1, G):
            got3 = _ops().costvol(nhwc.view(B, V, h, w, C), proj.to(dev), depth.to(dev), G, channels_last=True, impl="lds").cpu()
            assert torch.equal(got3, got)


@pytest.mark.parametrize("C,G,h,w,D", [(8, 1, 64, 128, 8), (16, 1, 48, 64, 32), (32, 1, 32, 64, 16), (16, 8, 48, 64, 8)])
def test_costvol_lds_with_taps_outside_the_box(dev, report, C, G, h, w, D):
    """Per-pixel depth noise of hundreds of units scatters the taps of a tile over far more source pixels than a
    view's LDS box holds: the lanes whose taps fall outside gather from the global map.  Same bits either way."""
    V, B = 3, 1
    g = torch.Generator().manual_seed(C + D)
    feats = torch.randn(B, V, C, h, w, generator=g)
    proj, dmin, dint = _proj_like(B, V, h, w, seed=3, level=1)
    depth = 430.0 + 500.0 * torch.rand(B, 1, h, w, generator=g) + torch.arange(D).view(1, D, 1, 1) * dint * 2
    nhwc = _ops().nchw_to_nhwc(feats.reshape(B * V, C, h, w).to(dev)).view(B, V, h, w, C)
    a = _ops().costvol(nhwc, proj.to(dev), depth.to(dev), G, channels_last=True, impl="gather")
    b = _ops().costvol(nhwc, proj.to(dev), depth.to(dev), G, channels_last=True, impl="lds")
    want = R.cost_volume(feats, proj, depth, G)
    report("costvol_lds_noisy", shape=[C, G, h, w, D], max_abs=max_abs(b.cpu(), want))
    assert torch.equal(a, b)
    assert max_abs(b.cpu(), want) < 5e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("B,C,G,h,w,D,depth_kind", [(4, 16, 1, 256, 320, 32, "smooth"), (8, 32, 1, 128, 160, 48, "smooth"), (4, 16, 1, 256, 320, 32, "noisy"),
                                                       (4, 16, 8, 256, 320, 32, "smooth")])
def test_plane_sweep_16_planes_per_workgroup_equals_the_gather_kernels(dev, report, B, C, G, h, w, D, depth_kind):
    """The LDS plane sweep takes 16 planes per workgroup (half the prologues) only where the launch keeps >= 4 rounds of resident workgroups, i.e. at
    the cascade's level-1 / level-2 shapes from batch 4 on - sizes no other parity test reaches.  Fused variance / correlation volume and the un-fused
    homo_warp (box staged from the channel planes, and from the pixel-major copy) against the gather kernels, bit for bit, three launches each: the
    first 16-plane build passed every small-shape test and stored a depth hypothesis into channels 12-15 of four pixels per wave and plane at these
    sizes - a store-data hazard the compiler does not pad (csrc/costvol_lds.hip: store_plane_transposed; tools/store_hazard_lint.py) - differently
    in every run."""
    ops = _ops()
    V = 3
    g = torch.Generator().manual_seed(C + D + G)
    feats = torch.randn(B, V, C, h, w, generator=g)
    proj, dmin, dint = _proj_like(B, V, h, w, seed=5, level=1)
    k = torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)
    if depth_kind == "smooth":
        base = 680.0 - D / 2 * dint * 2 + 60.0 * torch.sin(torch.linspace(0, 6.0, w)).view(1, 1, 1, w)
        depth = (base + k * dint * 2).expand(B, D, h, w).contiguous()
    else:   # what the previous level's regression gives with random weights: taps of a tile spread beyond its box
        coarse = 680.0 + 40.0 * torch.randn(B, 1, h // 2, w // 2, generator=g)
        depth = (torch.nn.functional.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True) - D / 2 * dint * 2 + k * dint * 2).contiguous()
    fd, P, dv = feats.to(dev), proj.to(dev), depth.to(dev)
    nhwc = ops.nchw_to_nhwc(fd.reshape(B * V, C, h, w)).view(B, V, h, w, C)
    want = ops.costvol(nhwc, P, dv, G, channels_last=True, impl="gather")
    bad = {"fused": sum(0 if torch.equal(ops.costvol(nhwc, P, dv, G, channels_last=True, impl="lds"), want) else 1 for _ in range(3))}
    if G == 1:
        src, P1 = fd[:, 1].contiguous(), P[:, 0].contiguous()
        wwant = ops.homo_warp(src, P1, dv, impl="gather")
        for impl in ("lds", "lds_copy"):
            bad["warp_" + impl] = sum(0 if torch.equal(ops.homo_warp(src, P1, dv, impl=impl), wwant) else 1 for _ in range(3))
    report("plane_sweep_16_planes", shape=[B, C, G, h, w, D], depth=depth_kind, differing_launches=bad)
    assert not any(bad.values()), bad


@pytest.mark.parametrize("V,C,G,h,w,D", [(3, 16, 1, 32, 64, 16), (5, 8, 1, 24, 32, 8), (5, 32, 8, 16, 32, 8), (4, 16, 4, 24, 40, 8)])
def test_view_sharded_partial_sums(dev, report, V, C, G, h, w, D):
    """Sum x / Sum x^2 (and the correlation) are linear in the source views.  One rank (all views,
    include_ref) + finalise == the fused kernel bit for bit; two ranks' partial sums added == fused to rounding."""
    ops = _ops()
    B = 2
    g = torch.Generator().manual_seed(V * 7 + C)
    feats = torch.randn(B, V, C, h, w, generator=g)
    proj, dmin, dint = _proj_like(B, V, h, w, seed=V)
    depth = dmin + torch.rand(B, 1, h, w, generator=g) * 300.0 + torch.arange(D).view(1, D, 1, 1) * dint * 2
    nhwc = ops.nchw_to_nhwc(feats.reshape(B * V, C, h, w).to(dev)).view(B, V, h, w, C)
    P, dv = proj.to(dev), depth.to(dev)
    fused = ops.costvol(nhwc, P, dv, G, channels_last=True, impl="lds")
    one = ops.costvol_finalize(ops.costvol_partial(nhwc, P, dv, 1, V, G, include_ref=True), V, G)
    assert torch.equal(one, fused)
    mid = 1 + (V - 1) // 2
    p0 = ops.costvol_partial(nhwc, P, dv, 1, mid, G, include_ref=True)     # "rank 0": views [1, mid) + the reference terms
    p1 = ops.costvol_partial(nhwc, P, dv, mid, V, G, include_ref=False)    # "rank 1": views [mid, V)
    two = ops.costvol_finalize(p0 + p1, V, G)                               # the all-reduce(SUM), then every rank finalises
    err = scaled_err(two, fused)
    report("view_sharded_partial", shape=[V, C, G, h, w, D], scaled_err=err)
    assert err < 2e-6  # a different summation order of the same terms
    want = R.cost_volume(feats, proj, depth, G)
    assert max_abs(two.cpu(), want) < 5e-5 * max(1.0, float(want.abs().max()))


def test_costvol_identical_views_have_zero_variance(dev):
    feats = torch.randn(1, 1, 8, 16, 16).expand(1, 3, 8, 16, 16).contiguous()
    P = torch.eye(4)[:3].expand(1, 2, 3, 4).contiguous()
    depth = torch.full((1, 4, 16, 16), 500.0)
    got = _ops().costvol(feats.to(dev), P.to(dev), depth.to(dev), 1).cpu()
    assert float(got.abs().max()) < 1e-5


@pytest.mark.parametrize("B,D,hp,wp", [(1, 8, 16, 20), (2, 32, 9, 7), (1, 48, 32, 40)])
def test_depth_hypotheses_match_oracle(dev, report, B, D, hp, wp):
    g = torch.Generator().manual_seed(D)
    prev = 400.0 + torch.rand(B, hp, wp, generator=g) * 600.0
    prev[0, 0, 0] = 3.0  # forces the clamp_min(1e-7) branch
    interval = torch.tensor([2.65 * 2, 1.7][:B]).view(B, 1)
    up = F.interpolate(prev.unsqueeze(1), scale_factor=2, mode="bilinear", align_corners=True)
    want = R.get_depth_values(up, D, interval)
    half = (D / 2) * interval.view(B)
    got = _ops().depth_hypotheses(prev.to(dev), None, interval.view(B).to(dev), half.to(dev), D, 2 * hp, 2 * wp).cpu()
    err = rel_err(got, want)
    report("hypotheses", shape=[B, D, hp, wp], rel=err)
    assert err < 3e-6  # measured 2.7e-7: bilinear x2 weights, ATen's separable order vs ours, a few ulp
    # coarsest level
    want0 = R.initial_depth_values(425.0, 2.65 * 4.0, D, B, hp, wp)
    got0 = _ops().depth_hypotheses(None, torch.full((B,), 425.0, device=dev), torch.full((B,), 2.65 * 4.0, device=dev),
                                  None, D, hp, wp).cpu()
    assert max_abs(got0, want0) == 0.0


@pytest.mark.parametrize("B,D,h,w", [(1, 8, 32, 40), (2, 32, 16, 24), (1, 48, 32, 40), (1, 12, 9, 11), (1, 64, 8, 8)])
def test_softmax_regress_matches_oracle(dev, report, B, D, h, w):
    g = torch.Generator().manual_seed(D + h)
    cost = torch.randn(B, D, h, w, generator=g) * 3.0
    dv = 425.0 + torch.rand(B, 1, h, w, generator=g) * 100 + torch.arange(D).view(1, D, 1, 1) * 2.65
    cost[0, :, 0, 0] = -1e4
    cost[0, D - 1, 0, 0] = 50.0  # one-hot at the last plane: window clipped at the end
    cost[0, :, 0, 1] = -1e4
    cost[0, 0, 0, 1] = 50.0      # one-hot at the first plane: window clipped at the start
    d_w, c_w, i_w = R.softmax_regress(cost, dv)
    d_g, c_g, i_g = _ops().softmax_regress(cost.to(dev), dv.to(dev), return_index=True)
    d_g, c_g, i_g = d_g.cpu(), c_g.cpu(), i_g.cpu().long()
    prob = F.softmax(cost, 1)
    e = (prob * torch.arange(D).view(1, D, 1, 1)).sum(1)
    near = ((e - e.round()).abs() < 1e-3)
    mism = (i_g != i_w)
    report("softmax_regress", shape=[B, D, h, w], depth_rel=rel_err(d_g, d_w), conf_abs=max_abs(c_g, c_w),
           index_mismatch=int(mism.sum()), index_mismatch_off_boundary=int((mism & ~near).sum()))
    assert rel_err(d_g, d_w) < 5e-6  # measured 4.5e-7
    assert int((mism & ~near).sum()) == 0
    assert max_abs(c_g[~mism], c_w[~mism]) < 2e-6  # measured 1.2e-7
    assert float(d_g[0, 0, 0]) == pytest.approx(float(dv[0, D - 1, 0, 0]), rel=1e-6)


def _conv_ref(kind, x, w, scale, shift, skip, slope):
    ops = _ops()
    if kind == ops.CONV_T2:
        y = F.conv_transpose3d(x, w, None, stride=2, padding=1, output_padding=1)
    else:
        y = F.conv3d(x, w, None, stride=1 if kind == ops.CONV_S1 else 2, padding=1)
    y = y * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    y = torch.where(y > 0, y, y * slope)
    return y if skip is None else y + skip


CONV_CASES = [  # kind, cin, cout, B, D, H, W, with_skip
    (0, 32, 8, 1, 8, 16, 40, False), (0, 8, 8, 2, 4, 8, 32, True), (0, 5, 8, 1, 6, 10, 37, False),
    (0, 8, 1, 1, 8, 16, 40, False), (0, 16, 16, 1, 12, 24, 40, False), (0, 16, 16, 1, 4, 8, 16, True),
    (0, 32, 32, 1, 6, 16, 20, False), (0, 64, 64, 1, 2, 16, 20, False), (0, 64, 64, 1, 1, 8, 10, False),
    (1, 8, 16, 1, 8, 16, 40, False), (1, 16, 32, 2, 4, 16, 24, False), (1, 32, 64, 1, 2, 8, 12, False),
    (2, 64, 32, 1, 1, 8, 10, True), (2, 32, 16, 1, 2, 8, 12, True), (2, 16, 8, 2, 4, 8, 20, True),
    (2, 16, 8, 1, 3, 5, 7, False),
    # large enough for the "wide" workgroup variants (>= 512 wide blocks)
    (0, 16, 32, 1, 16, 64, 128, True), (1, 8, 16, 1, 16, 256, 256, False), (2, 32, 16, 1, 8, 64, 128, True),
    (0, 3, 16, 1, 5, 9, 21, False), (2, 6, 32, 1, 2, 5, 19, False),
    # stride 2 with W % 4 != 0 (dword staging path) and a wide one with W % 4 == 0 whose tiles are ragged
    (1, 8, 16, 1, 4, 8, 10, False), (1, 16, 16, 1, 6, 44, 72, False)]


@pytest.mark.parametrize("kind,cin,cout,B,D,H,W,with_skip", CONV_CASES)
def test_conv3d_layer_matches_torch_cpu(dev, report, kind, cin, cout, B, D, H, W, with_skip):
    ops = _ops()
    g = torch.Generator().manual_seed(kind * 1000 + cin * 10 + cout + D)
    x = torch.randn(B, cin, D, H, W, generator=g)
    wshape = (cin, cout, 3, 3, 3) if kind == ops.CONV_T2 else (cout, cin, 3, 3, 3)
    w = torch.randn(wshape, generator=g) * (2.0 / (27 * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    slope = 1.0 if cout == 1 else 0.01
    want = _conv_ref(kind, x, w, scale, shift, None, slope)
    skip = torch.randn(want.shape, generator=g) if with_skip else None
    if skip is not None:
        want = want + skip
    packed = ops.conv3d_pack(kind, w, scale, shift).to(dev)
    got = ops.conv3d_forward(kind, packed, x.to(dev), cout, None if skip is None else skip.to(dev), slope).cpu()
    err = scaled_err(got, want)
    report("conv3d", kind=kind, cin=cin, cout=cout, shape=[B, D, H, W], scaled_err=err, max_abs=max_abs(got, want))
    assert got.shape == want.shape
    assert err < 1.2e-5  # measured <= 1.1e-6


def test_mfma_bf16_lane_semantics_selftest(report):
    rc, dump, msg = _ops().selftest_mfma_bf16()
    report("mfma_bf16_selftest", rc=rc, msg=msg, reg0=dump[0, :8].tolist())
    assert rc == 0, msg


SB_CASES = [(8, 1, 8, 16, 32), (16, 2, 4, 8, 64), (32, 1, 12, 24, 40), (8, 1, 5, 9, 36), (16, 1, 6, 20, 100), (32, 1, 3, 5, 8)]


@pytest.mark.parametrize("cin,B,D,H,W", SB_CASES)
def test_conv0_splitbf16_matches_torch_cpu(dev, report, cin, B, D, H, W):
    """csrc/conv0_splitbf16.hip: conv0 (Conv3d cin -> 8 + folded ABN + leaky-relu, mvsnet.py:63) with every float32 operand
    as three exact bf16 slices on the bf16 matrix cores: vs torch CPU float64 at the SAME bound as the float32-MFMA layer
    kernels (1.2e-5 of the range; measured ~3e-7: at or below the float32 kernel's own error), six and nine partial
    products, ragged tiles in x / y / z; and against the float32-MFMA kernel of the same layer."""
    ops = _ops()
    g = torch.Generator().manual_seed(cin * 100 + D + W)
    x = torch.randn(B, cin, D, H, W, generator=g)
    w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    want = _conv_ref(ops.CONV_S1, x.double(), w.double(), scale.double(), shift.double(), None, 0.01)
    packed = ops.conv0_splitbf16_pack(w, scale, shift).to(dev)
    xd = x.to(dev)
    got6 = ops.conv0_splitbf16_forward(packed, xd, slope=0.01, terms=6).cpu()
    got9 = ops.conv0_splitbf16_forward(packed, xd, slope=0.01, terms=9).cpu()
    f32 = ops.conv3d_forward(ops.CONV_S1, ops.conv3d_pack(ops.CONV_S1, w, scale, shift).to(dev), xd, 8, slope=0.01).cpu()
    e6, e9, ef = scaled_err(got6, want), scaled_err(got9, want), scaled_err(f32, want)
    report("conv0_splitbf16", shape=[cin, B, D, H, W], err_6_terms=e6, err_9_terms=e9, err_f32_mfma=ef, vs_f32_kernel=scaled_err(got6, f32))
    assert e6 < 1.2e-5 and e9 < 1.2e-5
    assert e6 < 4 * max(ef, 2e-7)     # float32-grade: no worse than a few times the float32 kernel's own distance to float64


def test_mfma_f16_lane_semantics_selftest(report):
    rc, dump, msg = _ops().selftest_mfma_f16()
    report("mfma_f16_selftest", rc=rc, msg=msg, reg0=dump[0, :8].tolist())
    assert rc == 0, msg


SF_CASES = [c + (1.0,) for c in SB_CASES] + [(8, 1, 5, 9, 36, 3e4), (16, 1, 6, 10, 68, 1e-30), (32, 2, 9, 7, 32, 1e-3)]


@pytest.mark.parametrize("cin,B,D,H,W,amp", SF_CASES)
def test_conv0_splitf16_matches_torch_cpu(dev, report, cin, B, D, H, W, amp):
    """csrc/conv0_splitf16.hip: conv0 with every float32 operand as two float16 slices behind exact power-of-two scalings (one
    per weight tensor, one per staged tile and chunk) on the f16 matrix cores: vs torch CPU float64 at the SAME bound as the
    float32-MFMA layer kernels and no worse than a few times that kernel's own error; three and four partial products;
    ragged tiles; inputs far outside float16's range (3e4, 1e-30) and a corner 10^6 times smaller than the rest."""
    ops = _ops()
    g = torch.Generator().manual_seed(cin * 100 + D + W)
    x = torch.randn(B, cin, D, H, W, generator=g) * amp
    x[..., -2:, -3:] *= 1e-6
    w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1 * amp
    want = _conv_ref(ops.CONV_S1, x.double(), w.double(), scale.double(), shift.double(), None, 0.01)
    packed = ops.conv0_splitf16_pack(w, scale, shift).to(dev)
    xd = x.to(dev)
    got3 = ops.conv0_splitf16_forward(packed, xd, slope=0.01, terms=3).cpu()
    got4 = ops.conv0_splitf16_forward(packed, xd, slope=0.01, terms=4).cpu()
    f32 = ops.conv3d_forward(ops.CONV_S1, ops.conv3d_pack(ops.CONV_S1, w, scale, shift).to(dev), xd, 8, slope=0.01).cpu()
    e3, e4, ef = scaled_err(got3, want), scaled_err(got4, want), scaled_err(f32, want)
    report("conv0_splitf16", shape=[cin, B, D, H, W], amp=amp, err_3_terms=e3, err_4_terms=e4, err_f32_mfma=ef, vs_f32_kernel=scaled_err(got3, f32))
    assert torch.isfinite(got3).all()
    assert e3 < 1.2e-5 and e4 < 1.2e-5
    assert e3 < 4 * max(ef, 2e-7)     # float32-grade: no worse than a few times the float32 kernel's own distance to float64


CI_CASES = [(16, 1, 4, 8, 16, 1.0), (16, 2, 5, 9, 36, 1.0), (32, 1, 6, 10, 20, 1.0), (32, 2, 3, 5, 50, 1e-3), (16, 1, 9, 6, 34, 3e4), (32, 1, 4, 4, 16, 1e-30),
            (16, 1, 2, 3, 2, 1.0), (32, 2, 2, 20, 36, 1.0), (16, 1, 1, 9, 16, 1.0), (32, 8, 2, 128, 160, 1.0), (64, 1, 4, 8, 16, 1.0), (64, 2, 6, 10, 36, 1e-3),
            (64, 8, 4, 32, 40, 1.0)]


@pytest.mark.parametrize("c,B,D,H,W,amp", CI_CASES)
def test_conv_ci_splitf16_matches_torch_cpu(dev, report, c, B, D, H, W, amp):
    """csrc/conv_ci_splitf16.hip: conv2 (16 -> 16) / conv4 (32 -> 32) of CostRegNet (mvsnet.py:66,69) on the f16 matrix cores, every
    float32 operand as two scaled float16 slices: vs torch CPU float64 at the SAME bound as the float32-MFMA layer kernels and no
    worse than a few times that kernel's own error; ragged tiles in all three directions, inputs far outside float16's range."""
    ops = _ops()
    g = torch.Generator().manual_seed(c * 100 + D + W)
    x = torch.randn(B, c, D, H, W, generator=g) * amp
    x[..., -1:, -1:] *= 1e-6
    w = torch.randn(c, c, 3, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1 * amp
    want = _conv_ref(ops.CONV_S1, x.double(), w.double(), scale.double(), shift.double(), None, 0.01)
    packed = ops.conv_ci_splitf16_pack(w, scale, shift).to(dev)
    xd = x.to(dev)
    got = ops.conv_ci_splitf16_forward(packed, xd, c, slope=0.01).cpu()
    f32 = ops.conv3d_forward(ops.CONV_S1, ops.conv3d_pack(ops.CONV_S1, w, scale, shift).to(dev), xd, c, slope=0.01).cpu()
    e3, ef = scaled_err(got, want), scaled_err(f32, want)
    report("conv_ci_splitf16", shape=[c, B, D, H, W], amp=amp, err_splitf16=e3, err_f32_mfma=ef, vs_f32_kernel=scaled_err(got, f32))
    assert torch.isfinite(got).all()
    assert e3 < 1.2e-5
    assert e3 < 4 * max(ef, 2e-7)


S2_CASES = [(8, 1, 4, 8, 16, 1.0), (8, 2, 6, 10, 72, 1.0), (16, 1, 6, 14, 40, 1.0), (16, 2, 4, 26, 136, 1e-3), (8, 1, 10, 12, 36, 3e4), (16, 1, 4, 4, 16, 1e-30),
            (8, 1, 5, 27, 132, 1.0), (16, 1, 9, 13, 8, 1.0), (8, 1, 1, 3, 4, 1.0), (8, 8, 32, 256, 320, 1.0), (16, 8, 16, 128, 160, 1.0), (8, 1, 8, 512, 640, 1.0)]


@pytest.mark.parametrize("cin,B,D,H,W,amp", S2_CASES)
def test_conv_s2_splitf16_matches_torch_cpu(dev, report, cin, B, D, H, W, amp):
    """csrc/conv_s2_splitf16.hip: conv1 (8 -> 16) / conv3 (16 -> 32) of CostRegNet (mvsnet.py:64-65,67-68: Conv3d k3 s2 p1 + ABN) on the f16 matrix cores,
    input-stationary along z: vs torch CPU float64 at the bound of the float32-MFMA layer kernels and no worse than a few times that kernel's own error;
    odd sizes along every axis (the float32 kernel needs even ones: compared where it runs), several z segments, the cascade's shapes at batch 8, inputs
    far outside float16's range; twice for the bits."""
    ops = _ops()
    cout = 2 * cin
    g = torch.Generator().manual_seed(cin * 100 + D + W)
    x = torch.randn(B, cin, D, H, W, generator=g) * amp
    x[..., -1:, -1:] *= 1e-6
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1 * amp
    if x.numel() > 5e7:   # the cascade's shapes: float64 on the CPU takes minutes - the float32 kernel is the reference there
        want = None
    else:
        want = F.conv3d(x.double(), w.double(), stride=2, padding=1) * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
        want = torch.where(want > 0, want, want * 0.01)
    packed = ops.conv_s2_splitf16_pack(w, scale, shift).to(dev)
    xd = x.to(dev)
    got_d = ops.conv_s2_splitf16_forward(packed, xd, cout, slope=0.01)
    assert torch.equal(got_d, ops.conv_s2_splitf16_forward(packed, xd, cout, slope=0.01))
    got = got_d.cpu()
    assert torch.isfinite(got).all()
    even = D % 2 == 0 and H % 2 == 0 and W % 2 == 0
    f32 = ops.conv3d_forward(ops.CONV_S2, ops.conv3d_pack(ops.CONV_S2, w, scale, shift).to(dev), xd, cout, slope=0.01).cpu() if even else None
    if want is None:
        e = scaled_err(got, f32.double())
        report("conv_s2_splitf16", shape=[cin, B, D, H, W], amp=amp, vs_f32_kernel=e)
        assert e < 3e-6
        return
    e3 = scaled_err(got, want)
    ef = scaled_err(f32, want) if even else None
    report("conv_s2_splitf16", shape=[cin, B, D, H, W], amp=amp, err_splitf16=e3, err_f32_mfma=ef)
    assert e3 < 1.2e-5
    assert ef is None or e3 < 4 * max(ef, 2e-7)


K5S2_CASES = [(8, 1, 20, 72, 1.0), (16, 1, 18, 40, 1.0), (8, 3, 34, 136, 1e-3), (16, 2, 6, 8, 3e4), (8, 1, 2, 4, 1.0), (16, 1, 38, 132, 1e-30),
              (8, 24, 512, 640, 1.0), (16, 24, 256, 320, 1.0), (8, 3, 1184, 1600, 1.0)]


@pytest.mark.parametrize("cin,N,H,W,amp", K5S2_CASES)
def test_conv2d_k5s2_splitf16_matches_torch_cpu(dev, report, cin, N, H, W, amp):
    """csrc/conv2d_k5s2_splitf16.hip: conv1.0 (8 -> 16) / conv2.0 (16 -> 32) of FeatureNet (mvsnet.py:19,24: Conv2d k5 s2 p2 + ABN) on the f16 matrix cores:
    vs torch CPU float64 at the bound of the float32-MFMA layer kernel and no worse than a few times that kernel's own error; borders inside a tile, several
    tiles, the smallest image, the cascade's shapes at batch 8 x 3 views, a full-resolution DTU image, inputs far outside float16's range; twice for the
    bits."""
    ops = _ops()
    cout = 2 * cin
    g = torch.Generator().manual_seed(cin * 100 + H + W)
    x = torch.randn(N, cin, H, W, generator=g) * amp
    x[..., -1:, -1:] *= 1e-6
    w = torch.randn(cout, cin, 5, 5, generator=g) * 0.1
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1 * amp
    want = F.conv2d(x.double(), w.double(), stride=2, padding=2) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.01)
    packed = ops.conv2d_k5s2_splitf16_pack(w, scale, shift).to(dev)
    xd = x.to(dev)
    got_d = ops.conv2d_k5s2_splitf16_forward(packed, xd, cout, slope=0.01)
    assert torch.equal(got_d, ops.conv2d_k5s2_splitf16_forward(packed, xd, cout, slope=0.01))
    got = got_d.cpu()
    assert torch.isfinite(got).all()
    f32 = ops.conv2d_forward(ops.CONV2D_K5S2, ops.conv2d_pack(ops.CONV2D_K5S2, w, scale, shift).to(dev), xd, cout, slope=0.01).cpu()
    e3, ef = scaled_err(got, want), scaled_err(f32, want)
    report("conv2d_k5s2_splitf16", shape=[cin, N, H, W], amp=amp, err_splitf16=e3, err_f32_mfma=ef)
    assert e3 < 1.2e-5
    assert e3 < 4 * max(ef, 2e-7)


FNET0_CASES = [(1, 22, 64, 1.0), (2, 6, 8, 1.0), (3, 44, 92, 1e-3), (1, 2, 2, 1.0), (1, 20, 30, 1e-30), (1, 24, 36, 3e4), (24, 512, 640, 1.0), (3, 1184, 1600, 1.0), (5, 864, 1152, 1.0)]


@pytest.mark.parametrize("N,H,W,amp", FNET0_CASES)
def test_fnet_conv0_mm_matches_torch_cpu(dev, report, N, H, W, amp):
    """csrc/fnet_conv0_mm.hip: FeatureNet.conv0 = ConvBnReLU(3, 8, 3) -> ConvBnReLU(8, 8, 3) (mvsnet.py:14-16) as ONE kernel with both layers on the f16
    matrix cores: vs the two layers in torch CPU float64 at the bound of the float32-MFMA layer kernels and no worse than a few times the error of the two
    float32-MFMA launches it replaces; borders inside a tile, tiles 4 pixels wide, images below one tile, the benched shape at batch 8 x 3 views, full-resolution
    DTU images, inputs far outside float16's range; twice for the bits."""
    ops = _ops()
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(N, 3, H, W, generator=g) * amp
    x[..., -1:, -1:] *= 1e-6
    w0, w1 = torch.randn(8, 3, 3, 3, generator=g) * 0.3, torch.randn(8, 8, 3, 3, generator=g) * 0.2
    sc0, sh0 = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1 * amp
    sc1, sh1 = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1 * amp
    big = N * H * W > 4_000_000   # float64 on the CPU for 24 images takes a while: the first three images
    xs = x[:3] if big else x
    mid = F.conv2d(xs.double(), w0.double(), padding=1) * sc0.double().view(1, -1, 1, 1) + sh0.double().view(1, -1, 1, 1)
    mid = torch.where(mid > 0, mid, mid * 0.01)
    want = F.conv2d(mid, w1.double(), padding=1) * sc1.double().view(1, -1, 1, 1) + sh1.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.01)
    packed = ops.fnet_conv0_mm_pack(w0, sc0, sh0, w1, sc1, sh1).to(dev)
    xd = x.to(dev)
    got_d = ops.fnet_conv0_mm(packed, xd, slope=0.01)
    assert torch.equal(got_d, ops.fnet_conv0_mm(packed, xd, slope=0.01))
    got = got_d[:xs.shape[0]].cpu()
    assert torch.isfinite(got_d).all()
    p0, p1 = ops.conv2d_pack(ops.CONV2D_K3, w0, sc0, sh0).to(dev), ops.conv2d_pack(ops.CONV2D_K3, w1, sc1, sh1).to(dev)
    f32 = ops.conv2d_forward(ops.CONV2D_K3, p1, ops.conv2d_forward(ops.CONV2D_K3, p0, xd[:xs.shape[0]].contiguous(), 8, slope=0.01), 8, slope=0.01).cpu()
    e3, ef = scaled_err(got, want), scaled_err(f32, want)
    report("fnet_conv0_mm", shape=[N, H, W], amp=amp, err_fused_f16=e3, err_two_f32_mfma_layers=ef)
    assert e3 < 1.2e-5
    assert e3 < 4 * max(ef, 2e-7)


def test_featurenet_with_the_fused_conv0_equals_the_two_layer_form(dev, report):
    """FeatureNet.fuse_conv0 (the engine's default): the three feature maps with conv0.0 + conv0.1 as one f16 kernel against the same module running the two
    float32-MFMA layers - both inside the float32-MFMA kernels' own distance from a float64 FeatureNet (asserted against the oracle elsewhere)."""
    from casmvsnet_pl_amd import ABN
    from casmvsnet_pl_amd.mvsnet import FeatureNet
    from casmvsnet_pl_amd.synthetic import randomize_state_dict
    net = FeatureNet(ABN)
    randomize_state_dict(net.state_dict(), seed=3)
    net = net.to(dev).eval()
    x = torch.randn(3, 3, 128, 160, generator=torch.Generator().manual_seed(1)).to(dev)
    a = {k: v.clone() for k, v in net(x).items()}
    assert net._ci2d is not None and net._ci2d[7] is not None
    net.fuse_conv0 = False
    b = net(x)
    assert net._ci2d[7] is None
    errs = {k: scaled_err(a[k].cpu(), b[k].cpu().double()) for k in a}
    report("featurenet_fused_conv0_vs_two_layers", errs=errs)
    assert all(e < 5e-6 for e in errs.values()), errs


PROB_CASES = [(1, 8, 8, 64), (2, 8, 32, 40), (1, 32, 16, 72), (2, 48, 24, 132), (1, 12, 9, 36), (1, 4, 5, 8), (1, 16, 70, 196)]


@pytest.mark.parametrize("B,D,h,w", PROB_CASES)
def test_prob_head_depth_walk_matches_conv3d(dev, report, B, D, h, w):
    """csrc/prob_regress.hip (mvsnet.py:89,104): Conv3d(8 -> 1, k3 p1) + bias with the depth axis walked by one workgroup
    per pixel tile - whole range, library-chosen chunks and explicit chunk sizes (ragged last chunk, one-plane halos,
    ragged tiles in x / y) - vs torch CPU float64; the chunked forms are BIT-equal to the unchunked one (same FMA order)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + D * 10 + w)
    x = torch.randn(B, 8, D, h, w, generator=g)
    wt = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.2
    bias = torch.randn(1, generator=g)
    want = F.conv3d(x.double(), wt.double(), bias.double(), padding=1)[:, 0]
    packed = ops.conv3d_pack(ops.CONV_S1, wt, None, bias).to(dev)
    xd = x.to(dev)
    whole = ops.prob_regress(packed, xd, zchunk=D).cpu()
    err = scaled_err(whole, want)
    report("prob_zwalk", shape=[B, D, h, w], scaled_err=err)
    assert err < 1.2e-5
    for zc in (0, 4, 8, 5, 1):
        assert torch.equal(ops.prob_regress(packed, xd, zchunk=zc).cpu(), whole), f"zchunk={zc}"
    # the layer entry point of the C ABI (casmvs_conv3d_forward_f32, cout == 1) runs the same kernel
    assert torch.equal(ops.conv3d_forward(ops.CONV_S1, packed, xd, 1, slope=1.0).cpu()[:, 0], whole)


@pytest.mark.parametrize("B,D,h,w", [(1, 8, 32, 64), (2, 32, 16, 40), (1, 48, 16, 72), (1, 12, 9, 36), (1, 16, 12, 20)])
def test_prob_head_fused_regression_equals_the_two_kernel_form(dev, report, B, D, h, w):
    """mvsnet.py:174-193 inside the head's kernel (one chunk: every thread regresses the cost values it produced) ==
    casmvs_softmax_regress_f32 on the same cost, bit for bit - depth, confidence, index - and the chunked call (which
    launches the regression separately) as well; the regression itself vs the oracle."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 100 + D + w)
    x = torch.randn(B, 8, D, h, w, generator=g)
    wt = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.6   # a peaked softmax
    bias = torch.randn(1, generator=g)
    dv = 425.0 + torch.rand(B, 1, h, w, generator=g) * 100.0 + 2.5 * torch.arange(D).view(1, D, 1, 1)
    packed = ops.conv3d_pack(ops.CONV_S1, wt, None, bias).to(dev)
    xd, dvd = x.to(dev), dv.contiguous().to(dev)
    cost, depth, conf, idx = ops.prob_regress(packed, xd, dvd, zchunk=D, return_index=True)       # fused
    d2, c2, i2 = ops.softmax_regress(cost, dvd, return_index=True)
    assert torch.equal(depth, d2) and torch.equal(conf, c2) and torch.equal(idx, i2)
    cost3, d3, c3, i3 = ops.prob_regress(packed, xd, dvd, zchunk=4, return_index=True)            # chunked + regression launch
    assert torch.equal(cost3, cost) and torch.equal(d3, depth) and torch.equal(c3, conf) and torch.equal(i3, idx)
    want_d, _, _ = R.softmax_regress(cost.cpu(), dv.contiguous())
    report("prob_regress_fused", shape=[B, D, h, w], depth_rel=rel_err(depth, want_d))
    assert rel_err(depth, want_d) < 1e-5


@pytest.mark.parametrize("cin,B,D,h,w", [(8, 1, 8, 32, 40), (16, 1, 32, 16, 24), (32, 2, 16, 16, 16)])
def test_costreg_matches_oracle(dev, report, cin, B, D, h, w):
    from casmvsnet_pl_amd import ABN, CostRegNet
    from casmvsnet_pl_amd.synthetic import randomize_state_dict
    net = CostRegNet(cin, ABN)
    sd = net.state_dict()
    randomize_state_dict({("cost_reg_2." + k): v for k, v in sd.items()}, seed=cin)
    net.eval()
    x = torch.randn(B, cin, D, h, w, generator=torch.Generator().manual_seed(cin)).abs()
    want, inter = R.cost_reg_net(x, {"n." + k: v for k, v in net.state_dict().items()}, "n", return_intermediates=True)
    with torch.no_grad():
        got = net.to(dev)(x.to(dev)).cpu()
    err = scaled_err(got, want)
    report("costreg", cin=cin, shape=[B, D, h, w], scaled_err=err)
    assert got.shape == want.shape
    assert err < 1.2e-5  # measured 1.1e-6


def test_one_plane_conv6_as_a_2d_layer_equals_the_3d_kernel(dev):
    """CostRegNet.conv6 at cascade level 0 (D / 8 = 1: mvsnet.py:72 on a one-plane volume) runs as the 2D convolution of the weight's middle z slice (the taps
    kz = 0 / 2 multiply zero padding only; csrc/conv3d_mfma.hip costreg_run).  Same taps, same order, same float32 MFMA: the 3D kernel's result bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(66)
    B, h, w = 2, 16, 24
    x = torch.randn(B, 64, 1, h, w, generator=g)
    wt = torch.randn(64, 64, 3, 3, 3, generator=g) * 0.05
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1
    y3 = ops.conv3d_forward(ops.CONV_S1, ops.conv3d_pack(ops.CONV_S1, wt, sc, sh).to(dev), x.to(dev), 64).cpu()
    y2 = ops.conv2d_forward(ops.CONV2D_K3, ops.conv2d_pack(ops.CONV2D_K3, wt[:, :, 1].contiguous(), sc, sh).to(dev), x[:, :, 0].contiguous().to(dev), 64, None, 0.01).cpu()
    assert torch.equal(y3[:, :, 0], y2)
    want = F.leaky_relu(F.conv3d(x.double(), wt.double(), padding=1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1), 0.01)
    assert scaled_err(y2.unsqueeze(2), want.float()) < 3e-6


CONV2D_CASES = [  # kind, cin, cout, N, H, W
    (3, 3, 8, 3, 64, 96), (3, 8, 8, 2, 40, 72), (3, 32, 8, 1, 37, 50), (3, 16, 16, 2, 32, 48), (3, 32, 16, 1, 24, 70),
    (3, 32, 32, 3, 16, 20), (3, 5, 16, 1, 9, 21), (3, 8, 8, 1, 10, 23), (4, 8, 16, 2, 64, 96), (4, 16, 32, 1, 36, 52), (5, 32, 32, 2, 16, 24),
    (5, 16, 32, 1, 11, 13), (6, 16, 32, 2, 32, 48), (6, 8, 32, 1, 64, 96), (6, 8, 32, 1, 2, 2),
    # big enough to leave more work items than resident workgroups (persistent loop, cross-tile prefetch)
    (3, 8, 8, 3, 512, 640), (4, 8, 16, 3, 512, 640), (6, 8, 32, 1, 512, 640)]


@pytest.mark.parametrize("kind,cin,cout,N,H,W", CONV2D_CASES)
def test_conv2d_layer_matches_torch_cpu(dev, report, kind, cin, cout, N, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(kind * 1000 + cin * 10 + cout + H)
    k = {ops.CONV2D_K3: 3, ops.CONV2D_K5S2: 5}.get(kind, 1)
    x = torch.randn(N, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    slope = 0.01 if kind in (ops.CONV2D_K3, ops.CONV2D_K5S2) else 1.0
    want = F.conv2d(x, w, None, stride=2 if kind == ops.CONV2D_K5S2 else 1, padding=k // 2)
    want = want * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * slope)
    up = None
    if kind == ops.CONV2D_K1_UP:
        up = torch.randn(N, cout, H // 2, W // 2, generator=g)
        want = want + F.interpolate(up, scale_factor=2, mode="bilinear", align_corners=True)
    packed = ops.conv2d_pack(kind, w, scale, shift).to(dev)
    got = ops.conv2d_forward(kind, packed, x.to(dev), cout, None if up is None else up.to(dev), slope).cpu()
    err = scaled_err(got, want)
    report("conv2d", kind=kind, cin=cin, cout=cout, shape=[N, H, W], scaled_err=err, max_abs=max_abs(got, want))
    assert got.shape == want.shape
    assert err < 1e-5  # measured <= 7e-7


def test_convbnrelu_module_runs_one_hip_layer(dev):
    from casmvsnet_pl_amd import ABN, ConvBnReLU
    g = torch.Generator().manual_seed(5)
    m = ConvBnReLU(8, 16, 5, 2, 2, norm_act=ABN).eval()
    with torch.no_grad():
        m.bn.running_var.uniform_(0.5, 1.5, generator=g)
        m.bn.running_mean.normal_(0, 0.1, generator=g)
        m.bn.weight.uniform_(0.6, 1.4, generator=g)
        m.bn.bias.normal_(0, 0.1, generator=g)
    x = torch.randn(2, 8, 24, 40, generator=g)
    want = F.leaky_relu(F.batch_norm(F.conv2d(x, m.conv.weight, None, stride=2, padding=2), m.bn.running_mean,
                                     m.bn.running_var, m.bn.weight, m.bn.bias, False, 0.0, m.bn.eps), 0.01)
    with torch.no_grad():
        got = m.to(dev)(x.to(dev)).cpu()
    assert got.shape == want.shape and scaled_err(got, want) < 2e-5


@pytest.mark.parametrize("N,H,W", [(3, 64, 96), (2, 32, 64), (5, 160, 128), (1, 36, 44)])
def test_featurenet_matches_oracle(dev, report, N, H, W):
    from casmvsnet_pl_amd import ABN, FeatureNet
    from casmvsnet_pl_amd.synthetic import randomize_state_dict
    net = FeatureNet(ABN)
    randomize_state_dict({("feature." + k): v for k, v in net.state_dict().items()}, seed=H)
    net.eval()
    x = torch.rand(N, 3, H, W, generator=torch.Generator().manual_seed(W)) * 2 - 1
    want = R.feature_net(x, {"feature." + k: v for k, v in net.state_dict().items()})
    with torch.no_grad():
        got = net.to(dev)(x.to(dev))
    errs = {}
    for l in range(3):
        g_l, w_l = got[f"level_{l}"].cpu(), want[f"level_{l}"]
        assert g_l.shape == w_l.shape
        errs[l] = scaled_err(g_l, w_l)
        assert torch.equal(net.last_channels_last[f"level_{l}"].cpu(), g_l.permute(0, 2, 3, 1).contiguous())
    report("featurenet", shape=[N, H, W], scaled_err=errs)
    assert max(errs.values()) < 1.4e-5  # measured 1.4e-6


@pytest.mark.parametrize("tail_mode", ["splitf16", "f32"])
def test_featurenet_pixel_major_only_call_equals_the_full_call(dev, tail_mode):
    """CascadeMVSNet.forward's own FeatureNet call drops the (N, C, h, w) stores of levels 0 / 1 (feat0 / feat1 = NULL at the C ABI): the three pixel-major
    maps are bit-equal to those of the full call, on the f16 output kernels (the stores fall into an empty buffer range) and on the float32 ones (the
    layout nobody asked for goes to the workspace)."""
    from casmvsnet_pl_amd import ABN, FeatureNet
    from casmvsnet_pl_amd.synthetic import randomize_state_dict
    net = FeatureNet(ABN)
    randomize_state_dict({("feature." + k): v for k, v in net.state_dict().items()}, seed=11)
    net = net.to(dev).eval()
    net.tail_mode = tail_mode
    x = (torch.rand(3, 3, 64, 96, generator=torch.Generator().manual_seed(2)) * 2 - 1).to(dev)
    with torch.no_grad():
        full = net(x)
        want = {k: v.clone() for k, v in net.last_channels_last.items()}
        got = net(x, pixel_major_only=True)
    assert net._split_active == (tail_mode == "splitf16")
    for l in range(3):
        assert torch.equal(got[f"level_{l}"], want[f"level_{l}"])
        assert torch.equal(got[f"level_{l}"], full[f"level_{l}"].permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize("N,H,W", [(1, 8, 64), (2, 36, 72), (1, 64, 196), (3, 128, 160)])
def test_fpn_fused_tail_matches_lat_upsample_smooth(dev, report, N, H, W):
    """csrc/fpn_fused.hip (mvsnet.py:36-38,50-51,54): feat0 = smooth0(lat0(conv0) + interpolate(feat1')) as one kernel over
    the composed 40-channel layer, vs torch CPU float64 - ragged tiles, image borders (the nine bias classes), both outputs."""
    from casmvsnet_pl_amd.mvsnet import compose_fpn_tail
    ops = _ops()
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    lw, lb = torch.randn(32, 8, 1, 1, generator=g) * 0.3, torch.randn(32, generator=g)
    sw, sb = torch.randn(8, 32, 3, 3, generator=g) * 0.2, torch.randn(8, generator=g)
    x, y = torch.randn(N, 8, H, W, generator=g), torch.randn(N, 32, H // 2, W // 2, generator=g)
    want = F.conv2d(F.conv2d(x.double(), lw.double(), lb.double()) + F.interpolate(y.double(), scale_factor=2, mode="bilinear", align_corners=True),
                    sw.double(), sb.double(), padding=1)
    w40, bias9 = compose_fpn_tail(lw, lb, sw, sb)
    packed = ops.conv2d_pack(ops.CONV2D_K3, w40, None, None).to(dev)
    got, got_cl = ops.fpn_tail0(packed, bias9.to(dev), x.to(dev), y.to(dev), channels_last_copy=True)
    err = scaled_err(got, want)
    report("fpn_tail0", shape=[N, H, W], scaled_err=err)
    assert err < 1.2e-5
    assert torch.equal(got_cl, got.permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize("N,H,W,amp", [(1, 8, 64, 1.0), (2, 36, 72, 1.0), (1, 64, 196, 1e-4), (3, 128, 160, 1.0), (1, 18, 32, 3e4)])
def test_fpn_fused_tail_splitf16_matches_lat_upsample_smooth(dev, report, N, H, W, amp):
    """csrc/fpn_fused_sf.hip: the fused tail on the f16 matrix cores (two scaled float16 slices per operand, float32 accumulation) vs
    torch CPU float64 at the float32 kernel's bound and no worse than a few times that kernel's own error; both outputs."""
    from casmvsnet_pl_amd.mvsnet import compose_fpn_tail
    ops = _ops()
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    lw, lb = torch.randn(32, 8, 1, 1, generator=g) * 0.3, torch.randn(32, generator=g) * amp
    sw, sb = torch.randn(8, 32, 3, 3, generator=g) * 0.2, torch.randn(8, generator=g) * amp
    x, y = torch.randn(N, 8, H, W, generator=g) * amp, torch.randn(N, 32, H // 2, W // 2, generator=g) * amp
    want = F.conv2d(F.conv2d(x.double(), lw.double(), lb.double()) + F.interpolate(y.double(), scale_factor=2, mode="bilinear", align_corners=True),
                    sw.double(), sb.double(), padding=1)
    w40, bias9 = compose_fpn_tail(lw, lb, sw, sb)
    psf = ops.fpn_tail0_splitf16_pack(w40).to(dev)
    got, got_cl = ops.fpn_tail0_splitf16(psf, bias9.to(dev), x.to(dev), y.to(dev), channels_last_copy=True)
    f32 = ops.fpn_tail0(ops.conv2d_pack(ops.CONV2D_K3, w40, None, None).to(dev), bias9.to(dev), x.to(dev), y.to(dev))
    err, ef = scaled_err(got, want), scaled_err(f32, want)
    report("fpn_tail0_splitf16", shape=[N, H, W], amp=amp, scaled_err=err, err_f32_kernel=ef)
    assert torch.isfinite(got).all()
    assert err < 1.2e-5 and err < 4 * max(ef, 2e-7)
    assert torch.equal(got_cl, got.permute(0, 2, 3, 1).contiguous())


@pytest.mark.parametrize("c,N,H,W,amp,cout", [(16, 1, 16, 16, 1.0, 16), (16, 2, 36, 72, 1.0, 16), (32, 1, 40, 50, 1.0, 32), (32, 3, 128, 160, 1e-3, 32), (16, 1, 9, 34, 3e4, 16),
                                              (32, 1, 4, 2, 1e-30, 32), (16, 6, 256, 320, 1.0, 16), (32, 2, 36, 72, 1.0, 16), (32, 6, 256, 320, 1.0, 16)])
def test_conv2d_ci_splitf16_matches_torch_cpu(dev, report, c, N, H, W, amp, cout):
    """csrc/conv2d_ci_splitf16.hip: FeatureNet's conv1.1 / conv1.2 (16 -> 16) and conv2.1 / conv2.2 (32 -> 32) on the f16 matrix cores vs
    torch CPU float64 at the float32 kernel's bound and no worse than a few times that kernel's own error; ragged tiles, persistent loop
    (the last case: 1920 tiles), inputs far outside float16's range."""
    ops = _ops()
    g = torch.Generator().manual_seed(c * 100 + H + W)
    x = torch.randn(N, c, H, W, generator=g) * amp
    w = torch.randn(cout, c, 3, 3, generator=g) * 0.1
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1 * amp
    want = F.conv2d(x.double(), w.double(), None, padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    want = torch.where(want > 0, want, want * 0.01)
    xd = x.to(dev)
    got, got_cl = ops.conv2d_ci_splitf16_forward(ops.conv2d_ci_splitf16_pack(w, scale, shift).to(dev), xd, cout=cout, slope=0.01, channels_last_copy=True)
    assert torch.equal(got_cl, got.permute(0, 2, 3, 1).contiguous())
    got = got.cpu()
    f32 = ops.conv2d_forward(ops.CONV2D_K3, ops.conv2d_pack(ops.CONV2D_K3, w, scale, shift).to(dev), xd, cout, slope=0.01).cpu()
    e3, ef = scaled_err(got, want), scaled_err(f32, want)
    report("conv2d_ci_splitf16", shape=[c, cout, N, H, W], amp=amp, err_splitf16=e3, err_f32_mfma=ef)
    assert torch.isfinite(got).all()
    assert e3 < 1.2e-5 and e3 < 4 * max(ef, 2e-7)


def test_split_f16_kernels_are_bit_stable_at_full_occupancy(dev, report):
    """Every launch of EVERY f16-matrix-core kernel instantiation the forward can select reproduces the first launch's bits, 30 launches each, at sizes
    that keep two workgroups per CU busy (thousands of tiles), and so does the batch-8 hipGraph bench.py times (30 replays).  Guards a hazard found in
    round 3: floating-point VALU work issued between a wave's own f16 MFMAs (the FPN tail's per-tile interpolation constants) came out wrong in lanes
    48-63 in ~1 of 500 tiles - only under full occupancy, so the small parity cases above could not see it.  The compiled schedules these
    launches validate are pinned by tests/golden/mfma_phase_fp_instructions.json (tests/test_device_code_lints.py, CPU)."""
    from casmvsnet_pl_amd.mvsnet import compose_fpn_tail
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    cases = []
    rnd = lambda *shape, amp=1.0: (torch.randn(*shape, generator=g) * amp)
    # conv0: the tiled kernel at cin 8 / 16 / 32, the warp-specialised z-marching kernel the regulariser runs, bf16 form (opt-in mode)
    for cin, (B, D, H, W) in ((8, (2, 8, 512, 640)), (16, (2, 32, 128, 160)), (32, (2, 48, 128, 160))):
        x0 = rnd(B, cin, D, H, W).to(dev)
        w0 = rnd(8, cin, 3, 3, 3, amp=0.1)
        p0 = ops.conv0_splitf16_pack(w0).to(dev)
        cases.append((f"conv0_sf<{cin}>", lambda p0=p0, x0=x0: ops.conv0_splitf16_forward(p0, x0)))
        cases.append((f"conv0_zw<{cin}>", lambda p0=p0, x0=x0: ops.conv0_zmarch_forward(p0, x0)))   # the warp-specialised z-march kernel: what the regulariser runs
        if cin == 16:
            pb = ops.conv0_splitbf16_pack(w0).to(dev)
            cases.append(("conv0_sb<16>", lambda pb=pb, x0=x0: ops.conv0_splitbf16_forward(pb, x0)))
    # conv2 / conv4 / conv6: every (channels, planes-per-tile) form - TZ 4, and TZ 2 for 2-plane volumes (conv4 at level 0)
    for c, (B, D, H, W) in ((16, (2, 16, 128, 160)), (16, (8, 2, 256, 320)), (32, (4, 8, 64, 80)), (32, (8, 2, 128, 160)), (64, (8, 6, 32, 40))):
        xc = rnd(B, c, D, H, W).to(dev)
        pc = ops.conv_ci_splitf16_pack(rnd(c, c, 3, 3, 3, amp=0.1)).to(dev)
        cases.append((f"conv_ci_sf<{c},{c},{2 if D <= 2 else 4}>", lambda pc=pc, xc=xc, c=c: ops.conv_ci_splitf16_forward(pc, xc, c)))
    # conv1 / conv3 (stride 2, z-marching)
    for cin, (B, D, H, W) in ((8, (2, 32, 256, 320)), (16, (4, 16, 128, 160))):
        xs = rnd(B, cin, D, H, W).to(dev)
        ps = ops.conv_s2_splitf16_pack(rnd(2 * cin, cin, 3, 3, 3, amp=0.1)).to(dev)
        cases.append((f"conv_s2_sf<{cin},{2 * cin}>", lambda ps=ps, xs=xs, cin=cin: ops.conv_s2_splitf16_forward(ps, xs, 2 * cin)))
    # conv9 / conv11 (transposed, with their skip tensors)
    x9, s9 = rnd(4, 32, 8, 64, 80).to(dev), rnd(4, 16, 16, 128, 160).to(dev)
    p9 = ops.deconv9_splitf16_pack(rnd(32, 16, 3, 3, 3, amp=0.1), torch.ones(16), torch.zeros(16)).to(dev)
    cases.append(("deconv9_sf", lambda: ops.deconv9_splitf16_forward(p9, x9, s9)))
    x11, s11 = rnd(2, 16, 16, 128, 160).to(dev), rnd(2, 8, 32, 256, 320).to(dev)
    p11 = ops.deconv11_splitf16_pack(rnd(16, 8, 3, 3, 3, amp=0.1), torch.ones(8), torch.zeros(8)).to(dev)
    cases.append(("deconv11_sf", lambda: ops.deconv11_splitf16_forward(p11, x11, s11)))
    # conv11 + prob + regression walking the depth axis (levels 1 / 0 of the cascade; level 2's 48 planes run inside the batch-8 graph below)
    for Di, Hi, Wi in ((16, 64, 80), (4, 128, 160)):
        xz, sz = rnd(2, 16, Di, Hi, Wi).to(dev), rnd(2, 8, 2 * Di, 2 * Hi, 2 * Wi).to(dev)
        dz = (400.0 + rnd(2, 2 * Di, 2 * Hi, 2 * Wi).abs()).to(dev)
        ppz = ops.conv3d_pack(ops.CONV_S1, rnd(1, 8, 3, 3, 3, amp=0.3), None, torch.zeros(1)).to(dev)
        cases.append((f"conv11_prob_zfused<{2 * Di}>", lambda xz=xz, sz=sz, dz=dz, ppz=ppz: torch.cat([t.flatten() for t in ops.conv11_prob_zfused(p11, ppz, xz, sz, dz)])))
    # FeatureNet: the fused tail and the three 2D channel-inner forms
    lw, lb = rnd(32, 8, 1, 1, amp=0.3), rnd(32)
    sw, sb = rnd(8, 32, 3, 3, amp=0.2), rnd(8)
    w40, bias9 = compose_fpn_tail(lw, lb, sw, sb)
    pf, b9 = ops.fpn_tail0_splitf16_pack(w40).to(dev), bias9.to(dev)
    xf, yf = rnd(6, 8, 512, 640).to(dev), rnd(6, 32, 256, 320).to(dev)
    cases.append(("fpn_tail0_sf", lambda: ops.fpn_tail0_splitf16(pf, b9, xf, yf)))
    for (cin, cout), (N, H, W) in (((16, 16), (6, 256, 320)), ((32, 32), (12, 128, 160)), ((32, 16), (6, 256, 320))):
        x2 = rnd(N, cin, H, W).to(dev)
        p2 = ops.conv2d_ci_splitf16_pack(rnd(cout, cin, 3, 3, amp=0.1)).to(dev)
        cases.append((f"conv2d_ci_sf<{cin},{cout}>", lambda p2=p2, x2=x2, cout=cout: ops.conv2d_ci_splitf16_forward(p2, x2, cout=cout)))
    # FeatureNet's stride-2 layers (conv1.0 / conv2.0)
    for cin, (N, H, W) in ((8, (6, 512, 640)), (16, (12, 256, 320))):
        x5 = rnd(N, cin, H, W).to(dev)
        p5 = ops.conv2d_k5s2_splitf16_pack(rnd(2 * cin, cin, 5, 5, amp=0.1)).to(dev)
        cases.append((f"conv2d_k5s2_sf<{cin},{2 * cin}>", lambda p5=p5, x5=x5, cin=cin: ops.conv2d_k5s2_splitf16_forward(p5, x5, 2 * cin)))
    # FeatureNet.conv0 as one kernel (both layers on the f16 cores)
    xm = rnd(6, 3, 512, 640).to(dev)
    pm = ops.fnet_conv0_mm_pack(rnd(8, 3, 3, 3, amp=0.3), None, None, rnd(8, 8, 3, 3, amp=0.2), None, None).to(dev)
    cases.append(("fnet_conv0_mm", lambda: ops.fnet_conv0_mm(pm, xm)))
    launches = 30
    bad = {}
    for name, fn in cases:
        ref = fn()
        bad[name] = sum(0 if torch.equal(fn(), ref) else 1 for _ in range(launches))
    del cases, ref
    torch.cuda.empty_cache()
    # the launch bench.py times: batch 8, one hipGraph replay per step
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.graph import GraphedForward
    from casmvsnet_pl_amd.synthetic import config_inputs, randomize_state_dict
    model = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=0)
    model = model.to(dev).eval()
    imgs, proj, dmin, dint = config_inputs("dtu_640x512_v3_var", 8, seed=0)
    gf = GraphedForward(model, imgs.to(dev), proj.to(dev), dmin, dint)
    ref = {k: v.clone() for k, v in gf().items()}
    bad["graph_batch8"] = 0
    for _ in range(launches):
        out = gf()
        bad["graph_batch8"] += 0 if all(torch.equal(out[k], ref[k]) for k in ref) else 1
    report("split_f16_bit_stability", launches=launches, differing=bad)
    assert len(bad) == 26 and not any(bad.values()), bad


@pytest.mark.parametrize("cin,shape", [(8, (2, 8, 48, 64)), (16, (1, 9, 17, 44)), (32, (1, 12, 32, 40))])
def test_conv0_zmarch_equals_the_tiled_kernel(dev, report, cin, shape):
    """conv0_zmarch.hip (input-stationary along z; the regulariser's conv0 at cin = 16) against the tiled split-f16 kernel on the same packed image, and
    both against the layer in float64: other staging units and summation grouping, the same float32-grade arithmetic."""
    ops = _ops()
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin)
    x = torch.randn(B, cin, D, H, W, generator=g) * 3
    w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    ref = F.conv3d(x.double(), w.double(), padding=1) * scale.double().view(1, 8, 1, 1, 1) + shift.double().view(1, 8, 1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * 0.01)
    packed = ops.conv0_splitf16_pack(w, scale, shift).to(dev)
    xd = x.to(dev)
    tiled = ops.conv0_splitf16_forward(packed, xd)
    got = ops.conv0_zmarch_forward(packed, xd)
    assert torch.equal(got, ops.conv0_zmarch_forward(packed, xd))
    e_zm, e_tiled = scaled_err(got.cpu(), ref), scaled_err(tiled.cpu(), ref)
    report("conv0_zmarch", cin=cin, shape=list(shape), err_zmarch=e_zm, err_tiled=e_tiled)
    assert e_zm < 2e-6 and e_zm < 3 * max(e_tiled, 2e-7)


@pytest.mark.parametrize("which,shape", [("deconv11", (2, 4, 12, 20)), ("deconv11", (1, 3, 5, 34)), ("deconv9", (2, 3, 6, 10)), ("deconv9", (1, 2, 5, 18))])
def test_deconv_splitf16_equals_the_layer(dev, report, which, shape):
    """conv9 / conv11 (ConvTranspose3d k3 s2 p1 op1 + folded ABN + leaky-relu + skip, mvsnet.py:80-86,99-101) on the f16 matrix cores against the layer in
    float64, beside the float32 MFMA kernel's own error."""
    ops = _ops()
    cin, cout = (16, 8) if which == "deconv11" else (32, 16)
    B, Di, Hi, Wi = shape
    g = torch.Generator().manual_seed(Di + Wi)
    x = torch.randn(B, cin, Di, Hi, Wi, generator=g) * 2
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.2
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    skip = torch.randn(B, cout, 2 * Di, 2 * Hi, 2 * Wi, generator=g)
    ref = F.conv_transpose3d(x.double(), w.double(), None, stride=2, padding=1, output_padding=1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * 0.01) + skip.double()
    f32 = ops.conv3d_forward(ops.CONV_T2, ops.conv3d_pack(ops.CONV_T2, w, sc, sh).to(dev), x.to(dev), cout, skip.to(dev)).cpu()
    pack, fwd = (ops.deconv11_splitf16_pack, ops.deconv11_splitf16_forward) if which == "deconv11" else (ops.deconv9_splitf16_pack, ops.deconv9_splitf16_forward)
    got = fwd(pack(w, sc, sh).to(dev), x.to(dev), skip.to(dev)).cpu()
    e_sf, e_f32 = scaled_err(got, ref), scaled_err(f32, ref)
    report("deconv_splitf16", which=which, shape=list(shape), err_splitf16=e_sf, err_f32_mfma=e_f32)
    assert torch.isfinite(got).all() and e_sf < 3e-6 and e_sf < 4 * max(e_f32, 2e-7)


@pytest.mark.parametrize("shape", [(1, 2, 5, 34), (2, 4, 9, 32), (1, 3, 8, 62), (1, 1, 1, 2), (2, 16, 24, 40), (8, 4, 32, 40)])
def test_conv11_prob_zfused_equals_the_three_layers(dev, report, shape):
    """csrc/conv11_prob_zfused.hip: conv11 (ConvTranspose3d 16 -> 8 k3 s2 p1 op1 + ABN + leaky-relu, + conv0's output), `prob` (Conv3d 8 -> 1 + bias) and the
    softmax regression (mvsnet.py:84-89,101,104,174-193) as ONE depth-walking kernel against the layers in float64 (cost, depth, confidence; index away from
    trunc() boundaries) and against the two kernels it replaces; image borders inside a tile, several tiles, 4 / 8 / 6 / 2 / 32 / 8 planes; twice for the bits."""
    ops = _ops()
    B, Di, Hi, Wi = shape
    D, h, w = 2 * Di, 2 * Hi, 2 * Wi
    g = torch.Generator().manual_seed(Di * 10 + Wi)
    x = torch.randn(B, 16, Di, Hi, Wi, generator=g) * 2
    x[:, :, ::2] *= 25.0                         # neighbouring input planes of very different magnitude: the two chains of an odd plane carry different scales
    skip = torch.randn(B, 8, D, h, w, generator=g)
    w11 = torch.randn(16, 8, 3, 3, 3, generator=g) * 0.2
    sc, sh = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    wp, bp = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.3, torch.randn(1, generator=g) * 0.1
    dv = (425.0 + 2.5 * torch.arange(D).view(1, D, 1, 1) + 0.01 * torch.rand(B, 1, h, w, generator=g)).expand(B, D, h, w).contiguous()
    u11 = F.conv_transpose3d(x.double(), w11.double(), None, stride=2, padding=1, output_padding=1) * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)
    u11 = torch.where(u11 > 0, u11, u11 * 0.01) + skip.double()
    cref = F.conv3d(u11, wp.double(), bp.double(), padding=1).squeeze(1)
    p = torch.softmax(cref, 1)
    dref = (p * dv.double()).sum(1)
    iref = (p * torch.arange(D, dtype=torch.float64).view(1, D, 1, 1)).sum(1)
    p11, pp = ops.deconv11_splitf16_pack(w11, sc, sh).to(dev), ops.conv3d_pack(ops.CONV_S1, wp, None, bp).to(dev)
    xd, sd, dd = x.to(dev), skip.to(dev), dv.to(dev)
    cost, depth, conf, index = ops.conv11_prob_zfused(p11, pp, xd, sd, dd, return_index=True)
    again = ops.conv11_prob_zfused(p11, pp, xd, sd, dd, return_index=True)
    assert all(torch.equal(a, b) for a, b in zip((cost, depth, conf, index), again))
    c2, d2, f2 = ops.prob_regress(pp, ops.deconv11_splitf16_forward(p11, xd, sd), dd)
    e_cost, e_two = scaled_err(cost.cpu(), cref), scaled_err(c2.cpu(), cref)
    e_depth = float(((depth.cpu().double() - dref).abs() / dref).max())
    settled = (iref - iref.round()).abs() > 1e-3
    idx_off = int(((index.cpu().long() != iref.floor().clamp(0, D - 1).long()) & settled).sum())
    report("conv11_prob_zfused", shape=list(shape), err_cost=e_cost, err_cost_two_kernels=e_two, err_depth_rel=e_depth, indices_off=idx_off,
           vs_two_kernels_depth_rel=float(((depth - d2).abs() / d2.abs()).max()))
    assert torch.isfinite(cost).all() and torch.isfinite(depth).all() and torch.isfinite(conf).all()
    assert e_cost < 3e-6 and e_cost < 4 * max(e_two, 2e-7) and e_depth < 2e-5 and idx_off == 0
    assert float((conf - f2).abs().max()) < 1e-2   # (a window of four probabilities: a trunc() boundary moves it)


@pytest.mark.parametrize("kernel", ["conv0_sf", "conv0_zm", "conv_ci_sf", "conv_s2_sf"])
@pytest.mark.parametrize("below", [12, 24, 34])
def test_split_f16_absolute_error_bound_far_below_the_unit_maximum(dev, report, kernel, below):
    """The split arithmetic scales a staged unit (tile / plane patch x 8 or 16 channels) by ONE power of two, so a voxel 2^-k below the unit's largest
    magnitude keeps fewer than 22 bits once its second slice falls into float16's subnormal range (k > 18) - RELATIVE accuracy is lost there, the ABSOLUTE
    error is not: each slice is rounded to a multiple of 2^-24 of the scaled unit, whose maximum sits in [2^14, 2^15), i.e. |x - (x_a + x_b) / 2^kx| <=
    2^-39 of the unit's maximum per element (round-3 verdict, weak #1: asserted here, not only documented).  One huge voxel per sample sets every unit's
    maximum; the outputs whose receptive fields do not contain it are sums of small products and must stay within
    taps x cin x max |w| x max(2^-39 x unit maximum, 2^-22 x their own magnitude) of the float64 layer (+ float32 rounding of the result itself)."""
    ops = _ops()
    cin, cout = {"conv0_sf": (8, 8), "conv0_zm": (16, 8), "conv_ci_sf": (16, 16), "conv_s2_sf": (8, 16)}[kernel]
    stride = 2 if kernel == "conv_s2_sf" else 1
    B, D, H, W = 1, 6, 8, 64
    g = torch.Generator().manual_seed(below)
    big = 1000.0
    x = torch.randn(B, cin, D, H, W, generator=g) * big * 2.0 ** -below
    x[:, :, :, 0, 0] = big                       # every (plane, channel) of every staged unit holds one voxel of the large magnitude
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.2
    scale, shift = torch.ones(cout), torch.zeros(cout)
    ref = F.conv3d(x.double(), w.double(), stride=stride, padding=1)
    ref = torch.where(ref > 0, ref, ref * 0.01)
    xd = x.to(dev)
    if kernel == "conv0_sf":
        got = ops.conv0_splitf16_forward(ops.conv0_splitf16_pack(w, scale, shift).to(dev), xd)
    elif kernel == "conv0_zm":
        got = ops.conv0_zmarch_forward(ops.conv0_splitf16_pack(w, scale, shift).to(dev), xd)
    elif kernel == "conv_ci_sf":
        got = ops.conv_ci_splitf16_forward(ops.conv_ci_splitf16_pack(w, scale, shift).to(dev), xd, cout)
    else:
        got = ops.conv_s2_splitf16_forward(ops.conv_s2_splitf16_pack(w, scale, shift).to(dev), xd, cout)
    got = got.cpu().double()
    far = torch.zeros_like(ref, dtype=torch.bool)
    far[..., 2:, 4:] = True                      # output rows / columns whose 3 x 3 x 3 receptive field (stride 1 or 2) cannot reach input (y, x) = (0, 0)
    err = (got - ref).abs()[far]
    rest = x.clone()
    rest[:, :, :, 0, 0] = 0
    small = float(rest.abs().max())             # per element: 2^-22 relative while both slices are normal float16 numbers, 2^-39 of the unit's maximum below that
    bound = 27 * cin * float(w.abs().max()) * max(big * 2.0 ** -39, small * 2.0 ** -22) + 2e-6 * ref.abs()[far]
    worst = float((err / bound).max())
    report("split_f16_absolute_bound", kernel=kernel, below_log2=below, worst_err_over_bound=worst, max_abs_err=float(err.max()), small_magnitude=big * 2.0 ** -below)
    assert torch.isfinite(got).all() and worst <= 1.0, worst
    # and next to the large voxel the result is float32-grade relative to ITS magnitude
    near = ~far
    assert float((got - ref).abs()[near].max()) <= 3e-6 * float(ref.abs()[near].max())


@pytest.mark.parametrize("kernel", ["conv0_sf", "conv0_zm", "conv_ci_sf"])
def test_split_f16_kernels_never_turn_non_finite_inputs_into_finite_wrong_values(dev, kernel):
    """Round-3 advisor finding (csrc/split_f16.h: tile_scale): a NaN voxel reaches the outputs whose matrix tile multiplies it (the per-tile maximum skips NaNs);
    an INFINITE voxel leaves its staged unit without a finite scaling, so the whole unit is poisoned: every output is either non-finite or bit-equal to the
    clean run's, and every output within the 3 x 3 x 3 reach of the bad voxel is non-finite - never a silently flushed finite value."""
    ops = _ops()
    g = torch.Generator().manual_seed(17)
    cin = {"conv0_sf": 8, "conv0_zm": 16, "conv_ci_sf": 16}[kernel]
    x = torch.randn(1, cin, 8, 32, 64, generator=g)
    if kernel == "conv_ci_sf":
        packed = ops.conv_ci_splitf16_pack(torch.randn(cin, cin, 3, 3, 3, generator=g) * 0.2).to(dev)
        run = lambda t: ops.conv_ci_splitf16_forward(packed, t.to(dev), cin).cpu()
    else:
        packed = ops.conv0_splitf16_pack(torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2).to(dev)
        fwd = ops.conv0_splitf16_forward if kernel == "conv0_sf" else ops.conv0_zmarch_forward
        run = lambda t: fwd(packed, t.to(dev)).cpu()
    clean = run(x)
    assert torch.isfinite(clean).all()
    z, y, xx = 4, 13, 37
    for bad in (float("nan"), float("inf"), float("-inf")):
        xb = x.clone()
        xb[0, 3, z, y, xx] = bad
        got = run(xb)
        finite = torch.isfinite(got)
        assert torch.equal(got[finite], clean[finite]), (kernel, bad)
        assert not finite[0, :, z - 1:z + 2, y - 1:y + 2, xx - 1:xx + 2].any(), (kernel, bad)
        if bad != bad:   # NaN: lost outputs stay inside the matrix tile's footprint around the voxel (the zero-weight K slots of the MFMA forms multiply
            touched = torch.zeros_like(finite)   # the bad value too - NaN x 0 = NaN -, in the float32 MFMA kernels as well), nothing farther away
            touched[0, :, z - 2:z + 3, y - 2:y + 3, xx - 8:xx + 9] = True
            assert bool(finite[~touched].all()), kernel
        assert float(finite.float().mean()) > 0.5, (kernel, bad)   # the poisoned unit is local


@pytest.mark.parametrize("kernel", ["conv_s2_sf<8>", "conv_s2_sf<16>", "conv11_prob_zfused"])
def test_round4_kernels_never_turn_non_finite_inputs_into_finite_wrong_values(dev, kernel):
    """The same property for the kernels added in round 4 (the stride-2 z-march, the fused tail): with one NaN / Inf voxel in the input every output value is
    either non-finite or bit-equal to the clean run's, and every output whose receptive field holds the voxel is non-finite (its reach computed with a
    ones-kernel convolution of the voxel's indicator).  A poisoned unit (plane patch x 8 channels) stays local."""
    ops = _ops()
    g = torch.Generator().manual_seed(23)
    if kernel.startswith("conv_s2"):
        cin = int(kernel[len("conv_s2_sf<"):-1])
        x = torch.randn(1, cin, 8, 32, 72, generator=g)
        packed = ops.conv_s2_splitf16_pack(torch.randn(2 * cin, cin, 3, 3, 3, generator=g) * 0.2).to(dev)
        run = lambda t: ops.conv_s2_splitf16_forward(packed, t.to(dev), 2 * cin).cpu()
        reach = lambda ind: F.conv3d(ind, torch.ones(1, 1, 3, 3, 3), stride=2, padding=1) > 0
        bad_at = (0, 3, 5, 13, 37)
    else:
        x = torch.randn(1, 16, 4, 16, 34, generator=g)
        skip = torch.randn(1, 8, 8, 32, 68, generator=g).to(dev)
        dv = (400.0 + torch.arange(8.0).view(1, 8, 1, 1).expand(1, 8, 32, 68)).contiguous().to(dev)
        p11 = ops.deconv11_splitf16_pack(torch.randn(16, 8, 3, 3, 3, generator=g) * 0.2, torch.ones(8), torch.zeros(8)).to(dev)
        pp = ops.conv3d_pack(ops.CONV_S1, torch.randn(1, 8, 3, 3, 3, generator=g) * 0.3, None, torch.zeros(1)).to(dev)
        run = lambda t: ops.conv11_prob_zfused(p11, pp, t.to(dev), skip, dv)[0].cpu().unsqueeze(1)   # the cost volume
        # conv11 spreads a voxel over outputs 2 i - 1 .. 2 i + 1, `prob` one more voxel each way
        reach = lambda ind: F.conv3d((F.conv_transpose3d(ind, torch.ones(1, 1, 3, 3, 3), stride=2, padding=1, output_padding=1) > 0).float(), torch.ones(1, 1, 3, 3, 3), padding=1) > 0
        bad_at = (0, 5, 2, 7, 19)
    clean = run(x)
    assert torch.isfinite(clean).all()
    ind = torch.zeros(1, 1, *x.shape[2:])
    ind[(0, 0) + bad_at[2:]] = 1.0
    hit = reach(ind)[0, 0]   # (Do, Ho, Wo)
    for bad in (float("nan"), float("inf"), float("-inf")):
        xb = x.clone()
        xb[bad_at] = bad
        got = run(xb)
        finite = torch.isfinite(got)
        assert torch.equal(got[finite], clean[finite]), (kernel, bad)
        assert not finite[0, :, hit].any(), (kernel, bad)
        assert float(finite.float().mean()) > 0.3, (kernel, bad)


def test_whole_forward_float32_layers_equal_the_split_f16_layers(dev, report):
    """The engine with every layer on the float32 MFMA kernels against the default layer set (conv0 - conv4 / 6 / 9 / 11 and eight FeatureNet layers on the
    f16 matrix cores) on one problem: depths agree to float32 rounding in the median and within the oracle bound everywhere."""
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict
    model = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=3)
    model = model.to(dev).eval()
    imgs, proj, dmin, dint = make_inputs(2, 3, 128, 160, seed=4)
    imgs, proj = imgs.to(dev), proj.to(dev)
    want = {k: v.clone() for k, v in model(imgs, proj, dmin, dint).items()}
    for l in range(3):
        getattr(model, f"cost_reg_{l}").conv0_mode = getattr(model, f"cost_reg_{l}").ci_mode = "f32"
    model.feature.tail_mode = "f32"
    got = model(imgs, proj, dmin, dint)
    assert model.cost_reg_0._conv0_active is None and not model.cost_reg_0._ci_active and not model.cost_reg_0._s2_active and not model.feature._split_active
    stats = {}
    for k in want:
        if k.startswith("depth"):
            rel = (got[k] - want[k]).abs() / want[k].abs()
            stats[k] = [float(rel.max()), float(rel.median())]
            assert float(rel.max()) < 1e-3 and float(rel.median()) < 2e-6, (k, stats[k])
    report("f32_vs_splitf16_layers", **stats)


def test_convbnrelu3d_module_runs_one_hip_layer(dev):
    """modules.py:21-31 called on its own (it was a raise stub)."""
    from casmvsnet_pl_amd import ABN
    from casmvsnet_pl_amd.modules import ConvBnReLU3D
    g = torch.Generator().manual_seed(9)
    for stride in (1, 2):
        m = ConvBnReLU3D(8, 16, stride=stride, norm_act=ABN).eval()
        with torch.no_grad():
            m.bn.running_var.uniform_(0.5, 1.5, generator=g)
            m.bn.running_mean.normal_(0, 0.1, generator=g)
            m.bn.weight.uniform_(0.6, 1.4, generator=g)
            m.bn.bias.normal_(0, 0.1, generator=g)
        x = torch.randn(1, 8, 8, 16, 24, generator=g)
        want = F.leaky_relu(F.batch_norm(F.conv3d(x, m.conv.weight, None, stride=stride, padding=1), m.bn.running_mean,
                                         m.bn.running_var, m.bn.weight, m.bn.bias, False, 0.0, m.bn.eps), 0.01)
        got = m.to(dev)(x.to(dev)).cpu()
        assert got.shape == want.shape and scaled_err(got, want) < 1.2e-5


def test_public_module_functions_match_oracle(dev, report):
    """The importable functions of models/modules.py (get_depth_values :34-49, depth_regression :95-104, homo_warp
    :52-92) through the host mirror, float and (B,1)-tensor intervals, (B,D,H,W) and (D,) depth values."""
    from casmvsnet_pl_amd import modules as M
    g = torch.Generator().manual_seed(77)
    B, D, H, W = 2, 32, 24, 40
    cur = 500.0 + 300.0 * torch.rand(B, 1, H, W, generator=g)
    cur[0, 0, 0, 0] = 2.0  # clamp_min(1e-7) branch
    for interval in (5.3, torch.tensor([[5.3], [4.1]])):
        want = R.get_depth_values(cur, D, interval)
        got = M.get_depth_values(cur.to(dev), D, interval if isinstance(interval, float) else interval.to(dev)).cpu()
        assert got.shape == want.shape and rel_err(got, want) < 3e-6
    p = F.softmax(torch.randn(B, D, H, W, generator=g) * 2, 1)
    dv = R.get_depth_values(cur, D, 5.3)
    for d in (dv, torch.linspace(425.0, 935.0, D)):
        want = R.depth_regression(p, d)
        got = M.depth_regression(p.to(dev), d.to(dev)).cpu()
        assert got.shape == want.shape
        report("depth_regression", per_plane=d.dim() == 1, rel=rel_err(got, want))
        assert rel_err(got, want) < 2e-6  # a D-term fp32 sum in a different order
    src = torch.randn(B, 16, H, W, generator=g)
    proj, _, _ = _proj_like(B, 2, H, W, seed=2)
    want = R.homo_warp(src, proj[:, 0], dv)
    got = M.homo_warp(src.to(dev), proj[:, 0].to(dev), dv.to(dev)).cpu()
    assert max_abs(got, want) < 1.5e-4


def _expected_index(cost):
    """e = sum_k softmax(cost)_k * k in float64 from the ORACLE's cost: where trunc(e) is discontinuous."""
    D = cost.shape[1]
    return (F.softmax(cost.double(), 1) * torch.arange(D, dtype=torch.float64).view(1, D, 1, 1)).sum(1)


def _check_levels(report, name, got, model, want, want_index, want_cost, extra=None, boundary_tol=1e-3):
    """Asserts the per-level parity contract of the module docstring and reports every index flip."""
    stats, flips, mism_pixels = dict(extra or {}), [], {}
    for l in (2, 1, 0):
        d, c = got[f"depth_{l}"].cpu(), got[f"confidence_{l}"].cpu()
        mism_pixels[l] = d.numel()
        gi, wi = model.last_index[l].cpu().long(), want_index[l]
        mism = gi != wi
        e = _expected_index(want_cost[l])
        dist = (e - e.round()).abs()
        stats[f"depth_rel_{l}"] = rel_err(d, want[f"depth_{l}"])
        stats[f"index_match_{l}"] = float((~mism).float().mean())
        stats[f"index_flips_{l}"] = int(mism.sum())
        stats[f"conf_abs_{l}"] = max_abs(c[~mism], want[f"confidence_{l}"][~mism]) if (~mism).any() else 0.0
        stats[f"flip_max_boundary_dist_{l}"] = float(dist[mism].max()) if mism.any() else 0.0
        stats[f"pixels_within_tol_of_boundary_{l}"] = int((dist < boundary_tol).sum())
        # (b) a flipped pixel is not exempt from the confidence check: there the engine must return the ORACLE's 4-plane window sum (mvsnet.py:179-183:
        # 4 * avg_pool3d of pad(1, 2) = p[i-1] + p[i] + p[i+1] + p[i+2]) evaluated at the ENGINE's index - the window moved by one plane, nothing else
        if mism.any():
            p = F.softmax(want_cost[l].double(), 1)
            pp = F.pad(p, (0, 0, 0, 0, 1, 2))
            sum4 = pp[:, :-3] + pp[:, 1:-2] + pp[:, 2:-1] + pp[:, 3:]
            at_engine_index = sum4.gather(1, gi.unsqueeze(1)).squeeze(1)
            stats[f"conf_abs_on_flips_{l}"] = float((c.double() - at_engine_index)[mism].abs().max())
        else:
            stats[f"conf_abs_on_flips_{l}"] = 0.0
        if l in model.last_cost:  # the engine's own cost volume: its error, and the noise it puts on e
            gc = model.last_cost[l].cpu()
            stats[f"cost_scaled_err_{l}"] = scaled_err(gc, want_cost[l])
            stats[f"e_noise_{l}"] = float((_expected_index(gc) - e).abs().max())
            # (a) north_star's words are "identical argmax-depth indices": argmax_D of the engine's cost volume against argmax_D of the oracle's.  The argmax
            # of two volumes that differ by at most E everywhere can only differ where the oracle's best two planes are closer than 2 E.
            cost_err = float((gc.double() - want_cost[l].double()).abs().max())
            top2 = want_cost[l].double().topk(2, dim=1).values
            gap = top2[:, 0] - top2[:, 1]
            am = gc.argmax(1) != want_cost[l].argmax(1)
            stats[f"argmax_match_{l}"] = float((~am).float().mean())
            stats[f"argmax_flips_{l}"] = int(am.sum())
            stats[f"argmax_flip_max_gap_{l}"] = float(gap[am].max()) if am.any() else 0.0
            stats[f"cost_abs_err_{l}"] = cost_err
            stats[f"pixels_with_top2_gap_below_2x_cost_err_{l}"] = int((gap <= 2 * cost_err).sum())
        for b, y, x in mism.nonzero()[:32].tolist():
            flips.append(dict(level=l, b=b, y=y, x=x, got=int(gi[b, y, x]), want=int(wi[b, y, x]),
                              expected_index=float(e[b, y, x]), boundary_dist=float(dist[b, y, x])))
    stats["flips"] = flips
    report(name, **stats)
    for l in (2, 1, 0):
        assert stats[f"depth_rel_{l}"] < 1e-4, (l, stats[f"depth_rel_{l}"])          # bar 1e-3, measured <= 5.4e-5
        assert stats[f"flip_max_boundary_dist_{l}"] < boundary_tol, (l, flips)        # every flip sits ON a trunc() boundary
        assert stats[f"index_flips_{l}"] <= stats[f"pixels_within_tol_of_boundary_{l}"]
        assert stats[f"conf_abs_{l}"] < 5 * boundary_tol, (l, stats[f"conf_abs_{l}"])  # measured 1.4e-4 (1.6e-3 for gwc8)
        assert stats[f"conf_abs_on_flips_{l}"] < 0.5 * boundary_tol, (l, stats[f"conf_abs_on_flips_{l}"])   # 5e-4: the window moved, the probabilities did not
        if f"argmax_flips_{l}" in stats:
            assert stats[f"argmax_flip_max_gap_{l}"] <= 2 * stats[f"cost_abs_err_{l}"], (l, stats[f"argmax_flip_max_gap_{l}"], stats[f"cost_abs_err_{l}"])
            assert stats[f"argmax_flips_{l}"] <= stats[f"pixels_with_top2_gap_below_2x_cost_err_{l}"]
            assert stats[f"argmax_flips_{l}"] <= max(3, 5e-4 * mism_pixels[l]), (l, stats[f"argmax_match_{l}"], stats[f"argmax_flips_{l}"])
        if f"cost_scaled_err_{l}" in stats:
            # level 2 is pure kernel error (measured 5e-6); a finer level's cost also carries the coarser levels' depth
            # error - its hypotheses are shifted by it - (measured up to 6.5e-4 at 1152x864)
            assert stats[f"cost_scaled_err_{l}"] < (5e-5 if l == 2 else 5e-3), (l, stats[f"cost_scaled_err_{l}"])
            assert stats[f"flip_max_boundary_dist_{l}"] <= stats[f"e_noise_{l}"] + 1e-5   # a flip needs e to cross the boundary
    return stats


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_end_to_end_matches_golden_fixture(dev, report, case):
    """CascadeMVSNet on the MI355X vs the REAL reference's CPU forward (committed fixture)."""
    g = Golden(case)
    imgs, proj = g.inputs()
    model = g.model(dev)
    model.keep_index = model.keep_cost = True
    res = model(imgs.to(dev), proj.to(dev), g.init_depth_min, g.depth_interval)
    # reference index, recomputed from the fixture's cost volumes with the oracle
    want_idx = {l: R.softmax_regress(g.t(f"cost_{l}"), g.t(f"depth_values_{l}"))[2] for l in (2, 1, 0)}
    want = {f"{k}_{l}": g.t(f"{k}_{l}") for k in ("depth", "confidence") for l in (2, 1, 0)}
    _check_levels(report, "e2e_golden", res, model, want, want_idx, {l: g.t(f"cost_{l}") for l in (2, 1, 0)},
                  extra=dict(case=case))


@pytest.mark.parametrize("G", [1, 8])
def test_end_to_end_tensor_depth_range_batch2(dev, report, G):
    """(B,1) tensor init_depth_min / depth_interval, as train.py's DataLoader passes them (B = 2)."""
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict
    model = CascadeMVSNet(num_groups=G, norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=20 + G)
    imgs, proj, dmin, dint = make_inputs(2, 3, 64, 96, seed=5)
    dmin_t = torch.tensor([[dmin], [dmin + 25.0]])
    dint_t = torch.tensor([[dint], [dint * 0.9]])
    want = R.cascade_forward(model.state_dict(), imgs, proj, dmin_t, dint_t, (8, 32, 48), (1, 2, 4), G)
    model = model.to(dev).eval()
    got = model(imgs.to(dev), proj.to(dev), dmin_t.to(dev), dint_t.to(dev))
    errs = {k: rel_err(got[k].cpu(), want[k]) for k in want if k.startswith("depth")}
    report("e2e_tensor_range", G=G, **errs)
    assert max(errs.values()) < 1e-4  # bar 1e-3, measured 5e-6


@pytest.mark.parametrize("config", ["dtu_640x512_v3_var", "dtu_640x512_v3_gwc8", "dtu_1152x864_v5_var", "blended_768x576_v7_var"])
def test_full_size_config_matches_oracle(dev, report, config):
    """Every BASELINE workload at its full size against the oracle (a few seconds to ~half a minute of CPU each):
    depth at all three levels, confidence on equal-index pixels, and every index flip on a trunc() boundary."""
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.synthetic import CONFIGS, config_inputs, randomize_state_dict
    H, W, V, G, n_depths, ratios, _ = CONFIGS[config]
    model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=0)
    imgs, proj, dmin, dint = config_inputs(config, 1, seed=0)  # the BlendedMVS config with its scaled depth interval
    want, inter = R.cascade_forward(model.state_dict(), imgs, proj, dmin, dint, n_depths, ratios, G, return_intermediates=True)
    model = model.to(dev).eval()
    model.keep_index = model.keep_cost = True
    got = model(imgs.to(dev), proj.to(dev), dmin, dint)
    tol = 1e-2 if G == 8 else 1e-3   # G = 8: one channel per group at level 0 (module docstring)
    split = _check_levels(report, "e2e_full_size", got, model, want, {l: inter[f"index_{l}"] for l in (2, 1, 0)},
                          {l: inter[f"cost_{l}"] for l in (2, 1, 0)}, extra=dict(config=config, layers="split-f16 (default)"), boundary_tol=tol)
    # (c) which arithmetic owns the noise on the expected index e: the same forward with EVERY layer on the float32 MFMA kernels, same checks, same report
    for l in range(3):
        getattr(model, f"cost_reg_{l}").conv0_mode = getattr(model, f"cost_reg_{l}").ci_mode = "f32"
    model.feature.tail_mode = "f32"
    got32 = model(imgs.to(dev), proj.to(dev), dmin, dint)
    assert model.cost_reg_0._conv0_active is None and not model.cost_reg_0._ci_active and not model.feature._split_active
    f32 = _check_levels(report, "e2e_full_size_all_float32", got32, model, want, {l: inter[f"index_{l}"] for l in (2, 1, 0)},
                        {l: inter[f"cost_{l}"] for l in (2, 1, 0)}, extra=dict(config=config, layers="all float32 MFMA"), boundary_tol=tol)
    report("e2e_full_size_noise_owner", config=config,
           **{f"{k}_{l}": [split[f"{k}_{l}"], f32[f"{k}_{l}"]] for l in (2, 1, 0) for k in ("index_flips", "e_noise", "argmax_flips", "cost_scaled_err")})


def test_benched_launch_matches_oracle(dev, report):
    """The launch bench.py times - dtu_640x512_v3_var at batch 8, ONE hipGraph replay (graph.GraphedForward) - against the
    oracle.  At batch 8 the regulariser selects kernels the batch-1 case above never runs (the f16 forms of conv4 / conv6 need
    >= 100 tiles, csrc/conv3d_mfma.hip `costreg_regress`), so the engine == oracle claim of the headline number rests on this
    test.  The oracle runs on two of the eight depth maps (the samples are independent: eval.py:213, train.py:85-97); the
    other six are checked for finite depths inside the hypothesis range."""
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.graph import GraphedForward
    from casmvsnet_pl_amd.synthetic import CONFIGS, config_inputs, randomize_state_dict
    config, B = "dtu_640x512_v3_var", 8
    H, W, V, G, n_depths, ratios, _ = CONFIGS[config]
    model = CascadeMVSNet(n_depths=list(n_depths), interval_ratios=list(ratios), num_groups=G, norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=0)
    imgs, proj, dmin, dint = config_inputs(config, B, seed=0)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()
    model.keep_index = model.keep_cost = True
    gf = GraphedForward(model, imgs.to(dev), proj.to(dev), dmin, dint)
    got = {k: v.clone() for k, v in gf(imgs.to(dev), proj.to(dev)).items()}
    torch.cuda.synchronize()
    index = {l: model.last_index[l].clone() for l in (2, 1, 0)}
    cost = {l: model.last_cost[l].clone() for l in (2, 1, 0)}

    class _Sample:   # what _check_levels reads from a model, restricted to one sample of the batch
        pass
    for b in (0, 5):
        want, inter = R.cascade_forward(sd, imgs[b:b + 1], proj[b:b + 1], dmin, dint, n_depths, ratios, G, return_intermediates=True)
        view = _Sample()
        view.last_index = {l: index[l][b:b + 1] for l in (2, 1, 0)}
        view.last_cost = {l: cost[l][b:b + 1] for l in (2, 1, 0)}
        _check_levels(report, "e2e_benched_launch", {k: v[b:b + 1] for k, v in got.items()}, view, want,
                      {l: inter[f"index_{l}"] for l in (2, 1, 0)}, {l: inter[f"cost_{l}"] for l in (2, 1, 0)},
                      extra=dict(config=config, batch=B, sample=b, launch="one hipGraph replay"))
    d0 = got["depth_0"]
    assert bool(torch.isfinite(d0).all()) and float(d0.min()) > dmin - 1.0 and float(d0.max()) < dmin + dint * 4 * 48 + 200.0
