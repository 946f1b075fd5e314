"""GPU parity cases of the kernels that were written without a GPU run at the end of round 3.  They are NOT part of the default GPU suite:
an unvalidated kernel must not be able to hang or fail the driver's run.  Enable them with CASMVS_TEST_EXPERIMENTAL=1 once the torch-free first tests
(tools/native/*_check) have passed on the MI355X; each case compares an experimental entry with the established path it would replace, and the whole
model with `experimental` sets against the default layer set."""
import os

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("CASMVS_TEST_EXPERIMENTAL") != "1",
                                                  reason="experimental kernels: set CASMVS_TEST_EXPERIMENTAL=1 after their native first tests passed")]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


@pytest.mark.parametrize("cin,shape", [(8, (2, 8, 48, 64)), (16, (1, 9, 17, 44)), (32, (1, 12, 32, 40))])
def test_conv0_zmarch_equals_the_tiled_kernel(dev, cin, shape):
    from casmvsnet_pl_amd import ops
    B, D, H, W = shape
    g = torch.Generator().manual_seed(cin)
    x = (torch.randn(B, cin, D, H, W, generator=g) * 3).to(dev)
    w = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.2
    packed = ops.conv0_splitf16_pack(w, torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1).to(dev)
    want = ops.conv0_splitf16_forward(packed, x)
    got = ops.conv0_zmarch_forward(packed, x)
    assert torch.equal(got, ops.conv0_zmarch_forward(packed, x)) and _rel(got, want) < 2e-6
    # both kernels on the tile grid shifted by 4 voxels in x (a mostly empty first tile column, other per-tile scalings)
    for shifted in (ops.conv0_splitf16_forward(packed, x, x_offset=4), ops.conv0_zmarch_forward(packed, x, x_offset=4)):
        assert bool(torch.isfinite(shifted).all()) and _rel(shifted, want) < 2e-6


@pytest.mark.parametrize("which,shape", [("deconv11", (2, 4, 12, 20)), ("deconv11", (1, 3, 5, 34)), ("deconv9", (2, 3, 6, 10)), ("deconv9", (1, 2, 5, 18))])
def test_deconv_splitf16_equals_the_float32_layer(dev, which, shape):
    from casmvsnet_pl_amd import ops
    cin, cout = (16, 8) if which == "deconv11" else (32, 16)
    B, Di, Hi, Wi = shape
    g = torch.Generator().manual_seed(Di + Wi)
    x = (torch.randn(B, cin, Di, Hi, Wi, generator=g) * 2).to(dev)
    w = torch.randn(cin, cout, 3, 3, 3, generator=g) * 0.2
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    skip = torch.randn(B, cout, 2 * Di, 2 * Hi, 2 * Wi, generator=g).to(dev)
    want = ops.conv3d_forward(ops.CONV_T2, ops.conv3d_pack(ops.CONV_T2, w, sc, sh).to(dev), x, cout, skip)
    pack, fwd = (ops.deconv11_splitf16_pack, ops.deconv11_splitf16_forward) if which == "deconv11" else (ops.deconv9_splitf16_pack, ops.deconv9_splitf16_forward)
    got = fwd(pack(w, sc, sh).to(dev), x, skip)
    assert _rel(got, want) < 3e-6


@pytest.mark.parametrize("shape", [(2, 48, 64), (1, 33, 44)])
def test_fnet_conv0_fused_equals_the_two_layers(dev, shape):
    from casmvsnet_pl_amd import ops
    N, H, W = shape
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, 3, H, W, generator=g).to(dev)
    w0, w1 = torch.randn(8, 3, 3, 3, generator=g) * 0.3, torch.randn(8, 8, 3, 3, generator=g) * 0.2
    s0, b0, s1, b1 = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1, torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    mid = ops.conv2d_forward(ops.CONV2D_K3, ops.conv2d_pack(ops.CONV2D_K3, w0, s0, b0).to(dev), x, 8)
    want = ops.conv2d_forward(ops.CONV2D_K3, ops.conv2d_pack(ops.CONV2D_K3, w1, s1, b1).to(dev), mid, 8)
    got = ops.fnet_conv0_fused(ops.fnet_conv0_fused_pack(w0, s0, b0, w1, s1, b1).to(dev), x)
    assert _rel(got, want) < 5e-6


@pytest.mark.parametrize("exp_cost,exp_feat", [({"zmarch"}, set()), ({"deconv9", "deconv11"}, set()), ({"tail"}, set()), (set(), {"conv0_fused"}),
                                               ({"xshift"}, set()), ({"zmarch32", "xshift", "deconv9", "tail"}, {"conv0_fused"})])
def test_whole_forward_with_experimental_layers_equals_the_default(dev, exp_cost, exp_feat):
    from casmvsnet_pl_amd import ABN, CascadeMVSNet
    from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict
    model = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(model.state_dict(), seed=3)
    model = model.to(dev).eval()
    imgs, proj, dmin, dint = make_inputs(2, 3, 128, 160, seed=4)
    imgs, proj = imgs.to(dev), proj.to(dev)
    want = {k: v.clone() for k, v in model(imgs, proj, dmin, dint).items()}
    for l in range(3):
        getattr(model, f"cost_reg_{l}").experimental = set(exp_cost)
    model.feature.experimental = set(exp_feat)
    got = model(imgs, proj, dmin, dint)
    for k in want:
        if k.startswith("depth"):
            assert float(((got[k] - want[k]).abs() / want[k].abs()).max()) < 1e-3, k   # the bound of the oracle comparison (depth_0: 1e-3 relative)
            assert float(((got[k] - want[k]).abs() / want[k].abs()).median()) < 2e-6, k


@pytest.mark.parametrize("kind_name,shape,cout", [("S1", (2, 8, 8, 12, 20), 8), ("S1", (1, 16, 4, 16, 32), 16), ("T2", (1, 16, 2, 6, 12), 8), ("K3", (2, 16, 18, 36), 16)])
def test_weight_gradient_with_the_conflict_free_lds_layout_is_bit_identical(dev, kind_name, shape, cout):
    """casmvs_conv_wgrad_x_f32(lds_layout = 1): operand tiles with channel strides = 2 (mod 32) - the same sums in the same order."""
    from casmvsnet_pl_amd import training
    from casmvsnet_pl_amd._lib import CONV2D_K3, CONV_S1, CONV_T2
    kind = {"S1": CONV_S1, "T2": CONV_T2, "K3": CONV2D_K3}[kind_name]
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(dev)
    cin = shape[1]
    out_shape = (shape[0], cout, *[2 * d for d in shape[2:]]) if kind_name == "T2" else (shape[0], cout, *shape[2:])
    gy = torch.randn(*out_shape, generator=g).to(dev)
    wshape = (cin, cout, 3, 3, 3) if kind_name == "T2" else ((cout, cin, 3, 3, 3) if kind_name == "S1" else (cout, cin, 3, 3))
    prob = training.PROB_WGRAD_KERNEL
    training.PROB_WGRAD_KERNEL = False
    try:
        training.WGRAD_LDS_LAYOUT = 0
        want = training.conv_wgrad(kind, x, gy, wshape)
        training.WGRAD_LDS_LAYOUT = 1
        got = training.conv_wgrad(kind, x, gy, wshape)
    finally:
        training.WGRAD_LDS_LAYOUT = 0
        training.PROB_WGRAD_KERNEL = prob
    assert torch.equal(got, want)
