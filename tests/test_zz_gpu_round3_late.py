"""GPU tests of what the last session of round 3 added after the round's GPU minutes were spent.  The kernels were checked on the MI355X
by torch-free C-ABI programs (tools/native/*.cpp, tools/notorch/: profiles/r03_final_native_checks.txt); these tests put the same
claims into the suite - through the Python host code, which those programs do not touch.  The file sorts last on purpose: the driver
runs `pytest -x`, and nothing here should be able to hide a result of the established tests."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape", [(2, 5, 12, 20), (1, 3, 7, 4), (1, 48, 32, 40), (1, 8, 128, 160)])
def test_prob_weight_gradient_kernel_equals_autograd_and_the_matrix_core_kernel(dev, shape):
    """training.conv_wgrad routes the `prob` layer (Conv3d 8 -> 1) to csrc/prob_wgrad.hip: against torch autograd in float64 and
    against the generic kernel it replaces; twice for bit-reproducibility (fixed-order reductions, no atomics)."""
    from casmvsnet_pl_amd import training
    from casmvsnet_pl_amd._lib import CONV_S1
    B, D, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + D)
    x = (torch.randn(B, 8, D, H, W, generator=g) + 0.3).to(dev)
    gy = (0.01 * torch.randn(B, 1, D, H, W, generator=g)).to(dev)
    assert training.PROB_WGRAD_KERNEL
    new = training.conv_wgrad(CONV_S1, x, gy, (1, 8, 3, 3, 3))
    again = training.conv_wgrad(CONV_S1, x, gy, (1, 8, 3, 3, 3))
    training.PROB_WGRAD_KERNEL = False
    try:
        old = training.conv_wgrad(CONV_S1, x, gy, (1, 8, 3, 3, 3))
    finally:
        training.PROB_WGRAD_KERNEL = True
    w = torch.zeros(1, 8, 3, 3, 3, dtype=torch.float64, requires_grad=True)   # the reference on the CPU, in float64 (as the other per-op tests: CPU autograd)
    torch.nn.functional.conv3d(x.cpu().double(), w, padding=1).backward(gy.cpu().double())
    scale = float(w.grad.abs().max())
    assert torch.equal(new, again)
    assert float((new.cpu().double() - w.grad).abs().max()) < 3e-6 * scale
    assert float((new - old).abs().max()) < 2e-5 * scale


def test_fusion_paired_tap_kernel_equals_the_one_tap_per_load_kernel(dev):
    """casmvs_fuse_reference_view_paired (the default of fusion.fuse_reference_view) against casmvs_fuse_reference_view: every
    output, the per-view ones included, bit for bit - on a scene with zero-depth pixels, outliers and taps outside every border."""
    from casmvsnet_pl_amd import fusion
    from oracle import fusion_scene
    for H, W, S, seed in ((64, 96, 4, 0), (48, 64, 1, 1), (130, 162, 6, 2)):
        Ps, depths, images, proba = fusion_scene.scene(H=H, W=W, S=S, seed=seed)
        pr = proba if H % 4 == 0 and W % 4 == 0 else None
        outs = [fusion.fuse_reference_view(depths[0], images[0], pr, Ps[0], depths[1:], images[1:], Ps[1:], conf=0.4, min_geo_consistent=min(S, 3),
                                           return_per_view=True, paired_taps=p) for p in (True, False)]
        torch.cuda.synchronize()
        for k in outs[0]:
            a, b = outs[0][k], outs[1][k]
            if a.dtype.is_floating_point:   # NaN-safe bit comparison
                assert torch.equal(a.view(torch.int64 if a.dtype == torch.float64 else torch.int32), b.view(torch.int64 if b.dtype == torch.float64 else torch.int32)), (k, H, W, S)
            else:
                assert torch.equal(a, b), (k, H, W, S)


@pytest.mark.order_tier(1)   # a subprocess: system tier (tests/conftest.py)
def test_torch_free_step_runner_runs_the_forward(dev):
    """tools/notorch/step_runner.py: the forward's library calls in mvsnet.py's order from a process without torch; depths finite and
    inside the hypothesis range, every stage timed."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "notorch", "step_runner.py"), "--batch", "2", "--hw", "128", "160", "--steps", "3", "--warmup", "1"],
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-800:], out.stderr[-1500:])
    assert "depth maps/s" in out.stdout and " ok" in out.stdout.splitlines()[-1]


def test_prefetcher_with_stager_thread_delivers_the_same_batches(dev):
    """DevicePrefetcher(threaded=True): staging (pinned copy, H2D enqueue, normalisation launch) on its own thread - the same batches in the
    same order, bit for bit, as the in-line form; an exception of the source iterator reaches the consumer."""
    from casmvsnet_pl_amd import pipeline as P
    g = torch.Generator().manual_seed(3)
    batches = [dict(imgs_u8=torch.randint(0, 256, (2, 3, 32, 48, 3), generator=g, dtype=torch.uint8), proj_mats=torch.randn(2, 2, 3, 3, 4, generator=g),
                    init_depth_min=torch.full((2, 1), 400.0 + i), depth_interval=torch.full((2, 1), 2.5), idx=[i, i]) for i in range(7)]
    want = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in P.DevicePrefetcher(batches, dev, depth=2)]
    got = [{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()} for b in P.DevicePrefetcher(batches, dev, depth=2, threaded=True)]
    torch.cuda.synchronize()
    assert len(got) == len(want) == 7
    for a, b in zip(got, want):
        assert a.keys() == b.keys() and a["idx"] == b["idx"]
        for k in a:
            if isinstance(a[k], torch.Tensor):
                assert a[k].is_cuda and torch.equal(a[k], b[k]), k

    def broken():
        yield batches[0]
        raise RuntimeError("source failed")
    it = P.DevicePrefetcher(broken(), dev, depth=2, threaded=True)
    assert next(it)["idx"] == [0, 0]
    with pytest.raises(RuntimeError, match="source failed"):
        next(it)
