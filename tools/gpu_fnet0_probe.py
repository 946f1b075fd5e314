"""FeatureNet.conv0 (ConvBnReLU 3 -> 8 -> 8) alone: the two float32-MFMA layer launches the engine runs today against the fused kernel with both layers on
the f16 matrix cores (fnet_conv0_mm.hip); error of each against a float64 convolution, us per call with dirtied caches.
   python tools/gpu_fnet0_probe.py [H W [N]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from casmvsnet_pl_amd import ops

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
dev = torch.device("cuda:0")
dirty = torch.empty(512 * 262144, device=dev)
SLOPE = 0.01


def timed(fn, reps=8):
    for _ in range(2):
        fn()
    tot = 0.0
    for _ in range(reps):
        dirty.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / reps * 1e3


g = torch.Generator().manual_seed(0)
w0, w1 = torch.randn(8, 3, 3, 3, generator=g) * 0.3, torch.randn(8, 8, 3, 3, generator=g) * 0.2
sc0, sh0 = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.2
sc1, sh1 = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.2
p0 = ops.conv2d_pack(ops.CONV2D_K3, w0, sc0, sh0).to(dev)
p1 = ops.conv2d_pack(ops.CONV2D_K3, w1, sc1, sh1).to(dev)
pm = ops.fnet_conv0_mm_pack(w0, sc0, sh0, w1, sc1, sh1).to(dev)


def ref64(x):
    x = x.double().cpu()
    y = F.leaky_relu(F.conv2d(x, w0.double(), padding=1) * sc0.double().view(1, 8, 1, 1) + sh0.double().view(1, 8, 1, 1), SLOPE)
    return F.leaky_relu(F.conv2d(y, w1.double(), padding=1) * sc1.double().view(1, 8, 1, 1) + sh1.double().view(1, 8, 1, 1), SLOPE)


def two(x):
    return ops.conv2d_forward(ops.CONV2D_K3, p1, ops.conv2d_forward(ops.CONV2D_K3, p0, x, 8, slope=SLOPE), 8, slope=SLOPE)


for (n, h, w) in ((2, 22, 36), (1, 40, 64), (3, 60, 90)):   # small shapes with ragged tiles: against float64
    x = torch.randn(n, 3, h, w, generator=g).to(dev)
    r = ref64(x)
    a, b = two(x).double().cpu(), ops.fnet_conv0_mm(pm, x, SLOPE).double().cpu()
    rng = float(r.abs().max())
    print(f"N {n} {h}x{w}: max error / range  two float32-MFMA layers {float((a - r).abs().max()) / rng:.2e}   fused f16 kernel {float((b - r).abs().max()) / rng:.2e}", flush=True)
for N in ([int(sys.argv[3])] if len(sys.argv) > 3 else [3, 24]):
    x = torch.randn(N, 3, H, W, generator=g).to(dev)
    t2 = timed(lambda: two(x))
    tm = timed(lambda: ops.fnet_conv0_mm(pm, x, SLOPE))
    a, b = two(x), ops.fnet_conv0_mm(pm, x, SLOPE)
    byt = 4 * (x.numel() + a.numel())
    print(f"N {N} {H}x{W}: two layers {t2:.1f} us, fused {tm:.1f} us ({byt / tm / 1e3:.0f} GB/s of its {byt / 1e6:.0f} MB), max diff / range {float((a - b).abs().max() / a.abs().max()):.1e}", flush=True)
    c = ops.fnet_conv0_mm(pm, x, SLOPE)
    print("   bit-stable run to run:", bool(torch.equal(b, c)))
