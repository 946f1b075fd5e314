"""CostRegNet.conv0 alone at the three cascade-level shapes: the float32-MFMA kernel (conv16db_kernel<PX>) against the
split-bf16 kernel (conv0_sb_kernel, 6 and 9 partial products) and the split-f16 kernel (conv0_sf_kernel, 3 and 4), each launch
timed on its own with dirtied caches; errors against a float64 convolution of a sub-volume.
   python tools/gpu_conv0_probe.py [H W [batch]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmvsnet_pl_amd import ops

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
dirty = torch.empty(512 * 262144, device=dev)


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


print(f"conv0 probe {H}x{W} batch {B}: us per launch (dirtied caches), TFLOP/s of the layer's 2*27*cin*8 FLOP per voxel")
for l, (cin, D) in ((2, (32, 48)), (1, (16, 32)), (0, (8, 8))):
    h, w = H >> l, W >> l
    g = torch.Generator().manual_seed(l)
    x = torch.randn(B, cin, D, h, w, generator=g).to(dev)
    wt = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.1
    sc, sh = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1
    p32 = ops.conv3d_pack(ops.CONV_S1, wt, sc, sh).to(dev)
    psb = ops.conv0_splitbf16_pack(wt, sc, sh).to(dev)
    gf = 2 * 27 * cin * 8 * B * D * h * w / 1e9
    t32 = timed(lambda: ops.conv3d_forward(ops.CONV_S1, p32, x, 8))
    t6 = timed(lambda: ops.conv0_splitbf16_forward(psb, x, terms=6))
    t9 = timed(lambda: ops.conv0_splitbf16_forward(psb, x, terms=9))
    psf = ops.conv0_splitf16_pack(wt, sc, sh).to(dev)
    t3 = timed(lambda: ops.conv0_splitf16_forward(psf, x, terms=3))
    t4 = timed(lambda: ops.conv0_splitf16_forward(psf, x, terms=4))
    # float64 truth on a sub-volume (all of D, 24 x 40 pixels; its border outputs are excluded: they see the crop's zero padding)
    xs = x[:1, :, :, :24, :40]
    ref = torch.nn.functional.conv3d(xs.double(), wt.to(dev).double(), None, padding=1) * sc.to(dev).double().view(1, -1, 1, 1, 1) + sh.to(dev).double().view(1, -1, 1, 1, 1)
    ref = torch.where(ref > 0, ref, ref * 0.01)[..., :, 1:-1, 1:-1]
    errs = []
    for fn in (lambda v: ops.conv3d_forward(ops.CONV_S1, p32, v, 8), lambda v: ops.conv0_splitbf16_forward(psb, v, terms=6), lambda v: ops.conv0_splitf16_forward(psf, v, terms=3)):
        got = fn(x)[:1, :, :, :24, :40][..., :, 1:-1, 1:-1].double()
        errs.append(float((got - ref).abs().max() / ref.abs().max()))
    print(f"level {l} (cin {cin}, D {D}, {h}x{w}): f32 MFMA {t32:.1f} ({gf / t32 * 1e3:.0f} TF/s)  split-bf16 x6 {t6:.1f} ({gf / t6 * 1e3:.0f} TF/s)  x9 {t9:.1f}  "
          f"split-f16 x3 {t3:.1f} ({gf / t3 * 1e3:.0f} TF/s)  x4 {t4:.1f}   max err / range vs float64: f32 {errs[0]:.1e}  bf16x6 {errs[1]:.1e}  f16x3 {errs[2]:.1e}", flush=True)
