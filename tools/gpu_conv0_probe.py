"""CostRegNet.conv0 alone at the three cascade-level shapes: the float32-MFMA kernel (conv16db_kernel<PX>) against the
split-bf16 kernel (conv0_sb_kernel, 6 and 9 partial products), each launch timed on its own with dirtied caches.
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
    a, b6 = ops.conv3d_forward(ops.CONV_S1, p32, x, 8), ops.conv0_splitbf16_forward(psb, x, terms=6)
    diff = float((a - b6).abs().max() / a.abs().max())
    print(f"level {l} (cin {cin}, D {D}, {h}x{w}): f32 MFMA {t32:.1f} ({gf / t32 * 1e3:.0f} TF/s)  split-bf16 x6 {t6:.1f} ({gf / t6 * 1e3:.0f} TF/s)  x9 {t9:.1f} ({gf / t9 * 1e3:.0f})  "
          f"max |f32 - x6| / max|f32| = {diff:.2e}", flush=True)
