"""The full-resolution FPN tail alone (N images of H x W): the reference's three steps as two kernels, the fused float32 kernel, the fused
split-f16 kernel - us per launch with dirtied caches.   python tools/gpu_fpn_probe.py [H W [N]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmvsnet_pl_amd import ops
from casmvsnet_pl_amd.mvsnet import compose_fpn_tail

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
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


g = torch.Generator().manual_seed(0)
lw, lb = torch.randn(32, 8, 1, 1, generator=g) * 0.3, torch.randn(32, generator=g)
sw, sb = torch.randn(8, 32, 3, 3, generator=g) * 0.2, torch.randn(8, generator=g)
w40, bias9 = compose_fpn_tail(lw, lb, sw, sb)
p32 = ops.conv2d_pack(ops.CONV2D_K3, w40, None, None).to(dev)
psf = ops.fpn_tail0_splitf16_pack(w40).to(dev)
b9 = bias9.to(dev)
for N in ([int(sys.argv[3])] if len(sys.argv) > 3 else [6, 24]):
    x, y = torch.randn(N, 8, H, W, device=dev), torch.randn(N, 32, H // 2, W // 2, device=dev)
    t32 = timed(lambda: ops.fpn_tail0(p32, b9, x, y, channels_last_copy=True))
    tsf = timed(lambda: ops.fpn_tail0_splitf16(psf, b9, x, y, channels_last_copy=True))
    a, b = ops.fpn_tail0(p32, b9, x, y), ops.fpn_tail0_splitf16(psf, b9, x, y)
    import hashlib
    print("   sha256 of the split-f16 output:", hashlib.sha256(b.cpu().numpy().tobytes()).hexdigest()[:16])
    byt = 4 * (x.numel() * 3 + y.numel())
    print(f"N {N} {H}x{W}: fused float32 {t32:.1f} us, fused split-f16 {tsf:.1f} us ({byt / tsf / 1e3:.0f} GB/s of its {byt / 1e6:.0f} MB), max diff / range {float((a - b).abs().max() / a.abs().max()):.1e}", flush=True)
