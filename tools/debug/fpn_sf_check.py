"""Debug: split-f16 FPN tail vs a torch float32 reference on the device at growing sizes; where do wrong pixels sit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from casmvsnet_pl_amd import ops
from casmvsnet_pl_amd.mvsnet import compose_fpn_tail
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
lw, lb = torch.randn(32, 8, 1, 1, generator=g) * 0.3, torch.randn(32, generator=g)
sw, sb = torch.randn(8, 32, 3, 3, generator=g) * 0.2, torch.randn(8, generator=g)
w40, bias9 = compose_fpn_tail(lw, lb, sw, sb)
p32 = ops.conv2d_pack(ops.CONV2D_K3, w40, None, None).to(dev)
psf = ops.fpn_tail0_splitf16_pack(w40).to(dev)
b9 = bias9.to(dev)
for N, H, W in ((13, 128, 160), (6, 512, 640)):
    x, y = torch.randn(N, 8, H, W, device=dev), torch.randn(N, 32, H // 2, W // 2, device=dev)
    ref = F.conv2d(F.conv2d(x, lw.to(dev), lb.to(dev)) + F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True), sw.to(dev), sb.to(dev), padding=1)
    a, b = ops.fpn_tail0(p32, b9, x, y), ops.fpn_tail0_splitf16(psf, b9, x, y)
    ea, eb = (a - ref).abs(), (b - ref).abs()
    rng = float(ref.abs().max())
    bad = torch.nonzero(eb > 1e-4 * rng)
    tiles = N * ((H + 15) // 16) * ((W + 31) // 32)
    msg = f"N {N} {H}x{W} ({tiles} tiles): float32 kernel err {float(ea.max()) / rng:.1e}, split-f16 err {float(eb.max()) / rng:.1e}, wrong pixels {bad.shape[0]}"
    if bad.shape[0]:
        ns, ys, xs = bad[:, 0], bad[:, 2], bad[:, 3]
        t = (ns * ((H + 15) // 16) + ys // 16) * ((W + 31) // 32) + xs // 32
        msg += f"; images {sorted(set(ns.tolist()))[:8]}, rows mod 16 {sorted(set((ys % 16).tolist()))}, channels {sorted(set(bad[:, 1].tolist()))}, tiles {sorted(set(t.tolist()))[:12]} .. ({len(set(t.tolist()))} tiles)"
    print(msg, flush=True)

# second look: determinism and the exact footprint of the wrong pixels
N, H, W = 13, 128, 160
torch.manual_seed(1)
x, y = torch.randn(N, 8, H, W, device=dev), torch.randn(N, 32, H // 2, W // 2, device=dev)
ref = F.conv2d(F.conv2d(x, lw.to(dev), lb.to(dev)) + F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True), sw.to(dev), sb.to(dev), padding=1)
rng = float(ref.abs().max())
for rep in range(4):
    b = ops.fpn_tail0_splitf16(psf, b9, x, y)
    eb = (b - ref).abs()
    bad = torch.nonzero(eb > 1e-4 * rng)
    if bad.shape[0] == 0:
        print("rep", rep, "no wrong pixels")
        continue
    ns, cs, ys, xs = bad[:, 0], bad[:, 1], bad[:, 2], bad[:, 3]
    print("rep", rep, "wrong", bad.shape[0], "images", sorted(set(ns.tolist())), "y range", int(ys.min()), int(ys.max()), "x set", sorted(set(xs.tolist()))[:40],
          "max err", float(eb.max()) / rng, "channels", sorted(set(cs.tolist())))
    n0 = int(ns[0])
    sel = bad[ns == n0]
    print("   image", n0, "rows -> xs:", {int(r): sorted(set(sel[sel[:, 2] == r][:, 3].tolist())) for r in sorted(set(sel[:, 2].tolist()))})
