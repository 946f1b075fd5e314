"""homo_warp implementations against the gather kernel at one level shape of the probe: where do they differ?   python tools/debug/warp_mismatch.py level batch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ops
from casmvsnet_pl_amd.synthetic import make_inputs
l, B = int(sys.argv[1]), int(sys.argv[2])
H, W, V = 512, 640, 3
dev = torch.device("cuda:0")
_, proj, dmin, dint = make_inputs(B, V, H, W, seed=0)
C, D = {2: (32, 48), 1: (16, 32), 0: (8, 8)}[l]
h, w = H >> l, W >> l
g = torch.Generator().manual_seed(l)
feats = torch.randn(B, V, C, h, w, generator=g).to(dev)
P = proj[:, :, l].contiguous().to(dev)
k = torch.arange(D, device=dev, dtype=torch.float32).view(1, D, 1, 1)
step = dint * 2 ** l
base_s = 680.0 - D / 2 * step + 60.0 * torch.sin(torch.linspace(0, 6.0, w, device=dev)).view(1, 1, 1, w)
smooth = (base_s + k * step).expand(B, D, h, w).contiguous()
src, P1 = feats[:, 1].contiguous(), P[:, 0].contiguous()
ref = ops.homo_warp(src, P1, smooth, impl="gather")
for impl in ("lds", "lds_copy"):
    for rep in range(2):
        out = ops.homo_warp(src, P1, smooth, impl=impl)
        bad = (out != ref) & ~(torch.isnan(out) & torch.isnan(ref))
        n = int(bad.sum())
        print(f"level {l} B={B} {impl} run {rep}: {n} of {out.numel()} differ", flush=True)
        if n:
            idx = bad.nonzero()
            print("   b:", sorted(set(idx[:, 0].tolist()))[:10], " c:", sorted(set(idx[:, 1].tolist()))[:20], " d:", sorted(set(idx[:, 2].tolist())))
            ys, xs = idx[:, 3], idx[:, 4]
            print(f"   y range {int(ys.min())}..{int(ys.max())}  x range {int(xs.min())}..{int(xs.max())}; first:", idx[:5].tolist())
            i = tuple(idx[0].tolist())
            print("   values:", float(out[i]), "want", float(ref[i]), " nan out:", int(torch.isnan(out).sum()), " max|diff|", float((out - ref).abs().max()))
            print("   x histogram (mod 32):", torch.bincount(xs % 32, minlength=32).tolist())
            print("   y histogram (mod 8):", torch.bincount(ys % 8, minlength=8).tolist())
