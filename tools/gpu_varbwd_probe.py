"""Variance cost-volume backward (casmvs_costvol_var_backward_f32) at the three cascade levels of the reference's training
configuration (batch 1, 3 views, 640x512): µs per call.   [VB_PROBE_NOISE_MM=30] python tools/gpu_varbwd_probe.py [H W [V [B]]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmvsnet_pl_amd import training as T
from casmvsnet_pl_amd.synthetic import make_inputs

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
V = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
_, proj, dmin, dint = make_inputs(B, V, H, W, seed=0)
g = torch.Generator().manual_seed(0)
for level, (C, D, ratio) in {2: (32, 48, 4.0), 1: (16, 32, 2.0), 0: (8, 8, 1.0)}.items():
    h, w = H >> level, W >> level
    P = proj[:, :, level].contiguous().to(dev)
    feats = torch.randn(B, V, C, h, w, generator=g).to(dev).requires_grad_(True)
    if level == 2:
        depth = (dmin + dint * ratio * torch.arange(D).float()).view(1, D, 1, 1).expand(B, D, h, w).contiguous()
    else:   # per-pixel hypotheses around a smooth surface, like the cascade's finer levels
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        centre = 600.0 + 40.0 * torch.sin(xx / w * 6.0) * torch.cos(yy / h * 5.0)
        depth = (centre.view(1, 1, h, w) + dint * ratio * (torch.arange(D).float() - D / 2).view(1, D, 1, 1)).expand(B, D, h, w).contiguous()
    noise = float(os.environ.get("VB_PROBE_NOISE_MM", "0"))   # per-pixel noise of the surface (an untrained model's depth maps)
    if noise > 0 and level != 2:
        depth = depth + noise * torch.randn(B, 1, h, w, generator=g)
    depth = depth.to(dev)
    vol = T.variance_volume(feats, P, depth)
    gv = torch.randn(vol.shape, generator=g).to(dev)
    times = []
    for it in range(6):
        feats.grad = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        vol.backward(gv, retain_graph=True)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3)
    times = sorted(times[1:])
    mb = (gv.numel() + 2 * feats.numel()) * 4 / 1e6
    print(f"level {level} C={C} D={D} {h}x{w} V={V} B={B}: variance backward {times[len(times) // 2]:9.1f} us  (algorithmic {mb:.0f} MB -> "
          f"{mb / times[len(times) // 2] * 1e3:.0f} GB/s)", flush=True)
