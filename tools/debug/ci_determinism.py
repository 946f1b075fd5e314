"""Debug: is conv_ci_sf_kernel deterministic (same bits run to run, alone and beside another stream's work)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ops
dev = torch.device("cuda:0")
for c, shape in ((16, (1, 16, 16, 24)), (16, (2, 24, 64, 80)), (32, (1, 8, 8, 12)), (32, (2, 12, 32, 40))):
    g = torch.Generator().manual_seed(c)
    x = torch.randn(shape[0], c, *shape[1:], generator=g).to(dev)
    w = torch.randn(c, c, 3, 3, 3, generator=g) * 0.1
    p = ops.conv_ci_splitf16_pack(w, torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1).to(dev)
    ref = ops.conv_ci_splitf16_forward(p, x, c)
    bad = 0
    s2 = torch.cuda.Stream()
    junk = torch.randn(64, 8, 32, 64, 80, device=dev)
    for it in range(20):
        if it >= 10:
            with torch.cuda.stream(s2):
                for _ in range(4):
                    junk = junk * 1.0001 + 0.1
        y = ops.conv_ci_splitf16_forward(p, x, c)
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            d = (y - ref).abs()
            print("  mismatch it", it, "max", float(d.max()), "count", int((d > 0).sum()), "of", d.numel(), "first idx", torch.nonzero(d > 0)[:3].tolist())
    print(c, shape, "mismatching runs:", bad, flush=True)
