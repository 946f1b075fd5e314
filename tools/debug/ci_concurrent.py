"""Debug: conv_ci_sf_kernel on two streams at once (different inputs), against its own single-stream results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ops
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for c, shape in ((16, (1, 24, 8, 12)), (16, (1, 16, 16, 24)), (16, (1, 4, 32, 48)), (16, (2, 24, 64, 80)), (32, (2, 12, 32, 40))):
    g = torch.Generator().manual_seed(c)
    xs = [torch.randn(shape[0], c, *shape[1:], generator=g).to(dev) for _ in range(2)]
    w = torch.randn(c, c, 3, 3, 3, generator=g) * 0.1
    ps = [ops.conv_ci_splitf16_pack(w, torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1).to(dev) for _ in range(2)]
    refs = [ops.conv_ci_splitf16_forward(ps[i], xs[i], c) for i in range(2)]
    torch.cuda.synchronize()
    bad = 0
    for it in range(200):
        ys = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                ys.append(ops.conv_ci_splitf16_forward(ps[i], xs[i], c))
        torch.cuda.synchronize()
        for i in range(2):
            if not torch.equal(ys[i], refs[i]):
                bad += 1
                d = (ys[i] - refs[i]).abs()
                if bad <= 3:
                    idx = torch.nonzero(d > 0)
                    print("  mismatch it", it, "stream", i, "max", float(d.max()), "count", int((d > 0).sum()), "of", d.numel(), "first", idx[:2].tolist(), "last", idx[-1:].tolist())
    print(c, shape, "mismatching results:", bad, "of 400", flush=True)
