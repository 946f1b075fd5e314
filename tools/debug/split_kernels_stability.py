"""Debug: bit-stability of the split-f16 kernels at full size (two workgroups per CU share SIMDs): every launch must reproduce the first
launch's bits.  Also times the FPN tail variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ops
from casmvsnet_pl_amd.mvsnet import compose_fpn_tail
dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", "60"))


def check(name, fn):
    ref = fn()
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for _ in range(REPS):
        y = fn()
        torch.cuda.synchronize()
        if not torch.equal(y, ref):
            bad += 1
            worst = max(worst, float((y - ref).abs().max() / ref.abs().max()))
    print(f"{name}: {bad} of {REPS} launches differ from the first (worst {worst:.1e} of the range)", flush=True)


g = torch.Generator().manual_seed(0)
for l, (cin, D) in (() if os.environ.get("ONLY_FPN") else ((2, (32, 48)), (1, (16, 32)), (0, (8, 8)))):
    h, w = 512 >> l, 640 >> l
    x = torch.randn(2, cin, D, h, w, generator=g).to(dev)
    wt = torch.randn(8, cin, 3, 3, 3, generator=g) * 0.1
    psf = ops.conv0_splitf16_pack(wt, torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1).to(dev)
    psb = ops.conv0_splitbf16_pack(wt, torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1).to(dev)
    check(f"conv0 split-f16 level {l}", lambda: ops.conv0_splitf16_forward(psf, x))
    check(f"conv0 split-bf16 level {l}", lambda: ops.conv0_splitbf16_forward(psb, x))
for c, shape in (() if os.environ.get("ONLY_FPN") else ((16, (2, 16, 128, 160)), (16, (2, 4, 256, 320)), (32, (2, 8, 64, 80)))):
    x = torch.randn(shape[0], c, *shape[1:], generator=g).to(dev)
    wt = torch.randn(c, c, 3, 3, 3, generator=g) * 0.1
    p = ops.conv_ci_splitf16_pack(wt, torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1).to(dev)
    check(f"conv_ci split-f16 {c} {shape}", lambda: ops.conv_ci_splitf16_forward(p, x, c))
lw, lb = torch.randn(32, 8, 1, 1, generator=g) * 0.3, torch.randn(32, generator=g)
sw, sb = torch.randn(8, 32, 3, 3, generator=g) * 0.2, torch.randn(8, generator=g)
w40, bias9 = compose_fpn_tail(lw, lb, sw, sb)
psf = ops.fpn_tail0_splitf16_pack(w40).to(dev)
p32 = ops.conv2d_pack(ops.CONV2D_K3, w40, None, None).to(dev)
b9 = bias9.to(dev)
x, y = torch.randn(6, 8, 512, 640, device=dev), torch.randn(6, 32, 256, 320, device=dev)
check("fpn tail split-f16 N 6", lambda: ops.fpn_tail0_splitf16(psf, b9, x, y))
check("fpn tail float32 N 6", lambda: ops.fpn_tail0(p32, b9, x, y))
