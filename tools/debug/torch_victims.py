"""Are kernels of OTHER code victims of the packed-float32 op_sel fault?  Float32 torch operators on stream B while stream A runs
kernels with f16 matrix instructions (this library's conv0 split-f16 kernel, or torch.mm on float16 = rocBLAS / hipBLASLt): every round's result is compared
bit for bit with the operator's solo result.   python tools/debug/torch_victims.py [rounds = 60]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ops, streams

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
n = 1 << 24
a, b, c = (torch.randn(n, generator=g).to(dev) for _ in range(3))
m2 = torch.randn(4096, 4096, generator=g).to(dev)
x4 = torch.randn(8, 32, 256, 320, generator=g).to(dev)
w4 = torch.randn(32, generator=g).to(dev)
victims = {
    "a * b + c": lambda: a * b + c,
    "addcmul(c, a, b, value=0.5)": lambda: torch.addcmul(c, a, b, value=0.5),
    "lerp(a, b, 0.3)": lambda: torch.lerp(a, b, 0.3),
    "a * 1.5 + 0.25": lambda: a * 1.5 + 0.25,
    "leaky_relu(a * b)": lambda: torch.nn.functional.leaky_relu(a * b, 0.01),
    "x * w[None, :, None, None] + w[...] (per-channel affine)": lambda: x4 * w4.view(1, -1, 1, 1) + w4.view(1, -1, 1, 1),
    "softmax(m2, dim=1)": lambda: torch.softmax(m2, dim=1),
    "layer_norm(m2)": lambda: torch.nn.functional.layer_norm(m2, (4096,)),
    "m2.sum(dim=0)": lambda: m2.sum(dim=0),
    "float32 mm (rocBLAS)": lambda: m2 @ m2,
    "batch_norm (eval) on x4": lambda: torch.nn.functional.batch_norm(x4, w4 * 0, w4.abs() + 1, w4, w4, False),
    "upsample bilinear x2": lambda: torch.nn.functional.interpolate(x4[:2], scale_factor=2, mode="bilinear", align_corners=True),
    "grid_sample": lambda: torch.nn.functional.grid_sample(x4[:2], torch.stack(torch.meshgrid(torch.linspace(-1, 1, 256, device=dev), torch.linspace(-1, 1, 320, device=dev), indexing="ij"), -1).flip(-1)[None].expand(2, -1, -1, -1) * 0.97, align_corners=True),
    "conv2d float32 3x3 (MIOpen)": lambda: torch.nn.functional.conv2d(x4[:2], torch.ones(32, 32, 3, 3, device=dev) * 0.01, padding=1),
}
h16 = torch.randn(4096, 4096, generator=g).to(dev).half()
x0 = torch.randn(2, 16, 32, 128, 160, generator=g).to(dev)
p0 = ops.conv0_splitf16_pack(torch.randn(8, 16, 3, 3, 3, generator=g) * 0.1).to(dev)
aggressors = {"none": lambda: None, "library conv0 split-f16 kernel": lambda: [ops.conv0_splitf16_forward(p0, x0) for _ in range(3)],
              "torch.mm float16 4096^3": lambda: [h16 @ h16 for _ in range(4)], "torch.mm bfloat16 4096^3": lambda: [h16.bfloat16() @ h16.bfloat16() for _ in range(3)]}
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
print(f"{rounds} rounds per pair; wrong rounds (elements that differ in the worst round)")
with streams.stream_guard(False), torch.no_grad():
    for vname, vf in victims.items():
        try:
            want = vf().clone()
        except Exception as e:   # an operator this build lacks
            print(f"{vname}: skipped ({type(e).__name__})")
            continue
        torch.cuda.synchronize()
        row = []
        for aname, af in aggressors.items():
            bad, worst = 0, 0
            for _ in range(rounds):
                with torch.cuda.stream(sa):
                    af()
                with torch.cuda.stream(sb):
                    got = vf()
                torch.cuda.synchronize()
                d = int((got.view(torch.int32) != want.view(torch.int32)).sum()) if got.dtype == torch.float32 else int((got != want).sum())
                bad += d > 0
                worst = max(worst, d)
            row.append(f"{aname}: {bad} ({worst})")
        print(f"{vname:60s} " + " | ".join(row))
