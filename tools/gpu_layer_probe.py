"""Single CostRegNet layers at the three cascade-level shapes of a config, each launch timed on its own with caches
dirtied by a 512 MB fill_ in between (the state inside the forward): the transposed layers (conv7 / conv9 / conv11 with
their skip inputs), the `prob` head at several depth-chunk sizes, the head fused with the regression, the regression alone.
   python tools/gpu_layer_probe.py [H W [batch]]      CASMVS_LIB_PATH=<variant .so> for A/B builds
   env LAYER_PROBE_ITEMS=deconv,prob,bottom (default all)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmvsnet_pl_amd import ops

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ITEMS = os.environ.get("LAYER_PROBE_ITEMS", "deconv,prob,bottom").split(",")
REPS = int(os.environ.get("LAYER_PROBE_REPS", "8"))
dev = torch.device("cuda:0")
dirty = torch.empty(512 * 262144, device=dev)


def timed(fn):
    for _ in range(2):
        fn()
    tot = 0.0
    for _ in range(REPS):
        dirty.fill_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / REPS * 1e3   # us


def pack(kind, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn((cin, cout, 3, 3, 3) if kind == ops.CONV_T2 else (cout, cin, 3, 3, 3), generator=g) * 0.1
    return ops.conv3d_pack(kind, w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1).to(dev)


print(f"layer probe {H}x{W} batch {B}, library {os.environ.get('CASMVS_LIB_PATH', 'production')}, us per launch (dirtied caches)")
for l, D in ((2, 48), (1, 32), (0, 8)):
    h, w = H >> l, W >> l
    n = B * D * h * w
    row = [f"level {l} (D {D}, {h}x{w}):"]
    if "deconv" in ITEMS:
        for name, cin, cout, f in (("conv7", 64, 32, 8), ("conv9", 32, 16, 4), ("conv11", 16, 8, 2)):
            x = torch.randn(B, cin, D // f, h // f, w // f, device=dev)
            skip = torch.randn(B, cout, 2 * D // f, 2 * h // f, 2 * w // f, device=dev)
            pk = pack(ops.CONV_T2, cin, cout, 1)
            us = timed(lambda: ops.conv3d_forward(ops.CONV_T2, pk, x, cout, skip=skip))
            gf = 2 * 27 * cin * cout * x[:, 0].numel() / 1e9
            byt = 4 * (x.numel() + 2 * skip.numel())
            row.append(f"{name} {us:.1f} ({gf / us * 1e3:.0f} TF/s, {byt / us / 1e3:.0f} GB/s)")
    if "bottom" in ITEMS:
        for name, kind, cin, cout, f in (("conv1", ops.CONV_S2, 8, 16, 1), ("conv2", ops.CONV_S1, 16, 16, 2), ("conv3", ops.CONV_S2, 16, 32, 2),
                                         ("conv4", ops.CONV_S1, 32, 32, 4), ("conv5", ops.CONV_S2, 32, 64, 4), ("conv6", ops.CONV_S1, 64, 64, 8)):
            x = torch.randn(B, cin, D // f, h // f, w // f, device=dev)
            pk = pack(kind, cin, cout, 2)
            us = timed(lambda: ops.conv3d_forward(kind, pk, x, cout))
            nout = x[:, 0].numel() / (8 if kind == ops.CONV_S2 else 1)
            row.append(f"{name} {us:.1f} ({2 * 27 * cin * cout * nout / 1e9 / us * 1e3:.0f} TF/s)")
            if kind == ops.CONV_S1 and cin == cout and cin in (16, 32):   # the same layer on the f16 matrix cores (conv_ci_splitf16.hip)
                g = torch.Generator().manual_seed(2)
                wt = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.1
                sc_, sh_ = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
                psf = ops.conv_ci_splitf16_pack(wt, sc_, sh_).to(dev)
                us2 = timed(lambda: ops.conv_ci_splitf16_forward(psf, x, cout))
                a, b = ops.conv3d_forward(kind, pk, x, cout), ops.conv_ci_splitf16_forward(psf, x, cout)
                row.append(f"[split-f16 {us2:.1f} ({2 * 27 * cin * cout * nout / 1e9 / us2 * 1e3:.0f} TF/s), max diff / range {float((a - b).abs().max() / a.abs().max()):.1e}]")
    print("  ".join(row), flush=True)
    if "prob" in ITEMS:
        x = torch.randn(B, 8, D, h, w, device=dev)
        g = torch.Generator().manual_seed(3)
        wt = torch.randn(1, 8, 3, 3, 3, generator=g) * 0.3
        pk = ops.conv3d_pack(ops.CONV_S1, wt, None, torch.randn(1, generator=g)).to(dev)
        dv = (425.0 + 2.65 * torch.arange(D, device=dev).view(1, D, 1, 1) + torch.rand(B, 1, h, w, device=dev)).expand(B, D, h, w).contiguous()
        byt = 4 * 9 * n
        row = [f"    prob head (HBM {byt / 8e3 / 1e3:.1f} us at 8 TB/s):"]
        us = timed(lambda: ops.conv3d_forward(ops.CONV_S1, pk, x, 1, slope=1.0))
        row.append(f"layer entry {us:.1f}")
        for zc in ([int(z) for z in os.environ['LAYER_PROBE_ZCHUNKS'].split(',')] if os.environ.get('LAYER_PROBE_ZCHUNKS') else sorted({0, 4, 8, 16, D})):
            if zc > D:
                continue
            us = timed(lambda: ops.prob_regress(pk, x, zchunk=zc))
            row.append(f"zchunk {zc}: {us:.1f}")
        cost = ops.prob_regress(pk, x, zchunk=0)
        us_sm = timed(lambda: ops.softmax_regress(cost, dv))
        row.append(f"| softmax alone {us_sm:.1f}")
        for zc in (0, D):
            us = timed(lambda: ops.prob_regress(pk, x, dv, zchunk=zc))
            row.append(f"head+regression zchunk {zc}: {us:.1f}")
        print("  ".join(row), flush=True)
