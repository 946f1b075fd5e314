"""Times casmvs_costvol_var_f32 at the three level shapes of the 640x512 config with (a) fronto-parallel
depth planes (perfect tap locality) and (b) per-pixel noisy depth (what random-init weights produce)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmvsnet_pl_amd import ops
from casmvsnet_pl_amd.synthetic import make_inputs

QUICK = bool(os.environ.get("CV_PROBE_QUICK"))
dev = torch.device("cuda:0")
_, proj, dmin, dint = make_inputs(1, 3, 512, 640, seed=0)
for l, (C, D) in {2: (32, 48), 1: (16, 32), 0: (8, 8)}.items():
    h, w = 512 >> l, 640 >> l
    feats = torch.randn(1, 3, C, h, w, device=dev)
    P = proj[:, :, l].contiguous().to(dev)
    k = torch.arange(D, device=dev, dtype=torch.float32).view(1, D, 1, 1)
    planes = (dmin + k * dint * 2 ** l * (48 * 4 / D / 2 ** l)).expand(1, D, h, w).contiguous()
    noisy = (425.0 + 500.0 * torch.rand(1, 1, h, w, device=dev) + k * dint * 2 ** l).contiguous()
    smooth = (600.0 + 100.0 * torch.sin(torch.linspace(0, 6.0, w, device=dev)).view(1, 1, 1, w) + k * dint * 2 ** l).expand(1, D, h, w).contiguous()
    nhwc = ops.nchw_to_nhwc(feats.view(3, C, h, w)).view(1, 3, h, w, C)
    for name, dv in ((("smooth", smooth),) if QUICK else (("planes", planes), ("smooth", smooth), ("noisy", noisy))):
        for lay, f in (("nchw", feats), ("nhwc", nhwc)):
            for _ in range(0 if QUICK else 3):
                ops.costvol(f, P, dv, 1, channels_last=lay == "nhwc")
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            NR = 2 if QUICK else 10
            for _ in range(NR):
                ops.costvol(f, P, dv, 1, channels_last=lay == "nhwc")
            e.record(); torch.cuda.synchronize()
            ms = s.elapsed_time(e) / NR
            byt = 4 * (3 * C * h * w + D * h * w + C * D * h * w)
            print(f"level {l} C={C} D={D} {h}x{w} depth={name:7s} {lay} {ms*1e3:8.1f} us  {byt/ms/1e6:8.1f} GB/s  frac {byt/ms/1e6/8000:.3f}")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.nchw_to_nhwc(feats.view(3, C, h, w))
    e.record(); torch.cuda.synchronize()
    print(f"level {l} nchw_to_nhwc {s.elapsed_time(e)*100:.1f} us")
