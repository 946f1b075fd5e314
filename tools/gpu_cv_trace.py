"""Profiling only (needs a -DCASMVS_TRACE build selected with CASMVS_LIB_PATH): shader-clock phase timeline of
costvol_lds_kernel workgroups (thread 0 of every 32nd workgroup of batch element 0).
   python tools/gpu_cv_trace.py [level [batch [op]]]     op: var (fused variance build) | warp (un-fused homo_warp); CV_TRACE_NOISY=1: noise-like hypotheses"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from casmvsnet_pl_amd import _lib, ops
from casmvsnet_pl_amd.synthetic import make_inputs

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
op = sys.argv[3] if len(sys.argv) > 3 else "var"
H, W, V = 512, 640, 3
C, D = {2: (32, 48), 1: (16, 32), 0: (8, 8)}[level]
h, w = H >> level, W >> level
dev = torch.device("cuda:0")
L = _lib.load()
L.casmvs_cv_trace_read.restype = ctypes.c_int
L.casmvs_cv_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
_, proj, dmin, dint = make_inputs(B, V, H, W, seed=0)
g = torch.Generator(device="cpu").manual_seed(level)
feats = torch.randn(B, V, C, h, w, generator=g).to(dev)
P = proj[:, :, level].contiguous().to(dev)
k = torch.arange(D, device=dev, dtype=torch.float32).view(1, D, 1, 1)
step = dint * 2 ** level
base = 680.0 - D / 2 * step + 60.0 * torch.sin(torch.linspace(0, 6.0, w, device=dev)).view(1, 1, 1, w)
dv = (base + k * step).expand(B, D, h, w).contiguous()
if os.environ.get("CV_TRACE_NOISY") == "1":   # the noise-like hypotheses of random weights (tools/gpu_costvol_probe.py: |d depth / dx| ~ 15 units)
    coarse = 680.0 + 40.0 * torch.randn(B, 1, h // 2, w // 2, generator=g).to(dev)
    up = torch.nn.functional.interpolate(coarse, scale_factor=2, mode="bilinear", align_corners=True)
    dv = (up - D / 2 * step + k * step).contiguous()
nhwc = ops.nchw_to_nhwc(feats.view(B * V, C, h, w)).view(B, V, h, w, C)
if op == "warp":
    src, P1 = feats[:, 1].contiguous(), P[:, 0].contiguous()
    fn = lambda: ops.homo_warp(src, P1, dv, impl="lds")
else:
    fn = lambda: ops.costvol(nhwc, P, dv, 1, channels_last=True, impl="lds")
buf = (ctypes.c_ulonglong * (64 * 32))()
for _ in range(3):
    fn()
torch.cuda.synchronize()
L.casmvs_cv_trace_read(buf, 1)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); fn(); e.record(); torch.cuda.synchronize()
print(f"level {level} C={C} D={D} {h}x{w} B={B} op={op}: {s.elapsed_time(e)*1e3:.1f} us (one traced launch)")
L.casmvs_cv_trace_read(buf, 1)
t = np.array(buf, dtype=np.uint64).reshape(64, 32).astype(np.int64)
rows = [r[: int((r > 0).sum())] for r in t if r[0] > 0]
t0 = min(int(r[0]) for r in rows)
names = ["extents", "sync1", "boxes", "staged", "sync3", "setup"]
print("clock ticks (shader clock); per workgroup: start, then the duration of each phase, planes, store drain, total")
acc = []
for r in rows:
    d = np.diff(r)
    acc.append(d)
    print(f"  start {int(r[0]-t0):8d}  " + " ".join(f"{n} {int(x):6d}" for n, x in zip(names, d[:6])) +
          "  planes " + " ".join(f"{int(x):5d}" for x in d[6:-1]) + f"  drain {int(d[-1]):6d}  total {int(r[-1]-r[0]):7d}")
n = min(len(a) for a in acc)
m = np.mean([a[:n] for a in acc], axis=0)
print("mean: " + " ".join(f"{n_} {x:.0f}" for n_, x in zip(names, m[:6])) + "  planes " + " ".join(f"{x:.0f}" for x in m[6:n-1]) + f"  drain {m[n-1]:.0f}")
print(f"kernel span {max(int(r[-1]) for r in rows) - t0} ticks")
