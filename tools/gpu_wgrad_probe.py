"""Weight-gradient / direct input-gradient kernels on the layer shapes of one training step (batch 1, 3 views, 640x512):
µs per call (median of 5).   [CASMVS_LIB_PATH=...] python tools/gpu_wgrad_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmvsnet_pl_amd import training as T
from casmvsnet_pl_amd._lib import CONV_S1, CONV_S2, CONV_T2, CONV2D_K3, CONV2D_K5S2, CONV2D_K1

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def timed(fn):
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts[1:])[2]


# (name, kind, input shape, cout)
layers = [("L1 conv0  16->8  s1", CONV_S1, (1, 16, 32, 256, 320), 8), ("L0 conv0   8->8  s1", CONV_S1, (1, 8, 8, 512, 640), 8),
          ("L2 conv0  32->8  s1", CONV_S1, (1, 32, 48, 128, 160), 8), ("L1 prob    8->1  s1", CONV_S1, (1, 8, 32, 256, 320), 1),
          ("L1 conv2  16->16 s1", CONV_S1, (1, 16, 16, 128, 160), 16), ("L1 conv1   8->16 s2", CONV_S2, (1, 8, 32, 256, 320), 16),
          ("L1 conv11 16->8  t2", CONV_T2, (1, 16, 16, 128, 160), 8), ("2D conv0.1 8->8  k3", CONV2D_K3, (3, 8, 512, 640), 8),
          ("2D conv1.0 8->16 k5s2", CONV2D_K5S2, (3, 8, 512, 640), 16), ("2D lat0    8->32 k1", CONV2D_K1, (3, 8, 512, 640), 32)]
total = 0.0
for name, kind, xs, cout in layers:
    x = torch.randn(xs, generator=g).to(dev)
    three_d = len(xs) == 5
    if kind == CONV_T2:
        os_ = (xs[0], cout) + tuple(2 * d for d in xs[2:]); ws = (xs[1], cout, 3, 3, 3)
    elif kind in (CONV_S2, CONV2D_K5S2):
        os_ = (xs[0], cout) + tuple(d // 2 for d in xs[2:]); ws = (cout, xs[1]) + ((3, 3, 3) if three_d else (5, 5))
    else:
        k = 1 if kind == CONV2D_K1 else 3
        os_ = (xs[0], cout) + tuple(xs[2:]); ws = (cout, xs[1]) + ((3, 3, 3) if three_d else (k, k))
    gy = torch.randn(os_, generator=g).to(dev)
    w = torch.randn(ws, generator=g).to(dev)
    t_w = timed(lambda: T.conv_wgrad(kind, x, gy, ws))
    line = f"{name:24s} wgrad {t_w:8.1f} us"
    total += t_w
    if kind in (CONV2D_K5S2, CONV2D_K1):
        t_d = timed(lambda: T.conv_dgrad(kind, w, gy, xs))
        line += f"   direct dgrad {t_d:8.1f} us"
    print(line, flush=True)
print(f"sum of the wgrad calls above: {total:.1f} us")
