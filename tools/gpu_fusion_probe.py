"""Depth fusion of one reference view (casmvs_fuse_reference_view) at the reference's eval size
(eval.py: --img_wh 1152 864, 10 source views per reference view as in DTU's pair.txt): kernel time and fraction of the
HBM roof.  Algorithmic bytes per pixel: reference depth 4 + colour 3 + 1/16 confidence, per source view one depth (4) and
one colour (3) sample, outputs depth 4 + colour 3 x 8 + count 4 + mask 1 + world point 12.
   python tools/gpu_fusion_probe.py [H W [S]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from casmvsnet_pl_amd import fusion
from test_fusion import _scene

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (864, 1152)
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
Ps, depths, images, proba = _scene(H, W, S, seed=1)
d_src = torch.from_numpy(np.stack(depths[1:])).to(dev)
i_src = torch.from_numpy(np.stack(images[1:])).to(dev)
d_ref, i_ref, pr = torch.from_numpy(depths[0]).to(dev), torch.from_numpy(images[0]).to(dev), torch.from_numpy(proba).to(dev)
byt = H * W * (4 + 3 + 4 / 16 + S * 7 + 4 + 24 + 4 + 1 + 12)
results = {}
for paired in (False, True):
    fn = lambda: fusion.fuse_reference_view(d_ref, i_ref, pr, Ps[0], d_src, i_src, Ps[1:], conf=0.5, min_geo_consistent=3, paired_taps=paired)
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    # the call also builds S relative transforms on the host and uploads them: time the launch with events around the whole call
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(10):
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    ms = ts[len(ts) // 2]
    results[paired] = fusion.fuse_reference_view(d_ref, i_ref, pr, Ps[0], d_src, i_src, Ps[1:], conf=0.5, min_geo_consistent=3, paired_taps=paired, return_per_view=True)
    print(f"fuse_reference_view{'_paired' if paired else ''} {H}x{W}, {S} source views: {ms*1e3:.1f} us per reference view (whole call, host set-up included), "
          f"{byt/1e6:.1f} MB algorithmic -> {byt/ms/1e6:.0f} GB/s = {byt/ms/1e6/8000:.3f} of the HBM roof; "
          f"{float(out['mask_final'].float().mean())*100:.1f} % of the pixels pass")
same = {k: bool(torch.equal(results[False][k], results[True][k]) or (results[False][k].dtype.is_floating_point and torch.equal(results[False][k].isnan(), results[True][k].isnan())
                                                                        and torch.equal(results[False][k].nan_to_num(), results[True][k].nan_to_num()))) for k in results[False]}
print("paired == one-tap-per-load, bit for bit:", same, "ALL EQUAL" if all(same.values()) else "DIFFERENT")
