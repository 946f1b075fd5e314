"""Is a training step the same step every time?  (`30 consecutive runs of the SGD test with one loss vector`.)
train.py:99-127 + opt.py:40-47 in miniature - SGD lr 1e-3, momentum 0.9, weight decay 1e-5, SL1 loss over the three levels, InPlaceABN - from the same
seeded state, RUNS times; prints every run's loss bits (hex of the float32 values) and a sha256 over all parameters and running statistics afterwards,
then the number of distinct trajectories (must be 1).   python tools/gpu_sgd_repro.py [RUNS [STEPS [B H W]]]"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from casmvsnet_pl_amd import CascadeMVSNet, InPlaceABN
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict

RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 12
B, H, W = (int(a) for a in sys.argv[3:6]) if len(sys.argv) > 5 else (2, 64, 96)
dev = torch.device("cuda:0")


def run():
    model = CascadeMVSNet(norm_act=InPlaceABN)
    randomize_state_dict(model.state_dict(), seed=0)
    model = model.to(dev).train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
    imgs, proj, dmin, dint = make_inputs(B, 3, H, W, seed=3)
    imgs, proj = imgs.to(dev), proj.to(dev)
    g = torch.Generator().manual_seed(0)
    gt = {l: (560.0 + 30.0 * torch.randn(B, H >> l, W >> l, generator=g)).to(dev) for l in range(3)}
    losses = []
    for _ in range(STEPS):
        opt.zero_grad(set_to_none=True)
        res = model(imgs, proj, dmin, dint)
        loss = sum(F.smooth_l1_loss(res[f"depth_{l}"], gt[l]) * 2 ** (1 - l) for l in range(3))
        loss.backward()
        opt.step()
        losses.append(loss.detach().cpu())
    h = hashlib.sha256()
    for k, v in sorted(model.state_dict().items()):
        h.update(k.encode())
        h.update(v.detach().cpu().numpy().tobytes())
    return losses, h.hexdigest()[:16]


seen = {}
for r in range(RUNS):
    losses, digest = run()
    key = (tuple(int(l.view(torch.int32)) & 0xffffffff for l in losses), digest)
    seen[key] = seen.get(key, 0) + 1
    print(f"run {r:2d}: losses {' '.join('%.3f' % float(l) for l in losses)} | bits {' '.join('%08x' % b for b in key[0])} | state sha256 {digest}", flush=True)
print(f"{RUNS} runs of {STEPS} SGD steps (B={B}, {H}x{W}, 3 views): {len(seen)} distinct trajectory / final state pair(s)")
sys.exit(0 if len(seen) == 1 else 1)
