"""Manual measurement (not collected by pytest; ~4 min, most of it MIOpen's kernel search in the first step): the
reference's training step - restated train-mode forward (oracle/cpu_restatement.py: cascade_forward_train), SL1-style
loss, backward, SGD - executed on the MI355X by STOCK PyTorch-ROCm operators, at the reference's default training
configuration (batch 1, 3 views, 640x512).  The comparison row of DESIGN 2.6: 2224 ms per step (6.5 GiB) against 35 ms
(2.3 GiB) through casmvsnet_pl_amd.training (tools/gpu_train_step.py).   python tests/manual/stock_pytorch_train_step.py"""
import sys, time, torch
import torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cpu_restatement as R
from casmvsnet_pl_amd import ABN, CascadeMVSNet
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict
dev = torch.device("cuda:0")
m = CascadeMVSNet(norm_act=ABN)
sd0 = randomize_state_dict(m.state_dict(), seed=0)
imgs, proj, dmin, dint = make_inputs(1, 3, 512, 640, seed=0)
sd = {k: v.to(dev) for k, v in sd0.items()}
params = []
for k, v in sd.items():
    if v.dtype.is_floating_point and "running" not in k:
        v.requires_grad_(True); params.append(v)
opt = torch.optim.SGD(params, lr=1e-3, momentum=0.9)
torch.set_default_device(dev)
imgs, proj = imgs.to(dev), proj.to(dev)
ts = []
for i in range(8):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = R.cascade_forward_train(sd, imgs, proj, dmin, dint)
    loss = sum(F.smooth_l1_loss(out[f"depth_{l}"], torch.full_like(out[f"depth_{l}"], 600.0)) * 2 ** (1 - l) for l in range(3))
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    print(i, round(ts[-1], 1), "ms", flush=True)
print("stock PyTorch-ROCm train step:", sorted(ts[3:])[2], "ms; peak mem GiB", torch.cuda.max_memory_allocated() / 2**30)
