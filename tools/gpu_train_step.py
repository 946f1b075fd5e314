"""One training step of the reference's default configuration (train.py / opt.py: batch 1, 3 views, 640x512,
n_depths [8,32,48], InPlaceABN, SL1 loss over the three levels, SGD lr 1e-3 momentum 0.9) through the HIP training path:
ms per forward / backward / optimizer step and peak memory.   python tools/gpu_train_step.py [H W [B [steps]]]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from casmvsnet_pl_amd import CascadeMVSNet, InPlaceABN
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 640)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
model = CascadeMVSNet(norm_act=InPlaceABN)
randomize_state_dict(model.state_dict(), seed=0)
model = model.to(dev).train()
opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-5)
imgs, proj, dmin, dint = make_inputs(B, 3, H, W, seed=0)
imgs, proj = imgs.to(dev), proj.to(dev)
g = torch.Generator().manual_seed(0)
gt = {l: (600.0 + 40.0 * torch.randn(B, H >> l, W >> l, generator=g)).to(dev) for l in range(3)}
mask = {l: (torch.rand(B, H >> l, W >> l, generator=g) > 0.2).to(dev) for l in range(3)}


def loss_fn(res):   # losses.py: SL1Loss
    return sum(F.smooth_l1_loss(res[f"depth_{l}"][mask[l]], gt[l][mask[l]]) * 2 ** (1 - l) for l in range(3))


ev = lambda: torch.cuda.Event(enable_timing=True)
rows = []
torch.cuda.reset_peak_memory_stats()
for it in range(steps + 2):
    e0, e1, e2, e3 = ev(), ev(), ev(), ev()
    t0 = time.perf_counter()
    e0.record()
    opt.zero_grad(set_to_none=True)
    res = model(imgs, proj, dmin, dint)
    loss = loss_fn(res)
    e1.record()
    loss.backward()
    e2.record()
    opt.step()
    e3.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    if it >= 2:
        rows.append((e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), wall, float(loss)))
rows.sort(key=lambda r: r[3])
f, b, o, wall, loss = rows[len(rows) // 2]
print(f"train step {H}x{W} B={B} V=3 (losses.py SL1Loss: boolean-mask indexing, a host sync per level): forward {f:.1f} ms  backward {b:.1f} ms  optimizer {o:.1f} ms  wall {wall:.1f} ms  "
      f"({B / wall * 1e3:.2f} samples/s)  loss {loss:.3f}  peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
del res, loss   # the first phase's autograd graph (its AccumulateGrad nodes live on the default stream) must be gone before the capture
opt.zero_grad(set_to_none=True)
# the same step with a sync-free masked loss, kernel by kernel and captured ONCE into a hipGraph (forward + loss + backward +
# SGD update: every op of the training path is capture-safe - no host read-back, no allocation outside torch's pool)
maskf = {l: mask[l].float() for l in range(3)}


def step():
    opt.zero_grad(set_to_none=False)
    res = model(imgs, proj, dmin, dint)
    loss = sum((F.smooth_l1_loss(res[f"depth_{l}"], gt[l], reduction="none") * maskf[l]).sum() / maskf[l].sum() * 2 ** (1 - l) for l in range(3))
    loss.backward()
    opt.step()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 5 * 1e3
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
for _ in range(2):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    graph.replay()
torch.cuda.synchronize()
print(f"same step, sync-free masked loss: {eager:.1f} ms kernel by kernel, {(time.perf_counter() - t0) / 10 * 1e3:.1f} ms as one hipGraph replay", flush=True)
model.eval()
with torch.no_grad():
    for _ in range(2):
        model(imgs, proj, dmin, dint)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model(imgs, proj, dmin, dint)
    torch.cuda.synchronize()
print(f"eval forward of the same batch: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
