"""Concurrent forwards WITH the split-f16 layers on several HIP streams (the combination casmvsnet_pl_amd/streams.py used to forbid): every replay of every
stream must reproduce the single-stream forward bit for bit, and the throughput is printed beside the single-stream graph of the same total batch.
   python tools/gpu_mixed_streams.py [streams = 2 [batch per stream = 1 [rounds = 200]]]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmvsnet_pl_amd import CascadeMVSNet, streams
from casmvsnet_pl_amd.graph import ConcurrentForwards, GraphedForward
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
model = CascadeMVSNet()
randomize_state_dict(model.state_dict(), seed=3)
model = model.to(dev).eval()
ins = [make_inputs(B, 3, 512, 640, seed=10 + s) for s in range(S)]
dmin, dint = ins[0][2], ins[0][3]
with torch.no_grad():
    want = [{k: v.clone() for k, v in model(i[0].to(dev), i[1].to(dev), dmin, dint).items()} for i in ins]   # one stream, kernel by kernel
    modes = {l: (getattr(model, f"cost_reg_{l}").conv0_mode, getattr(model, f"cost_reg_{l}").ci_mode) for l in range(3)}
    print(f"{S} streams x batch {B}, layer modes {modes}, FeatureNet tail {model.feature.tail_mode}")
    cf = ConcurrentForwards(model, ins[0][0].to(dev), ins[0][1].to(dev), dmin, dint, n_streams=S, mixed_matrix_types=True)
    batches = [(i[0].to(dev), i[1].to(dev)) for i in ins]
    bad = 0
    for r in range(ROUNDS):
        outs = cf.run(batches)
        torch.cuda.synchronize()
        for o, w in zip(outs, want):
            for k in w:
                if not torch.equal(o[k], w[k]):
                    bad += 1
                    if bad <= 5:
                        d = (o[k] - w[k]).abs()
                        print(f"  round {r} {k}: {int((d > 0).sum())} values differ, max {float(d.max()):.4g}")
    print(f"{bad} of {ROUNDS * S * len(want[0])} output tensors differ from the single-stream forward ({ROUNDS} rounds)")
    for _ in range(5):
        cf.run(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 50
    for _ in range(K):
        cf.run(None)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{S} streams x batch {B}: {1e3 * el / K:.3f} ms per round = {S * B * K / el:.1f} depth maps/s")
    big = make_inputs(S * B, 3, 512, 640, seed=10)
    streams.reset()
    gf = GraphedForward(model, big[0].to(dev), big[1].to(dev), dmin, dint)
    for _ in range(5):
        gf()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        gf()
    torch.cuda.synchronize()
    el1 = time.perf_counter() - t0
    print(f"1 stream x batch {S * B}: {1e3 * el1 / K:.3f} ms per step = {S * B * K / el1:.1f} depth maps/s")
