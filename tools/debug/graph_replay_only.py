"""The benched launch alone - model, batch-8 inputs, one captured hipGraph, N replays - for a rocprofv3 kernel trace of the REPLAY (tools/summarize_gaps.py reads
the repeating tail): which kernels a replay contains beyond the library's, and the gaps between them.   python tools/debug/graph_replay_only.py [replays]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ABN, CascadeMVSNet
from casmvsnet_pl_amd.graph import GraphedForward
from casmvsnet_pl_amd.synthetic import config_inputs, randomize_state_dict

dev = torch.device("cuda:0")
model = CascadeMVSNet(norm_act=ABN)
randomize_state_dict(model.state_dict(), seed=0)
model = model.to(dev).eval()
imgs, proj, dmin, dint = config_inputs("dtu_640x512_v3_var", 8, seed=0)
gf = GraphedForward(model, imgs.to(dev), proj.to(dev), dmin, dint)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for _ in range(3):
    gf()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n):
    gf()
e.record()
torch.cuda.synchronize()
print(f"{n} replays: {s.elapsed_time(e) / n:.3f} ms per replay")
