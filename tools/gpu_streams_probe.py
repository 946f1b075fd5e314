"""Throughput of N concurrent single-view forwards (one hipGraph each, one stream each) vs one batched forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from casmvsnet_pl_amd import ABN, CascadeMVSNet
from casmvsnet_pl_amd.graph import GraphedForward
from casmvsnet_pl_amd.synthetic import config_inputs, randomize_state_dict

dev = torch.device("cuda:0")
cfg = "dtu_640x512_v3_var"


def build(B, seed):
    m = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(m.state_dict(), seed=0)
    m = m.to(dev).eval()
    imgs, proj, dmin, dint = config_inputs(cfg, B, seed=seed)
    return m, imgs.to(dev), proj.to(dev), dmin, dint


for B, NS in ((1, 1), (2, 1), (1, 2), (1, 3), (2, 2), (4, 1)):
    items = []
    for s in range(NS):
        m, imgs, proj, dmin, dint = build(B, s)
        for _ in range(3):
            m(imgs, proj, dmin, dint)
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            gf = GraphedForward(m, imgs, proj, dmin, dint)
        items.append((gf, st))
    torch.cuda.synchronize()
    K = 30
    t0 = time.perf_counter()
    for _ in range(K):
        for gf, st in items:
            with torch.cuda.stream(st):
                gf.graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batch {B} x {NS} stream(s): {1e3 * dt / K:.3f} ms per round, {B * NS * K / dt:.1f} depth-maps/s", flush=True)
    del items
