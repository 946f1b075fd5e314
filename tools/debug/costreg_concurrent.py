"""Debug: CostRegNet.regress on two streams at once (own weights images, workspaces, inputs), eager launches, vs single-stream results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ABN
from casmvsnet_pl_amd.mvsnet import CostRegNet
from casmvsnet_pl_amd.synthetic import randomize_state_dict
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for cin, D, h, w in ((16, 32, 32, 48), (8, 8, 64, 96), (32, 48, 16, 24)):
    for ci_mode in ("splitf16", "f32"):
        nets, xs, dvs, refs = [], [], [], []
        for i in range(2):
            net = CostRegNet(cin, ABN)
            randomize_state_dict(net.state_dict(), seed=5 + i)
            net = net.to(dev).eval()
            net.ci_mode = ci_mode
            g = torch.Generator().manual_seed(i)
            x = (torch.rand(1, cin, D, h, w, generator=g) * 0.3).to(dev)
            dv = (425.0 + 2.65 * torch.arange(D).view(1, D, 1, 1) + torch.rand(1, 1, h, w, generator=g)).expand(1, D, h, w).contiguous().to(dev)
            nets.append(net); xs.append(x); dvs.append(dv)
            refs.append([t.clone() for t in net.regress(x, dv)])
        torch.cuda.synchronize()
        bad = [0, 0]
        for it in range(150):
            outs = []
            for i, st in enumerate(streams):
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    outs.append(nets[i].regress(xs[i], dvs[i]))
            torch.cuda.synchronize()
            for i in range(2):
                if not all(torch.equal(a, b) for a, b in zip(outs[i], refs[i])):
                    bad[i] += 1
        print(f"cin {cin} D {D} {h}x{w} ci_mode {ci_mode}: mismatching runs per stream {bad} of 150", flush=True)
