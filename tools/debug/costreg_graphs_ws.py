"""Debug: as costreg_graphs.py, but after every concurrent replay each layer's region of the workspace is compared with the eager run's:
which layer output goes wrong first?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ABN
from casmvsnet_pl_amd.mvsnet import CostRegNet
from casmvsnet_pl_amd.synthetic import randomize_state_dict
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
NAMES = ["c0", "c1", "c2", "c3", "c4", "c5", "c6", "u7", "u9", "u11"]
for cin, D, h, w in ((16, 32, 32, 48), (32, 48, 16, 24)):
    n = D * h * w
    sizes = [8 * n, 2 * n, 2 * n, n // 2, n // 2, n // 8, n // 8, n // 2, 2 * n, 8 * n]
    nets, xs, dvs, refs, graphs, outs, refws = [], [], [], [], [], [], []
    for i in range(2):
        net = CostRegNet(cin, ABN)
        randomize_state_dict(net.state_dict(), seed=5 + i)
        net = net.to(dev).eval()
        net.ci_mode, net.conv0_mode = "splitf16", "f32"
        g = torch.Generator().manual_seed(i)
        x = (torch.rand(1, cin, D, h, w, generator=g) * 0.3).to(dev)
        dv = (425.0 + 2.65 * torch.arange(D).view(1, D, 1, 1) + torch.rand(1, 1, h, w, generator=g)).expand(1, D, h, w).contiguous().to(dev)
        nets.append(net); xs.append(x); dvs.append(dv)
        refs.append([t.clone() for t in net.regress(x, dv)])
        torch.cuda.synchronize()
        refws.append(net._workspace.view(torch.float32).clone())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            o = net.regress(x, dv)
        graphs.append(gr); outs.append(o)
    torch.cuda.synchronize()
    first_bad = {}
    nbad = 0
    for it in range(600):
        for i, st in enumerate(streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                graphs[i].replay()
        torch.cuda.synchronize()
        for i in range(2):
            ws = nets[i]._workspace.view(torch.float32)
            off = 0
            wrong = []
            for name, sz in zip(NAMES, sizes):
                a, b = ws[off:off + sz], refws[i][off:off + sz]
                if not torch.equal(a, b):
                    d = (a - b).abs()
                    wrong.append((name, int((d > 0).sum()), sz, float(d.max())))
                off += sz
            if wrong:
                nbad += 1
                key = (i, wrong[0][0])
                first_bad[key] = first_bad.get(key, 0) + 1
                if nbad <= 4:
                    print("   it", it, "stream", i, "wrong regions (name, count, size, max):", wrong)
    print(f"cin {cin} D {D} {h}x{w}: replays with a wrong workspace region: {nbad}; first wrong region by (stream, layer): {first_bad}", flush=True)
