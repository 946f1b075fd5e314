"""Debug: with conv2 split-f16 at level 0 only, which other switch removes the level-1 instability under two concurrent graphs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ABN, CascadeMVSNet
from casmvsnet_pl_amd.graph import ConcurrentForwards
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict
dev = torch.device("cuda:0")
ins = [make_inputs(1, 3, 64, 96, seed=s) for s in (1, 2)]
dmin, dint = ins[0][2], ins[0][3]
dins = [(i[0].to(dev), i[1].to(dev)) for i in ins]


def model(levels, conv0="splitf16", fuse_regress=True, fuse_tail=True):
    m = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(m.state_dict(), seed=3)
    for l in range(3):
        getattr(m, f"cost_reg_{l}").ci_mode = "splitf16" if l in levels else "f32"
        getattr(m, f"cost_reg_{l}").conv0_mode = conv0
    m.fuse_regress = fuse_regress
    m.feature.fuse_tail = fuse_tail
    return m.to(dev).eval()


def count(run, want, n):
    bad = {}
    for it in range(n):
        outs = run()
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            for k in want[i]:
                if not torch.equal(o[k], want[i][k]):
                    bad[(i, k)] = bad.get((i, k), 0) + 1
    return bad


for name, kw in (("base", {}), ("conv0 f32", {"conv0": "f32"}), ("no fused regression", {"fuse_regress": False}), ("no fused FPN tail", {"fuse_tail": False}),
                 ("conv0 f32 + no fused regression + no fused tail", {"conv0": "f32", "fuse_regress": False, "fuse_tail": False})):
    m = model((0,), **kw)
    want = [{k: v.clone() for k, v in m(a, b, dmin, dint).items()} for a, b in dins]
    cf = ConcurrentForwards(m, dins[0][0], dins[0][1], dmin, dint, n_streams=2)
    print(name, "| concurrent mismatches in 80 replays:", count(lambda: cf.run(dins), want, 80), flush=True)
