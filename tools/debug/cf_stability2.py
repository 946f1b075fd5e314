"""Debug: which cascade level's conv2 (split-f16) is unstable under two concurrent captured forwards; single-graph stability."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ABN, CascadeMVSNet
from casmvsnet_pl_amd.graph import ConcurrentForwards, GraphedForward
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict
dev = torch.device("cuda:0")
ins = [make_inputs(1, 3, 64, 96, seed=s) for s in (1, 2)]
dmin, dint = ins[0][2], ins[0][3]
dins = [(i[0].to(dev), i[1].to(dev)) for i in ins]


def model(levels):
    m = CascadeMVSNet(norm_act=ABN)
    randomize_state_dict(m.state_dict(), seed=3)
    for l in range(3):
        getattr(m, f"cost_reg_{l}").ci_mode = "splitf16" if l in levels else "f32"
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


for levels in ((0, 1, 2), (1,), (0,), (2,), ()):
    m = model(levels)
    want = [{k: v.clone() for k, v in m(a, b, dmin, dint).items()} for a, b in dins]
    gf = GraphedForward(m, dins[1][0], dins[1][1], dmin, dint)
    b1 = count(lambda: [gf(*dins[1])], [want[1]], 60)
    cf = ConcurrentForwards(m, dins[0][0], dins[0][1], dmin, dint, n_streams=2)
    b2 = count(lambda: cf.run(dins), want, 60)
    b3 = count(lambda: cf.run([dins[1], dins[0]]), [want[1], want[0]], 60)
    print("split levels", levels, "| single graph (input 2):", b1, "| concurrent:", b2, "| concurrent, inputs swapped:", b3, flush=True)
