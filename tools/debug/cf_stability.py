"""Debug: bit-stability of the eager forward and of two concurrent captured forwards (64x96 inputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ABN, CascadeMVSNet
from casmvsnet_pl_amd.graph import ConcurrentForwards
from casmvsnet_pl_amd.synthetic import make_inputs, randomize_state_dict
dev = torch.device("cuda:0")
m = CascadeMVSNet(norm_act=ABN)
randomize_state_dict(m.state_dict(), seed=3)
m = m.to(dev).eval()
ins = [make_inputs(1, 3, 64, 96, seed=s) for s in (1, 2)]
dmin, dint = ins[0][2], ins[0][3]
dins = [(i[0].to(dev), i[1].to(dev)) for i in ins]
want = [{k: v.clone() for k, v in m(a, b, dmin, dint).items()} for a, b in dins]
bad = 0
for it in range(40):
    for i, (a, b) in enumerate(dins):
        o = m(a, b, dmin, dint)
        torch.cuda.synchronize()
        for k in want[i]:
            if not torch.equal(o[k], want[i][k]):
                bad += 1
                d = (o[k].float() - want[i][k].float()).abs()
                print("eager mismatch it", it, "input", i, k, "count", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()))
print("eager mismatches:", bad, flush=True)
cf = ConcurrentForwards(m, dins[0][0], dins[0][1], dmin, dint, n_streams=2)
bad = 0
for it in range(60):
    outs = cf.run(dins)
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        for k in want[i]:
            if not torch.equal(o[k], want[i][k]):
                bad += 1
                d = (o[k].float() - want[i][k].float()).abs()
                idx = torch.nonzero(d > 0)
                if bad < 12:
                    print("concurrent mismatch it", it, "stream", i, k, "count", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()), "bbox", idx.min(0).values.tolist(), idx.max(0).values.tolist())
print("concurrent mismatches:", bad, flush=True)
