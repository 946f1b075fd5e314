"""Debug: conv1 (float32 kernel, stride 2) -> conv2 (split-f16) -> conv3 (float32, stride 2) captured per stream, two graphs replayed
concurrently; every intermediate compared with the eager single-stream result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ops
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
MODE = os.environ.get("PAIR_MODE", "split")


def pk(kind, cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) * 0.1
    return w, torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1


for D, h, w in ((32, 32, 48), (48, 16, 24), (8, 64, 96)):
    items = []
    for i in range(2):
        g = torch.Generator().manual_seed(i)
        x = torch.randn(1, 8, D, h, w, generator=g).to(dev)
        w1 = pk(ops.CONV_S2, 8, 16, 10 + i); w2 = pk(ops.CONV_S1, 16, 16, 20 + i); w3 = pk(ops.CONV_S2, 16, 32, 30 + i)
        p1 = ops.conv3d_pack(ops.CONV_S2, *w1).to(dev)
        p2f = ops.conv3d_pack(ops.CONV_S1, *w2).to(dev)
        p2 = ops.conv_ci_splitf16_pack(*w2).to(dev)
        p3 = ops.conv3d_pack(ops.CONV_S2, *w3).to(dev)

        def chain(x=x, p1=p1, p2=p2, p2f=p2f, p3=p3):
            c1 = ops.conv3d_forward(ops.CONV_S2, p1, x, 16)
            c2 = ops.conv_ci_splitf16_forward(p2, c1, 16) if MODE == "split" else ops.conv3d_forward(ops.CONV_S1, p2f, c1, 16)
            c3 = ops.conv3d_forward(ops.CONV_S2, p3, c2, 32)
            return c1, c2, c3
        ref = [t.clone() for t in chain()]
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = chain()
        items.append((gr, out, ref))
    torch.cuda.synchronize()
    bad = [[0, 0, 0], [0, 0, 0]]
    for it in range(400):
        for i, st in enumerate(streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                items[i][0].replay()
        torch.cuda.synchronize()
        for i in range(2):
            for j in range(3):
                if not torch.equal(items[i][1][j], items[i][2][j]):
                    bad[i][j] += 1
                    if sum(map(sum, bad)) <= 4:
                        d = (items[i][1][j] - items[i][2][j]).abs()
                        idx = torch.nonzero(d > 0)
                        print("   it", it, "stream", i, "tensor c%d" % (j + 1), "count", int((d > 0).sum()), "of", d.numel(), "max", float(d.max()), "bbox", idx.min(0).values.tolist(), idx.max(0).values.tolist())
    print(f"mode {MODE} D {D} {h}x{w}: mismatching replays [stream][c1, c2, c3] = {bad} of 400", flush=True)
