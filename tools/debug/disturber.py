"""Debug: an all-float32 CostRegNet graph on stream 1 while stream 0 replays a graph of `disturber` kernels
(DISTURB=ci: conv_ci_sf_kernel, sf: conv0_sf_kernel, f32: the float32 conv2 kernel, none).  Which workspace region of the
float32 CostRegNet goes wrong?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ABN, ops
from casmvsnet_pl_amd.mvsnet import CostRegNet
from casmvsnet_pl_amd.synthetic import randomize_state_dict
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
NAMES = ["c0", "c1", "c2", "c3", "c4", "c5", "c6", "u7", "u9", "u11"]
DIST = os.environ.get("DISTURB", "ci")
cin, D, h, w = 16, 32, 32, 48
n = D * h * w
sizes = [8 * n, 2 * n, 2 * n, n // 2, n // 2, n // 8, n // 8, n // 2, 2 * n, 8 * n]
net = CostRegNet(cin, ABN)
randomize_state_dict(net.state_dict(), seed=6)
net = net.to(dev).eval()
net.ci_mode, net.conv0_mode = "f32", "f32"
g = torch.Generator().manual_seed(1)
x = (torch.rand(1, cin, D, h, w, generator=g) * 0.3).to(dev)
dv = (425.0 + 2.65 * torch.arange(D).view(1, D, 1, 1) + torch.rand(1, 1, h, w, generator=g)).expand(1, D, h, w).contiguous().to(dev)
ref = [t.clone() for t in net.regress(x, dv)]
torch.cuda.synchronize()
refws = net._workspace.view(torch.float32).clone()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = net.regress(x, dv)
# disturber graph
xc = torch.randn(1, 16, 16, 16, 24, device=dev)
wt = torch.randn(16, 16, 3, 3, 3) * 0.1
pci = ops.conv_ci_splitf16_pack(wt).to(dev)
pf32 = ops.conv3d_pack(ops.CONV_S1, wt, None, None).to(dev)
x0 = torch.randn(1, 8, 8, 64, 96, device=dev)
w0 = torch.randn(8, 8, 3, 3, 3) * 0.1
psf = ops.conv0_splitf16_pack(w0).to(dev)


sink = torch.zeros(16, device=dev)
from casmvsnet_pl_amd import _lib
import ctypes


def disturb():
    if DIST.startswith("k"):   # k<kind>:<lds bytes>, e.g. k1:1024 = f16 MFMA loop with 1 KB of LDS
        kind, lds = DIST[1:].split(":")
        for _ in range(6):
            rc = _lib.load().casmvs_debug_disturb(int(kind), 512, 3000, int(lds), ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0
        return
    for _ in range(12):
        if DIST == "ci":
            ops.conv_ci_splitf16_forward(pci, xc, 16)
        elif DIST == "sf":
            ops.conv0_splitf16_forward(psf, x0)
        elif DIST == "f32":
            ops.conv3d_forward(ops.CONV_S1, pf32, xc, 16)


disturb()
torch.cuda.synchronize()
gd = torch.cuda.CUDAGraph()
with torch.cuda.graph(gd):
    disturb()
torch.cuda.synchronize()
first_bad, nbad = {}, 0
for it in range(800):
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(streams[0]):
        if DIST != "none":
            gd.replay()
    with torch.cuda.stream(streams[1]):
        gr.replay()
    torch.cuda.synchronize()
    ws = net._workspace.view(torch.float32)
    off, wrong = 0, []
    for name, sz in zip(NAMES, sizes):
        a, b = ws[off:off + sz], refws[off:off + sz]
        if not torch.equal(a, b):
            d = (a - b).abs()
            idx = torch.nonzero(d > 0).flatten()
            wrong.append((name, int((d > 0).sum()), float(d.max()), idx[:6].tolist()))
        off += sz
    if wrong:
        nbad += 1
        first_bad[wrong[0][0]] = first_bad.get(wrong[0][0], 0) + 1
        if nbad <= 3:
            print("   it", it, "first wrong regions:", wrong[:2])
print(f"disturber {DIST}: float32 CostRegNet replays with a wrong workspace region: {nbad} of 800; first wrong layer: {first_bad}", flush=True)
