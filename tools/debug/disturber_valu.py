"""Debug: are NON-matrix kernels (pure VALU / memory: image normalisation, softmax regression, depth hypotheses, the cost-volume build)
safe beside another stream's f16 MFMA work?  Each victim is captured in a hipGraph and replayed on stream 1 while stream 0 replays the
f16-MFMA disturber (casmvs_debug_disturb kind 1); every replay must reproduce the single-stream bits."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ops, _lib
from casmvsnet_pl_amd import pipeline as P
from casmvsnet_pl_amd.synthetic import make_inputs
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
sink = torch.zeros(16, device=dev)
KIND = int(os.environ.get("DISTURB_KIND", "1"))


def disturb():
    for _ in range(6):
        rc = _lib.load().casmvs_debug_disturb(KIND, 512, 3000, 1024, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0


g = torch.Generator().manual_seed(0)
B, D, h, w = 2, 32, 128, 160
cost = torch.randn(B, D, h, w, generator=g).to(dev)
dv = (425.0 + 2.65 * torch.arange(D).view(1, D, 1, 1) + torch.rand(B, 1, h, w, generator=g)).expand(B, D, h, w).contiguous().to(dev)
u8 = torch.randint(0, 256, (2, 3, 512, 640, 3), generator=g, dtype=torch.uint8).to(dev)
imgs, proj, dmin, dint = make_inputs(2, 3, 512, 640, seed=3)
feats = torch.randn(2, 3, 64, 80, 32, generator=g).to(dev)      # level-2 pixel-major features
dvl2 = (425.0 + 10.6 * torch.arange(48).view(1, 48, 1, 1) + torch.zeros(2, 1, 64, 80)).contiguous().to(dev)
pm = proj[:, :, 2].contiguous().to(dev)
victims = {
    "normalize_images_u8": lambda: P.normalize_images_u8(u8),
    "softmax_regress": lambda: torch.stack(ops.softmax_regress(cost, dv)),
    "costvol (LDS kernel, C=32)": lambda: ops.costvol(feats, pm, dvl2, 1, channels_last=True),
}
disturb()
torch.cuda.synchronize()
gd = torch.cuda.CUDAGraph()
with torch.cuda.graph(gd):
    disturb()
for name, fn in victims.items():
    ref = fn().clone()
    torch.cuda.synchronize()
    gv = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gv):
        out = fn()
    torch.cuda.synchronize()
    bad = 0
    for it in range(400):
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(streams[0]):
            gd.replay()
        with torch.cuda.stream(streams[1]):
            for _ in range(4):
                gv.replay()
        torch.cuda.synchronize()
        bad += 0 if torch.equal(out, ref) else 1
    print(f"victim {name} beside the f16-MFMA loop (kind {KIND}): {bad} of 400 rounds differ from the single-stream bits", flush=True)
