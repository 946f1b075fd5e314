"""Debug: does conv_ci_sf_kernel write outside its output tensor?  Output placed in the middle of a sentinel-filled arena."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from casmvsnet_pl_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
for c, shape in ((16, (1, 24, 8, 12)), (16, (1, 16, 16, 24)), (16, (1, 4, 32, 48)), (16, (2, 5, 9, 36)), (32, (1, 12, 4, 6))):
    B, D, H, W = shape
    g = torch.Generator().manual_seed(c)
    x = torch.randn(B, c, D, H, W, generator=g).to(dev)
    w = torch.randn(c, c, 3, 3, 3, generator=g) * 0.1
    p = ops.conv_ci_splitf16_pack(w, torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1).to(dev)
    n = B * c * D * H * W
    arena = torch.full((16 * 1024 * 1024,), 12345.0, device=dev)   # 64 MB
    off = 8 * 1024 * 1024
    out = arena[off:off + n]
    rc = lib.casmvs_conv_ci_splitf16_forward_f32(ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                 B, c, c, D, H, W, ctypes.c_float(0.01), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    before, after = arena[:off], arena[off + n:]
    nb, na = int((before != 12345.0).sum()), int((after != 12345.0).sum())
    inside = int((out == 12345.0).sum())
    print(c, shape, "rc", rc, "stray writes before/after the output:", nb, na, "| untouched output elements:", inside, flush=True)
    if nb:
        print("   before idx", (torch.nonzero(before != 12345.0)[:5] - off).flatten().tolist())
    if na:
        print("   after idx", torch.nonzero(after != 12345.0)[:5].flatten().tolist())
