"""Profiling only: builds a -DCASMVS_TRACE copy of the library, runs CostRegNet.conv0 at the 640x512
level-1 shape and prints the shader-clock phase timeline of wave 0 of a few workgroups."""
import ctypes, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from casmvsnet_pl_amd import build as B
lib = "/tmp/libcasmvs_trace.so"
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
       "-DCASMVS_TRACE", "-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC] + B._sources() + ["-o", lib]
subprocess.run(cmd, check=True)
from casmvsnet_pl_amd import _lib, ops
_lib.LIB_PATH = lib
L = _lib.load()
L.casmvs_trace_read.restype = ctypes.c_int
L.casmvs_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
cin, D, H, W = (16, 32, 256, 320) if len(sys.argv) < 2 else tuple(int(x) for x in sys.argv[1:5])
x = torch.randn(1, cin, D, H, W, device=dev)
w = torch.randn(8, cin, 3, 3, 3) * 0.05
packed = ops.conv3d_pack(ops.CONV_S1, w, torch.ones(8), torch.zeros(8)).to(dev)
buf = (ctypes.c_ulonglong * (64 * 128))()
for it in range(3):
    y = ops.conv3d_forward(ops.CONV_S1, packed, x, 8)
    torch.cuda.synchronize()
    L.casmvs_trace_read(buf, 1)
import numpy as np
t = np.array(buf, dtype=np.uint64).reshape(64, 128).astype(np.int64)
for blk in (0, 1, 7, 63):
    row = t[blk]; n = int((row > 0).sum()); row = row[:n]
    d = np.diff(row)
    print("block", blk, "stamps", n, "total cycles", int(row[-1] - row[0]))
    print("  deltas:", d[:60].tolist())
