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
t0 = min(int(t[b][0]) for b in range(64) if t[b][0] > 0)
for slot in (0, 1, 15, 16, 17, 30, 40, 47):
    row = t[slot]; n = int((row > 0).sum())
    if n == 0:
        continue
    row = row[:n]
    d = np.diff(row)
    # stamps: 0 kernel start, then per chunk: begin, after barrier1, after store, mfma begin; per tile end: epilogue begin, epilogue done
    nch = 4 if len(sys.argv) < 2 else (int(sys.argv[1]) + 3) // 4
    per_tile = 4 * nch + 2
    ntiles = (n - 1) // per_tile
    print("block", slot * 16, "stamps", n, "tiles", ntiles, "start", int(row[0] - t0), "end", int(row[-1] - t0))
    for ti in range(ntiles):
        seg = d[ti * per_tile: (ti + 1) * per_tile]
        mf = [int(seg[4 * c + 3]) if 4 * c + 3 < len(seg) else -1 for c in range(nch)]
        pf = [int(seg[4 * c + 2]) for c in range(nch)]
        st = [int(seg[4 * c + 1]) for c in range(nch)]
        b1 = [int(seg[4 * c]) for c in range(nch)]
        print("   tile", ti, "wait/b1", b1, "store", st, "prefetch", pf, "mfma", mf, "epilogue", int(seg[-1]) if len(seg) == per_tile else -1)
