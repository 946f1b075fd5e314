"""Profiling only: runs one layer on a -DCASMVS_TRACE build of the library (built beforehand with
tools/build_trace_lib.sh into casmvsnet_pl_amd/libcasmvs_trace.so) and prints the raw shader-clock
stamp deltas of wave 0 of a few workgroups.   usage: gpu_trace2.py kind cin cout D H W"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from casmvsnet_pl_amd import _lib, ops
_lib.LIB_PATH = os.path.join(ROOT, "casmvsnet_pl_amd", "libcasmvs_trace.so")
L = _lib.load()
L.casmvs_trace_read.restype = ctypes.c_int
L.casmvs_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda:0")
kind, cin, cout, D, H, W = (int(x) for x in sys.argv[1:7])
x = torch.randn(1, cin, D, H, W, device=dev)
wshape = (cin, cout, 3, 3, 3) if kind == ops.CONV_T2 else (cout, cin, 3, 3, 3)
packed = ops.conv3d_pack(kind, torch.randn(wshape) * 0.05, torch.ones(cout), torch.zeros(cout)).to(dev)
oshape = {0: (D, H, W), 1: (D // 2, H // 2, W // 2), 2: (2 * D, 2 * H, 2 * W)}[kind]
skip = torch.randn(1, cout, *oshape, device=dev) if kind == ops.CONV_T2 else None
buf = (ctypes.c_ulonglong * (64 * 128))()
for it in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    y = ops.conv3d_forward(kind, packed, x, cout, skip, 0.01 if cout > 1 else 1.0)
    e.record()
    torch.cuda.synchronize()
    L.casmvs_trace_read(buf, 1)
print("kernel ms", s.elapsed_time(e))
t = np.array(buf, dtype=np.uint64).reshape(64, 128).astype(np.int64)
t0 = min(int(t[b][0]) for b in range(64) if t[b][0] > 0)
for slot in range(0, 64, 5):
    row = t[slot]; n = int((row > 0).sum())
    if n == 0:
        continue
    row = row[:n]
    print("block", slot * 16, "start", int(row[0] - t0), "end", int(row[-1] - t0), "deltas", np.diff(row).tolist())
