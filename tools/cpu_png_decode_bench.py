"""Host-side PNG decoding rate: libcasmvs_io.so (include/casmvs_io.h) beside PIL, the decoder the reference's dataset classes
use (datasets/dtu.py:168).  Images like tools/gpu_files_throughput.py's (640 x 512 photograph-like PNGs).  CPU only.
    python tools/cpu_png_decode_bench.py [n_images [threads ...]]"""
import io
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from casmvsnet_pl_amd import _io

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
THREADS = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8, 16]
H, W = 512, 640
g = np.random.default_rng(0)
yy, xx = np.mgrid[:H, :W]
root = tempfile.mkdtemp(prefix="casmvs_png_")
paths = []
for v in range(N):
    img = np.stack([128 + 80 * np.sin(xx / (23.0 + c) + v) * np.cos(yy / (31.0 + 2 * c)) + 12 * g.standard_normal((H, W)) for c in range(3)], -1)
    paths.append(os.path.join(root, f"rect_{v + 1:03d}_3_r5000.png"))
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(paths[-1])
print(f"{N} PNGs of {W}x{H}, {np.mean([os.path.getsize(p) for p in paths]) / 1024:.0f} KB each; host threads available: {len(os.sched_getaffinity(0))}")


def pil_decode(p):
    return np.asarray(Image.open(p).convert("RGB"))


def native_decode(p):
    with open(p, "rb") as f:
        return _io.decode_png(f.read())


assert all(np.array_equal(pil_decode(p), native_decode(p)) for p in paths[:8])
for name, fn in (("PIL", pil_decode), ("libcasmvs_io", native_decode)):
    t0 = time.perf_counter()
    for p in paths:
        fn(p)
    dt = time.perf_counter() - t0
    print(f"{name:>14}: {dt / N * 1e3:6.2f} ms per image on one thread ({N / dt:6.0f} images/s)")
for nt in THREADS:
    row = [f"{nt:3d} threads:"]
    for name, fn in (("PIL", pil_decode), ("libcasmvs_io", native_decode)):
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(fn, paths))
            t0 = time.perf_counter()
            for _ in range(3):
                list(ex.map(fn, paths))
            dt = (time.perf_counter() - t0) / 3
        row.append(f"{name} {N / dt:6.0f} images/s")
    batch = np.empty((N, H, W, 3), np.uint8)
    _io.decode_png_files(paths, W, H, 3, nt, out=batch)
    t0 = time.perf_counter()
    for _ in range(3):
        _io.decode_png_files(paths, W, H, 3, nt, out=batch)
    dt = (time.perf_counter() - t0) / 3
    row.append(f"one native call {N / dt:6.0f} images/s")
    print("  ".join(row))
