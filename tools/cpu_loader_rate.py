"""Files -> collated uint8 batches, CPU only: pipeline.DTUReader + ParallelLoader over a DTU-format tree (the host half of
tools/gpu_files_throughput.py), with the images decoded by libcasmvs_io.so (default) and by PIL (what the reference's
dataset classes use, and what this package used before round 3's last session).  Depth maps/s = samples/s (3 views each).
    python tools/cpu_loader_rate.py [n_views_on_disk [workers ...]]"""
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from casmvsnet_pl_amd import _io
from casmvsnet_pl_amd import pipeline as P

NV = int(sys.argv[1]) if len(sys.argv) > 1 else 49
WORKERS = [int(a) for a in sys.argv[2:]] or [1, 4, 8, 16, 32]
H, W, B = 512, 640, 2
root = tempfile.mkdtemp(prefix="casmvs_dtu_")
g = np.random.default_rng(0)
os.makedirs(os.path.join(root, "Cameras"))
os.makedirs(os.path.join(root, "Rectified", "scan1"))
lines = [str(NV)]
for v in range(NV):
    lines += [str(v), "4 " + " ".join(f"{(v + d) % NV} 0.9" for d in (1, 2, 3, 4))]
open(os.path.join(root, "Cameras", "pair.txt"), "w").write("\n".join(lines) + "\n")
yy, xx = np.mgrid[:H, :W]
for v in range(NV):
    K = np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1.0]])
    E = np.eye(4)
    txt = ["extrinsic"] + [" ".join(f"{x:.6f}" for x in r) for r in E] + ["", "intrinsic"] + [" ".join(f"{x:.6f}" for x in r) for r in K] + ["", "425.0 2.5"]
    open(os.path.join(root, "Cameras", f"{v:08d}_cam.txt"), "w").write("\n".join(txt) + "\n")
    img = np.stack([128 + 80 * np.sin(xx / (23.0 + c) + v) * np.cos(yy / (31.0 + 2 * c)) + 12 * g.standard_normal((H, W)) for c in range(3)], -1)
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, "Rectified", "scan1", f"rect_{v + 1:03d}_3_r5000.png"))
reader = P.DTUReader(root, ["scan1"], n_views=3, img_wh=(W, H), n_cameras=NV)
print(f"tree: {NV} views of {W}x{H} PNG; {len(reader)} reference views, 3 views per depth map, batch {B}; host threads available {len(os.sched_getaffinity(0))}")


def rate(workers, epochs=6):
    idx = list(range(len(reader))) * epochs
    loader = P.ParallelLoader(reader, batch_size=B, num_workers=workers, indices=idx, prefetch_batches=max(4, workers // B), drop_last=True)
    t0 = time.perf_counter()
    n = sum(b["imgs_u8"].shape[0] for b in loader)
    return n / (time.perf_counter() - t0)


native = _io.decode_png
default_pool = rate(8, 4)   # also the warm-up: the allocator's first few hundred 3 MB sample arrays are fresh mappings
default_pool = rate(8)
prev = P.configure_host_threads(1)
print(f"  8 loader threads, torch's default intra-op pool ({prev} threads): {default_pool:6.0f} depth-maps/s; below: pipeline.configure_host_threads(1)")
#   # warm-up: the allocator's first few hundred 3 MB sample arrays are fresh mappings (page faults serialise the threads)
for w in WORKERS:
    _io.decode_png = native
    a = rate(w)
    _io.decode_png = lambda data, channels=3, out=None: None      # declines every file: the PIL path of pipeline._decode_file
    b = rate(w)
    _io.decode_png = native
    print(f"{w:3d} loader threads: libcasmvs_io {a:6.0f} depth-maps/s ({3 * a:6.0f} images/s)   PIL {b:6.0f} depth-maps/s   x{a / b:.2f}")
