"""Files -> depth maps: a DTU-format tree on local disk (test layout: Rectified/<scan>/rect_XXX_3_r5000.png at 640x512,
Cameras/*.txt, pair.txt) read by pipeline.DTUReader through pipeline.ParallelLoader (threaded PIL decode) and
DevicePrefetcher (pinned staging, side-stream copy, uint8 -> normalised float on the device) into the engine - against
the same engine on device-resident inputs (bench.py's regime).  SURVEY 8f-4: "sustain depth-maps/sec on real data".
    python tools/gpu_files_throughput.py [n_views_on_disk [workers ...]]"""
import os
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from casmvsnet_pl_amd import ABN, CascadeMVSNet
from casmvsnet_pl_amd import pipeline as P
from casmvsnet_pl_amd.synthetic import randomize_state_dict

NV = int(sys.argv[1]) if len(sys.argv) > 1 else 49
WORKERS = [int(a) for a in sys.argv[2:]] or [1, 4, 16, 32, 64]
H, W = 512, 640
B = int(os.environ.get("FT_BATCH", "2"))                   # FT_BATCH=8 FT_GRAPH=1: the bench's launch (one hipGraph replay of 8 reference views)
GRAPH = os.environ.get("FT_GRAPH", "0") == "1"
THREADED = os.environ.get("FT_THREADED", "0") == "1"       # DevicePrefetcher(threaded=True): staging on its own thread
dev = torch.device("cuda:0")
root = tempfile.mkdtemp(prefix="casmvs_dtu_")
g = np.random.default_rng(0)
os.makedirs(os.path.join(root, "Cameras"))
os.makedirs(os.path.join(root, "Rectified", "scan1"))
lines = [str(NV)]
for v in range(NV):
    src = [(v + d) % NV for d in (1, 2, 3, 4)]
    lines += [str(v), "4 " + " ".join(f"{s} 0.9" for s in src)]
open(os.path.join(root, "Cameras", "pair.txt"), "w").write("\n".join(lines) + "\n")
yy, xx = np.mgrid[:H, :W]
for v in range(NV):
    a = 0.01 * v
    K = np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1.0]])
    E = np.eye(4)
    E[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
    E[:3, 3] = [-600 * np.sin(a), 0, 600 * (1 - np.cos(a))]
    txt = ["extrinsic"] + [" ".join(f"{x:.6f}" for x in r) for r in E] + ["", "intrinsic"] + [" ".join(f"{x:.6f}" for x in r) for r in K] + ["", "425.0 2.5"]
    open(os.path.join(root, "Cameras", f"{v:08d}_cam.txt"), "w").write("\n".join(txt) + "\n")
    # a photograph-like image (smooth gradients + texture + mild noise): PNG size / decode time comparable to DTU's rectified images
    img = np.stack([128 + 80 * np.sin(xx / (23.0 + c) + v) * np.cos(yy / (31.0 + 2 * c)) + 12 * g.standard_normal((H, W)) for c in range(3)], -1)
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, "Rectified", "scan1", f"rect_{v + 1:03d}_3_r5000.png"))
png_kb = np.mean([os.path.getsize(os.path.join(root, "Rectified", "scan1", f)) for f in os.listdir(os.path.join(root, "Rectified", "scan1"))]) / 1024
reader = P.DTUReader(root, ["scan1"], n_views=3, img_wh=(W, H), n_cameras=NV)
print(f"tree: {NV} views of {W}x{H} PNG ({png_kb:.0f} KB each) under {root}; {len(reader)} reference views, 3 views per depth map, batch {B}; host threads {os.cpu_count()}")

P.configure_host_threads(1)   # torch's spinning intra-op pool would take the decode threads' cores (pipeline.configure_host_threads)
model = CascadeMVSNet(norm_act=ABN)
randomize_state_dict(model.state_dict(), seed=0)
model = model.to(dev).eval()


def run_from_files(workers, epochs=3, processes=False, skip=3):
    idx = list(range(len(reader))) * epochs
    loader = P.ParallelLoader(reader, batch_size=B, num_workers=workers, indices=idx, prefetch_batches=max(4, workers // B), drop_last=True, processes=processes)
    n = 0
    t0 = None
    for i, b in enumerate(P.DevicePrefetcher(loader, dev, depth=3, threaded=THREADED)):
        out = forward(b["imgs"], b["proj_mats"], b["init_depth_min"].to(dev), b["depth_interval"].to(dev))
        if i == skip:                  # warm-up: first batches fill the pipeline (worker processes: also their start-up)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
        n += b["imgs"].shape[0]
    torch.cuda.synchronize()
    return (n - B) / (time.perf_counter() - t0), out


def run_decode_only(workers, processes=False):
    idx = list(range(len(reader))) * (20 if processes else 2)
    loader = P.ParallelLoader(reader, batch_size=B, num_workers=workers, indices=idx, prefetch_batches=max(4, workers // B), drop_last=True, processes=processes)
    t0 = time.perf_counter()
    n = sum(b["imgs_u8"].shape[0] for b in loader)
    return n / (time.perf_counter() - t0)


# device-resident reference: the same batch shape, inputs already in HBM, kernel by kernel (what the files path also runs)
b0 = P.collate([reader[i] for i in range(B)])
imgs = P.normalize_images_u8(b0["imgs_u8"].to(dev))
proj, dmin, dint = b0["proj_mats"].to(dev), b0["init_depth_min"].to(dev), b0["depth_interval"].to(dev)
forward = model
if GRAPH:
    from casmvsnet_pl_amd.graph import GraphedForward
    forward = GraphedForward(model, imgs, proj, dmin, dint)   # (B,1) range tensors: copied into the captured buffers per call
for _ in range(3):
    forward(imgs, proj, dmin, dint)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    forward(imgs, proj, dmin, dint)
torch.cuda.synchronize()
resident = 30 * B / (time.perf_counter() - t0)
print(f"device-resident inputs, {'one hipGraph replay per batch' if GRAPH else 'kernel by kernel'}, batch {B}, stager thread {THREADED}: {resident:.1f} depth maps/s")
for procs in (False, True):
    for wk in (WORKERS if not procs else [w for w in WORKERS if 16 <= w <= 64] or [32]):
        dec = run_decode_only(wk, procs)
        rate, _ = run_from_files(wk, epochs=60 if procs else 6, processes=procs, skip=40 if procs else 3)   # processes: ~3000 samples, timed after their start-up
        print(f"{'processes' if procs else 'threads  '} {wk:3d}: decode + collate alone {dec:7.1f} depth maps/s ({3 * dec:.0f} images/s); files -> depth maps {rate:7.1f} /s = "
              f"{rate / resident:.2f} of resident", flush=True)
