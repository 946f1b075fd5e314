"""eval.py's two steps on the MI355X engine, in memory (SURVEY 8 f-3 / f-4 wired to the hot path): depth maps for every
reference view of a scan (eval.py:209-241) - files -> DTUReader -> collate -> DevicePrefetcher -> CascadeMVSNet.forward -,
then depth filtering / fusion into a coloured point cloud (eval.py:255-350, fusion.fuse_scan) and a PLY file.  Nothing
is written between the steps: depth_0 / confidence_2 stay on the device (the reference round-trips them through PFM files).

One deliberate deviation when `img_wh` differs from the files' native size: the fusion step colours its points from the image
the depth step saw - the reader's PIL `Image.BILINEAR` resize (datasets/dtu.py:162; PIL anti-aliases on downscale) - while
eval.py:266-268 re-reads the file with cv2 and resizes it with `cv2.resize(INTER_LINEAR)` (no anti-aliasing): point colours and
the 8-bit `image_refined` of processed views can differ in the last bits there.  At the native size (the reference's DTU
evaluation: 1152 x 864 files read at 1152 x 864) the two paths are identical.
"""
import torch

from . import fusion
from .pipeline import DevicePrefetcher, collate


def reconstruct_scan(model, reader, scan, out_ply=None, conf=0.999, min_geo_consistent=5, skip=1, n_fuse_src=10, device="cuda"):
    """model: CascadeMVSNet in eval mode on `device`; reader: a pipeline.DTUReader in test layout (img_wh set).
    -> (points (N,3) float32, colors (N,3) uint8) device tensors; writes `out_ply` when given.
    n_fuse_src: source views per reference view in the fusion step (eval.py uses all of pair.txt's, meta[3])."""
    dev = torch.device(device)
    idxs = [i for i, m in enumerate(reader.metas) if m[0] == scan]
    samples = [reader[i] for i in idxs]
    views = {}
    with torch.no_grad():
        for b, s in zip(DevicePrefetcher([collate([s]) for s in samples], dev, depth=2), samples):
            res = model(b["imgs"], b["proj_mats"], b["init_depth_min"], b["depth_interval"])        # eval.py:222
            vid = s["scan_vid"][1]
            views[vid] = dict(depth=torch.nan_to_num(res["depth_0"][0]), image=s["imgs_u8"][0],      # eval.py:224-227
                              proba=torch.nan_to_num(res["confidence_2"][0]), P=reader.proj_mats[vid][0][0].numpy())
    metas = [(reader.metas[i][2], reader.metas[i][3][:n_fuse_src]) for i in idxs]
    points, colors, _ = fusion.fuse_scan(views, metas, conf=conf, min_geo_consistent=min_geo_consistent, skip=skip, device=dev)
    if out_ply is not None:
        fusion.write_ply(out_ply, points, colors)
    return points, colors
