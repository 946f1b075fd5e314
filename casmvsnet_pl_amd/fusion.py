"""Depth filtering / fusion of a reference view on the MI355X: the host-side mirror of the
reference's `eval.py:113-182` (`check_geo_consistency`) and `eval.py:273-326` (masks, depth / colour averaging,
world points), one casmvs_fuse_reference_view launch per reference view.  The scan loop around it, the "refined
depth is reused by later views" dictionaries and the PFM / PNG / PLY files stay host code, as in the reference.

Inputs may be numpy arrays or torch tensors (any device); results are torch tensors on the GPU.
"""
import ctypes

import numpy as np
import torch

from . import _lib, streams


def relative_transform(P_to, P_from):
    """(P_to @ inv(P_from))[:3] in float32, as eval.py:120 / :135 compute it (np.linalg.inv of the float32 4x4)."""
    P_to, P_from = np.asarray(P_to, dtype=np.float32), np.asarray(P_from, dtype=np.float32)
    return (P_to @ np.ascontiguousarray(np.linalg.inv(P_from)))[:3].astype(np.float32)


def _dev(x, dtype, device):
    t = torch.as_tensor(np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x)
    return t.to(device=device, dtype=dtype).contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def fuse_reference_view(depth_ref, image_ref, proba_ref_quarter, P_world2ref, depth_srcs, image_srcs, P_world2srcs,
                        conf=0.999, min_geo_consistent=5, return_points=True, return_per_view=False, device="cuda", paired_taps=True):
    """eval.py:273-326 for one reference view.
    depth_ref (H,W) float32; image_ref (H,W,3) uint8 RGB; proba_ref_quarter (H/4,W/4) float32 (= confidence_2, what
    eval.py:226 saves) or None; P_world2ref (4,4); depth_srcs (S,H,W); image_srcs (S,H,W,3) uint8; P_world2srcs (S,4,4).
    -> dict: depth_refined (H,W) f32, image_refined (H,W,3) f64, mask_geo_sum (H,W) i32, mask_final (H,W) bool,
       xyz_world (H,W,3) f32 [return_points], mask_geo (S,H,W) bool / depth_ref_reproj (S,H,W) / image_src2ref
       (S,H,W,3) u8 [return_per_view: what check_geo_consistency returns for every source view].
    paired_taps=True (default) launches casmvs_fuse_reference_view_paired - one load per tap row, matrices from LDS: bit-identical
    to casmvs_fuse_reference_view on every output (tools/native/fusion_check.cpp on the MI355X: profiles/r03_fusion_paired_check.txt),
    95 -> 83 us per 1152 x 864 reference view with 10 source views; shapes it does not take (W < 2, > 64 source views) use the
    other entry."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("casmvsnet_pl_amd.fusion runs on the MI355X only; there is no CPU fallback")
    depth_ref = _dev(depth_ref, torch.float32, dev)
    H, W = depth_ref.shape
    image_ref = _dev(image_ref, torch.uint8, dev)
    S = len(depth_srcs)
    depth_src = _dev(np.stack([np.asarray(d.cpu() if isinstance(d, torch.Tensor) else d) for d in depth_srcs]) if not isinstance(depth_srcs, torch.Tensor) else depth_srcs, torch.float32, dev) if S else None
    image_src = _dev(np.stack([np.asarray(i.cpu() if isinstance(i, torch.Tensor) else i) for i in image_srcs]) if not isinstance(image_srcs, torch.Tensor) else image_srcs, torch.uint8, dev) if S else None
    if image_ref.shape != (H, W, 3) or (S and (depth_src.shape != (S, H, W) or image_src.shape != (S, H, W, 3))):
        raise ValueError("fuse_reference_view: inconsistent map shapes")
    P_ref = np.asarray(P_world2ref.cpu() if isinstance(P_world2ref, torch.Tensor) else P_world2ref, dtype=np.float32)
    P_srcs = [np.asarray(p.cpu() if isinstance(p, torch.Tensor) else p, dtype=np.float32) for p in P_world2srcs]
    m_r2s = _dev(np.stack([relative_transform(p, P_ref) for p in P_srcs]), torch.float32, dev) if S else None
    m_s2r = _dev(np.stack([relative_transform(P_ref, p) for p in P_srcs]), torch.float32, dev) if S else None
    m_r2w = _dev(np.linalg.inv(P_ref)[:3].astype(np.float32), torch.float32, dev) if return_points else None
    proba = _dev(proba_ref_quarter, torch.float32, dev) if proba_ref_quarter is not None else None
    if proba is not None and tuple(proba.shape) != (H // 4, W // 4):
        raise ValueError(f"fuse_reference_view: confidence map {tuple(proba.shape)} is not (H/4, W/4)")
    out = dict(depth_refined=torch.empty((H, W), dtype=torch.float32, device=dev),
               image_refined=torch.empty((H, W, 3), dtype=torch.float64, device=dev),
               mask_geo_sum=torch.empty((H, W), dtype=torch.int32, device=dev),
               mask_final=torch.empty((H, W), dtype=torch.uint8, device=dev))
    if return_points:
        out["xyz_world"] = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    if return_per_view:
        out["mask_geo"] = torch.empty((S, H, W), dtype=torch.uint8, device=dev)
        out["depth_ref_reproj"] = torch.empty((S, H, W), dtype=torch.float32, device=dev)
        out["image_src2ref"] = torch.empty((S, H, W, 3), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        entry = _lib.load().casmvs_fuse_reference_view_paired if (paired_taps and W >= 2 and H >= 2 and S <= 64) else _lib.load().casmvs_fuse_reference_view
        rc = entry(
            _ptr(depth_ref), _ptr(image_ref), _ptr(proba), _ptr(depth_src), _ptr(image_src), _ptr(m_r2s), _ptr(m_s2r), _ptr(m_r2w),
            _ptr(out["depth_refined"]), _ptr(out["image_refined"]), _ptr(out["mask_geo_sum"]), _ptr(out["mask_final"]),
            _ptr(out.get("xyz_world")), _ptr(out.get("mask_geo")), _ptr(out.get("depth_ref_reproj")), _ptr(out.get("image_src2ref")),
            S, H, W, float(conf), int(min_geo_consistent), streams.launch_stream(depth_ref))
    _lib.check(rc, "casmvs_fuse_reference_view")
    out["mask_final"] = out["mask_final"].bool()
    if return_per_view:
        out["mask_geo"] = out["mask_geo"].bool()
    return out


def check_geo_consistency(depth_ref, P_world2ref, depth_src, P_world2src, image_ref, image_src, img_wh=None, device="cuda"):
    """eval.py:156-182 with the reference's argument list -> (depth_ref_reproj, mask_geo, image_src2ref), masked like
    the reference's return values (image_ref / img_wh are accepted for signature parity; the shapes come from the maps)."""
    if image_ref is None:
        image_ref = np.zeros(tuple(np.asarray(depth_ref.cpu() if isinstance(depth_ref, torch.Tensor) else depth_ref).shape) + (3,), np.uint8)
    r = fuse_reference_view(depth_ref, image_ref, None, P_world2ref, [depth_src], [image_src], [P_world2src],
                            min_geo_consistent=0, return_points=False, return_per_view=True, device=device)
    return r["depth_ref_reproj"][0], r["mask_geo"][0], r["image_src2ref"][0]


def fuse_scan(views, metas, conf=0.999, min_geo_consistent=5, skip=1, device="cuda"):
    """The scan loop of eval.py:255-326 in memory: `views` maps a view id to a dict with `depth` (H,W) float32, `image`
    (H,W,3) uint8, `proba` (H/4,W/4) float32 (= confidence_2, eval.py:226) and `P` (4,4) world -> pixel projection;
    `metas` is the list of (ref_vid, src_vids) in processing order.  As in the reference, a view that was already refined
    as a reference view is used with its refined depth and 8-bit refined image when it later serves as a source (or
    reference) view; views without a prediction are skipped (the reference's FileNotFoundError branch).
    -> points (N,3) float32, colors (N,3) uint8 (device tensors), and the dict of refined depth maps."""
    depth_refined, image_refined = {}, {}
    pts, cols = [], []

    def current(vid):
        if vid in image_refined:                                  # eval.py:263-265 / :284-286
            return depth_refined[vid], image_refined[vid]
        return views[vid]["depth"], views[vid]["image"]

    for ref_vid, src_vids in metas:
        if ref_vid not in views or any(s not in views for s in src_vids):
            continue                                              # eval.py:323-326
        depth_ref, image_ref = current(ref_vid)
        srcs = [current(s) for s in src_vids]
        for s, (d, _) in zip(src_vids, srcs):
            depth_refined.setdefault(s, d)                        # eval.py:293
        r = fuse_reference_view(depth_ref, image_ref, views[ref_vid]["proba"], views[ref_vid]["P"], [d for d, _ in srcs],
                                [i for _, i in srcs], [views[s]["P"] for s in src_vids], conf=conf,
                                min_geo_consistent=min_geo_consistent, return_points=True, device=device)
        depth_refined[ref_vid] = r["depth_refined"]               # eval.py:304-305
        # save_refined_image / read_refined_image (cv2.imwrite of the float64 means, cv2.imread): an 8-bit round trip,
        # saturate_cast<uchar> = round half to even
        image_refined[ref_vid] = torch.clamp(torch.round(r["image_refined"]), 0, 255).to(torch.uint8)
        m = r["mask_final"]
        pts.append(r["xyz_world"][m][::skip])                     # eval.py:313-320 (row-major pixel order)
        cols.append(r["image_refined"][m][::skip].to(torch.uint8))    # eval.py:338: astype(uint8) of the float64 means
    dev = torch.device(device)
    points = torch.cat(pts) if pts else torch.empty((0, 3), dtype=torch.float32, device=dev)
    colors = torch.cat(cols) if cols else torch.empty((0, 3), dtype=torch.uint8, device=dev)
    return points, colors, depth_refined


def write_ply(filename, points, colors):
    """eval.py:336-350 without plyfile: binary little-endian PLY with vertex properties x y z (float) red green blue (uchar)."""
    points = np.ascontiguousarray(points.detach().cpu().numpy() if isinstance(points, torch.Tensor) else points, dtype="<f4")
    colors = np.ascontiguousarray(colors.detach().cpu().numpy() if isinstance(colors, torch.Tensor) else colors, dtype=np.uint8)
    if points.ndim != 2 or points.shape[1] != 3 or colors.shape != points.shape:
        raise ValueError("write_ply: points and colors must both be (N, 3)")
    vertex = np.empty(len(points), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    for i, k in enumerate(("x", "y", "z")):
        vertex[k] = points[:, i]
    for i, k in enumerate(("red", "green", "blue")):
        vertex[k] = colors[:, i]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
              "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(points))
    with open(filename, "wb") as f:
        f.write(header.encode("ascii"))
        vertex.tofile(f)
