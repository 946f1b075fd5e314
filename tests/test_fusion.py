"""Depth fusion: the oracle's own consistency on CPU, and the HIP kernel against it on the GPU -
masks / counts / 8-bit colours bit-exact (integer work), float maps to the last bit where the oracle fixes the order."""
import numpy as np
import pytest
import torch

from oracle import fusion_restatement as F


def _scene(H=64, W=96, S=4, seed=0, noise=0.3, outliers=0.05):
    """A slanted plane seen by a ring of cameras: depth maps rendered analytically per view (+ noise, + outliers),
    random 8-bit images, 4x4 world->camera projection matrices (pixel coordinates, like dtu.py's level-0 proj_mats)."""
    g = np.random.default_rng(seed)
    f = 80.0 * W / 96.0
    K = np.array([[f, 0, W / 2, 0], [0, f, H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)
    n, d0 = np.array([0.15, -0.1, 1.0]), 600.0          # plane n . X = d0 in world coordinates

    def cam(i):
        if i == 0:
            R, c = np.eye(3), np.zeros(3)
        else:
            a = 2 * np.pi * i / S
            c = np.array([40.0 * np.cos(a), 40.0 * np.sin(a), 5.0 * i])
            ry, rx = -np.arctan2(c[0], 600.0) * 0.9, np.arctan2(c[1], 600.0) * 0.9
            Ry = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
            Rx = np.array([[1, 0, 0], [0, np.cos(rx), -np.sin(rx)], [0, np.sin(rx), np.cos(rx)]])
            R = Rx @ Ry
        E = np.eye(4)
        E[:3, :3], E[:3, 3] = R, -R @ c
        return (K @ E).astype(np.float32), R, c
    Ps, depths, images = [], [], []
    ys, xs = np.mgrid[:H, :W]
    for i in range(S + 1):
        P, R, c = cam(i)
        rays = R.T @ np.linalg.inv(K[:3, :3]) @ np.stack([xs.ravel(), ys.ravel(), np.ones(H * W)])   # world directions, z_cam = 1
        t = (d0 - n @ c) / (n @ rays)                                                                # depth along z_cam
        d = t.reshape(H, W) + noise * g.standard_normal((H, W))
        bad = g.random((H, W)) < outliers
        d[bad] *= g.uniform(0.7, 1.3, bad.sum())
        Ps.append(P)
        depths.append(d.astype(np.float32))
        images.append(g.integers(0, 256, (H, W, 3), dtype=np.uint8))
    depths[0][:2, :3] = 0.0   # zero depth: division by zero inside the masks must end as "inconsistent"
    proba = g.random((H // 4, W // 4)).astype(np.float32)
    return Ps, depths, images, proba


def test_oracle_remap_matches_exact_bilinear_on_the_32nd_grid():
    """On coordinates that are multiples of 1/32 the fixed-point remap IS exact bilinear interpolation; a zero-motion
    map returns the source; everything outside the image is 0."""
    g = np.random.default_rng(1)
    src = g.random((9, 11)).astype(np.float32)
    ys, xs = np.mgrid[:9, :11].astype(np.float32)
    assert np.array_equal(F.remap_linear_f32(src, xs, ys), src)
    mx, my = xs + np.float32(0.25), ys + np.float32(0.5)
    x0, y0 = xs.astype(int), ys.astype(int)
    pad = np.zeros((11, 13), np.float32)
    pad[:9, :11] = src
    want = (pad[y0, x0] * 0.5 * 0.75 + pad[y0, x0 + 1] * 0.5 * 0.25 + pad[y0 + 1, x0] * 0.5 * 0.75 + pad[y0 + 1, x0 + 1] * 0.5 * 0.25)
    assert np.allclose(F.remap_linear_f32(src, mx, my), want, atol=1e-6)
    assert np.all(F.remap_linear_f32(src, xs + 100, ys) == 0) and np.all(F.remap_linear_f32(src, xs * np.nan, ys) == 0)
    img = g.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    assert np.array_equal(F.remap_linear_u8(img, xs, ys), img)
    wi = F._int_weights(np.arange(32)[None].repeat(32, 0), np.arange(32)[:, None].repeat(32, 1))
    assert np.all(wi.sum(0) == 32768) and wi.min() >= 0


def test_oracle_resize_and_fusion_semantics():
    p = np.arange(12, dtype=np.float32).reshape(3, 4)
    up = F.resize_linear_x4(p)
    assert up.shape == (12, 16) and up[0, 0] == 0 and up[-1, -1] == 11
    assert np.allclose(up[0, :4], [0, 0, 0.125, 0.375])            # (dst + 0.5) / 4 - 0.5, clamped at the border
    Ps, depths, images, proba = _scene(noise=0.0, outliers=0.0)
    r = F.fuse_reference_view(depths[0], images[0], proba, Ps[0], depths[1:], images[1:], Ps[1:], conf=0.3, min_geo_consistent=3)
    inner = r["mask_geo_sum"][16:-16, 16:-16]
    assert inner.min() >= 2 and inner.mean() > 3.5                   # a clean plane is consistent wherever the views overlap
    assert np.all(r["mask_geo_sum"][:2, :3] == 0)                    # the zero-depth pixels are consistent with nothing
    # back-projected points lie on the plane n . X = d0
    X = r["xyz_world"][16:-16, 16:-16].reshape(-1, 3).astype(np.float64)
    assert np.abs(X @ np.array([0.15, -0.1, 1.0]) - 600.0).max() < 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,S,seed", [(64, 96, 4, 0), (48, 64, 1, 1), (128, 160, 6, 2), (512, 640, 4, 3)])
def test_fusion_kernel_matches_oracle(H, W, S, seed, report):
    from casmvsnet_pl_amd import fusion
    Ps, depths, images, proba = _scene(H, W, S, seed)
    want = F.fuse_reference_view(depths[0], images[0], proba, Ps[0], depths[1:], images[1:], Ps[1:], conf=0.4, min_geo_consistent=min(S, 3))
    got = fusion.fuse_reference_view(depths[0], images[0], proba, Ps[0], depths[1:], images[1:], Ps[1:], conf=0.4,
                                     min_geo_consistent=min(S, 3), return_per_view=True)
    torch.cuda.synchronize()
    g = {k: v.cpu().numpy() for k, v in got.items()}
    stats = dict(shape=[H, W, S], geo_sum_mismatch=int((g["mask_geo_sum"] != want["mask_geo_sum"]).sum()),
                 final_mismatch=int((g["mask_final"] != want["mask_final"]).sum()),
                 depth_max_abs=float(np.abs(g["depth_refined"] - want["depth_refined"]).max()),
                 image_max_abs=float(np.abs(g["image_refined"] - want["image_refined"]).max()),
                 xyz_rel=float((np.abs(g["xyz_world"] - want["xyz_world"]) / np.maximum(np.abs(want["xyz_world"]), 1.0)).max()),
                 consistent_frac=float(want["mask_final"].mean()))
    report("fusion", **stats)
    assert 0.05 < stats["consistent_frac"] < 0.95                    # the test exercises both outcomes
    assert stats["geo_sum_mismatch"] == 0 and stats["final_mismatch"] == 0          # integer work: bit-exact
    assert np.array_equal(g["depth_refined"], want["depth_refined"])                # same float32 operations in the same order
    assert np.array_equal(g["image_refined"], want["image_refined"])                # integer sums, one float64 division
    assert stats["xyz_rel"] < 1e-6
    # the per-view pieces (what check_geo_consistency returns)
    for s in range(S):
        d, m, im = F.check_geo_consistency(depths[0], Ps[0], depths[s + 1], Ps[s + 1], images[s + 1])
        assert np.array_equal(g["mask_geo"][s], m) and np.array_equal(g["depth_ref_reproj"][s], d) and np.array_equal(g["image_src2ref"][s], im)
    d1, m1, i1 = fusion.check_geo_consistency(depths[0], Ps[0], depths[1], Ps[1], images[0], images[1], (W, H))
    assert np.array_equal(m1.cpu().numpy(), g["mask_geo"][0]) and np.array_equal(d1.cpu().numpy(), g["depth_ref_reproj"][0])


@pytest.mark.gpu
def test_scan_loop_matches_oracle_and_writes_ply(tmp_path, report):
    """eval.py:255-350: every view in turn as the reference view, later views re-using the refined depth / 8-bit refined
    image of earlier ones; masked world points + truncated colours; PLY file.  Bit-exact against the numpy restatement."""
    from casmvsnet_pl_amd import fusion
    H, W, S = 48, 64, 4
    Ps, depths, images, proba = _scene(H, W, S, seed=7)
    g = np.random.default_rng(3)
    views = {v: dict(depth=depths[v], image=images[v], proba=g.random((H // 4, W // 4)).astype(np.float32), P=Ps[v]) for v in range(S + 1)}
    metas = [(r, [s for s in range(S + 1) if s != r][:3]) for r in range(S + 1)] + [(9, [0, 1, 2])]   # view 9 has no prediction: skipped
    want_p, want_c, want_d = F.fuse_scan(views, metas, conf=0.3, min_geo_consistent=2, skip=2)
    got_p, got_c, got_d = fusion.fuse_scan(views, metas, conf=0.3, min_geo_consistent=2, skip=2)
    torch.cuda.synchronize()
    report("fusion_scan", points=int(len(want_p)), views=S + 1)
    assert len(want_p) > 200 and got_p.shape == want_p.shape and got_c.shape == want_c.shape
    assert np.array_equal(got_c.cpu().numpy(), want_c)
    assert float(np.abs(got_p.cpu().numpy() - want_p).max() / np.abs(want_p).max()) < 1e-6
    for v in want_d:
        d = got_d[v].cpu().numpy() if isinstance(got_d[v], torch.Tensor) else got_d[v]
        assert np.array_equal(d, want_d[v]), v
    f = tmp_path / "scan.ply"
    fusion.write_ply(str(f), got_p, got_c)
    raw = f.read_bytes()
    head, body = raw.split(b"end_header\n", 1)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\n") and (b"element vertex %d\n" % len(want_p)) in head
    rec = np.frombuffer(body, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    assert len(rec) == len(want_p) and np.array_equal(rec["red"], want_c[:, 0]) and np.array_equal(rec["z"], got_p.cpu().numpy()[:, 2])
