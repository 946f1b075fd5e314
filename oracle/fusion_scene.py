"""TEST INFRASTRUCTURE ONLY - seeded synthetic scene for the depth-fusion step (eval.py:113-182, 245-353): a slanted plane
seen by a ring of cameras, depth maps rendered analytically per view (+ noise, + outliers), random 8-bit images, 4x4
world->camera projection matrices in pixel coordinates (like dtu.py's level-0 `proj_mats[vid][0][0]`).  Shared by the
tests, by oracle/make_fusion_golden.py (which runs the reference on it) and regenerated bit-identically on the GPU box."""
import numpy as np


def scene(H=64, W=96, S=4, seed=0, noise=0.3, outliers=0.05):
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


def checksum(arrays):
    """Order-sensitive float64 checksum of a list of arrays (detects drift of the regenerated inputs)."""
    tot = 0.0
    for i, a in enumerate(arrays):
        a = np.asarray(a, dtype=np.float64).ravel()
        tot += float((a * (1.0 + (np.arange(a.size) % 97) / 97.0)).sum()) * (i + 1)
    return tot
