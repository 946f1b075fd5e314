"""Seeded synthetic inputs and weights for benchmarks and parity tests (no dataset / checkpoint is
available offline).  Restates the input contract of the reference's datasets:

* images `(B, V, 3, H, W)` ~ N(0, 1) (datasets normalise with ImageNet mean/std, dtu.py:134-137);
* projection matrices built exactly like `datasets/dtu.py:66-74,181-186`: per level l the 4x4
  `P_l = [[K_l [R|t]], [0 0 0 1]]` with the first two rows of K divided by 2**l, and
  `proj_mats[b, v, l] = (P_src,l @ inverse(P_ref,l))[:3, :4]`, level axis fine -> coarse;
* `init_depth_min = 425.0`, `depth_interval = 2.65` (opt.py:16-17, DTU range 425..935 mm).

Everything is generated on CPU with a seeded torch.Generator so that the same bytes are produced
in the build container (golden fixtures) and on the GPU box.
"""
import math

import torch

DTU_DEPTH_MIN = 425.0
DTU_DEPTH_INTERVAL = 2.65


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float64)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return torch.tensor([[1.0, 0.0, 0.0], [0.0, c, -s], [0.0, s, c]], dtype=torch.float64)


def dtu_like_cameras(V, H, W, depth_mid=680.0, baseline=60.0, levels=3):
    """Returns per-view lists of level-wise 4x4 projection matrices (float32), fine -> coarse.

    View 0 is the reference camera at the world origin; source views sit on a ring of radius
    `baseline` (alternating +x, -x, +y, -y, diagonals...) and are rotated to look at the point at
    `depth_mid` on the reference optical axis (convergent DTU-like rig, <= ~5 degrees).
    """
    f = 1446.0 * W / 640.0  # DTU: f = 361.54 px at 160x128 (1/4 of 640x512)
    K0 = torch.tensor([[f, 0.0, W / 2.0], [0.0, f, H / 2.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    dirs = [(1, 0), (-1, 0), (0, 1), (0, -1), (0.7071, 0.7071), (-0.7071, -0.7071), (0.7071, -0.7071), (-0.7071, 0.7071)]
    views = []
    for v in range(V):
        if v == 0:
            R, c = torch.eye(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)
        else:
            dx, dy = dirs[(v - 1) % len(dirs)]
            scale = baseline * (1.0 + 0.15 * ((v - 1) // len(dirs)) + 0.07 * (v - 1))
            c = torch.tensor([dx * scale, dy * scale, 0.0], dtype=torch.float64)
            # world -> camera rotation that points the optical axis at (0, 0, depth_mid): the direction
            # (-cx, -cy, depth_mid) from the camera centre to that point must map onto +z.  (Round 1 had both
            # signs flipped - a DIVERGENT rig whose source views saw only about half of the reference frustum.)
            R = _rot_x(-math.atan2(c[1].item(), depth_mid)) @ _rot_y(math.atan2(c[0].item(), depth_mid))
        t = -R @ c
        mats = []
        for l in range(levels):  # fine -> coarse
            K = K0.clone()
            K[:2] /= 2 ** l
            P = torch.eye(4, dtype=torch.float64)
            P[:3, :3] = K @ R
            P[:3, 3] = K @ t
            mats.append(P.float())
        views.append(torch.stack(mats))  # (levels, 4, 4)
    return views


def make_inputs(B=1, V=3, H=512, W=640, seed=0, geometry="dtu", levels=3, depth_scale=1.0):
    """-> imgs (B,V,3,H,W), proj_mats (B,V-1,levels,3,4), init_depth_min (float), depth_interval (float).
    depth_scale: the scene (camera baselines and the depth the rig converges on) is scaled by this factor -
    BlendedMVS rescales every scene so that depth_min = 100 (blendedmvs.py:98-104); the returned depth range is
    DTU's and must then be replaced by the caller (blendedmvs_like_interval)."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(B, V, 3, H, W, generator=g, dtype=torch.float32)
    proj = []
    for b in range(B):
        if geometry == "dtu":
            cams = dtu_like_cameras(V, H, W, depth_mid=680.0 * depth_scale, baseline=60.0 * (1.0 + 0.1 * b) * depth_scale, levels=levels)
        elif geometry == "random":
            # near-identity homographies + translations of mixed sign: exercises out-of-bounds taps
            # and the z <= 1e-7 branch (modules.py:76-79)
            cams = [torch.eye(4).repeat(levels, 1, 1)]
            for v in range(1, V):
                mats = []
                A = torch.eye(4)
                A[:3, :3] += 0.02 * torch.randn(3, 3, generator=g)
                A[:3, 3] = torch.randn(3, generator=g) * torch.tensor([3000.0, 3000.0, 400.0])
                for l in range(levels):
                    S = torch.diag(torch.tensor([1.0 / 2 ** l, 1.0 / 2 ** l, 1.0, 1.0]))
                    mats.append(S @ A @ torch.inverse(S))
                cams.append(torch.stack(mats))
        else:
            raise ValueError(geometry)
        ref_inv = torch.inverse(cams[0])                      # dtu.py:181
        proj.append(torch.stack([cams[v] @ ref_inv for v in range(1, V)])[:, :, :3])  # dtu.py:183-186
    return imgs, torch.stack(proj).contiguous(), DTU_DEPTH_MIN, DTU_DEPTH_INTERVAL


def blendedmvs_like_interval(depth_min=100.0, depth_max=100.0 * 935.0 / 425.0, n_intervals=192):
    """BlendedMVS: scenes are rescaled so depth_min -> 100 (blendedmvs.py:98-104) and
    depth_interval = (depth_max - depth_min) / 192 (blendedmvs.py:170-173)."""
    return depth_min, (depth_max - depth_min) / n_intervals


# The BASELINE.json workloads: name -> (H, W, V, num_groups, n_depths, interval_ratios, depth range kind)
CONFIGS = {
    "dtu_640x512_v3_var": (512, 640, 3, 1, (8, 32, 48), (1.0, 2.0, 4.0), "dtu"),
    "dtu_640x512_v3_gwc8": (512, 640, 3, 8, (8, 32, 48), (1.0, 2.0, 4.0), "dtu"),
    "dtu_1152x864_v5_var": (864, 1152, 5, 1, (8, 32, 48), (1.0, 2.0, 4.0), "dtu"),
    "blended_768x576_v7_var": (576, 768, 7, 1, (8, 32, 48), (1.0, 2.0, 4.0), "blended"),
}


def config_inputs(name, B=1, seed=0):
    """Synthetic inputs of a BASELINE workload: imgs, proj_mats, init_depth_min, depth_interval.  The BlendedMVS
    config uses that dataset's scene scaling: depth_min -> 100 and depth_interval = (depth_max - depth_min) / 192
    (blendedmvs.py:98-104,170-173), with the whole rig scaled accordingly."""
    H, W, V, _, _, _, kind = CONFIGS[name]
    if kind == "blended":
        dmin, dint = blendedmvs_like_interval()
        imgs, proj, _, _ = make_inputs(B, V, H, W, seed=seed, depth_scale=dmin / DTU_DEPTH_MIN)
        return imgs, proj, dmin, dint
    return make_inputs(B, V, H, W, seed=seed)


def randomize_state_dict(state_dict, seed=0, prob_gain=(0.2, 0.5, 2.0)):
    """Deterministic, well-conditioned random weights for a CascadeMVSNet state dict (in place).

    Conv weights ~ N(0, 2/fan_in); ABN gamma ~ U(0.6, 1.4), beta ~ N(0, 0.1), running_mean ~
    N(0, 0.1), running_var ~ U(0.6, 1.4); the `prob` head of level l is scaled by `prob_gain[l]` so that
    the softmax over depth is moderately peaked at every level (mean 4-bin confidence ~0.5; with
    default init max p ~ 1/D and parity would be vacuous, with a large gain it is one-hot).
    """
    g = torch.Generator().manual_seed(seed)
    for k in sorted(state_dict.keys()):
        t = state_dict[k]
        if k.endswith("num_batches_tracked"):
            continue
        if k.endswith("running_var"):
            v = torch.rand(t.shape, generator=g) * 0.8 + 0.6
        elif k.endswith("running_mean"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() == 1 and k.endswith(".weight"):      # ABN gamma
            v = torch.rand(t.shape, generator=g) * 0.8 + 0.6
        elif t.dim() == 1:                                  # biases / ABN beta
            v = torch.randn(t.shape, generator=g) * 0.1
        else:
            if ".conv7.0." in k or ".conv9.0." in k or ".conv11.0." in k:  # ConvTranspose3d (Cin, Cout, ...)
                fan_in = t.shape[0] * t[0, 0].numel() / 8.0                 # ~27/8 taps hit each output
            else:
                fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * math.sqrt(2.0 / fan_in)
            if ".prob." in k:
                level = int(k.split(".")[0].rsplit("_", 1)[1])  # "cost_reg_<l>.prob.weight"
                v = v * (prob_gain[level] if isinstance(prob_gain, (tuple, list)) else prob_gain)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return state_dict


def tensor_checksum(t):
    """Order-sensitive float64 checksum used to pin regenerated inputs/weights to the fixtures."""
    x = t.detach().double().flatten().cpu()
    w = torch.arange(1, x.numel() + 1, dtype=torch.float64) % 251 + 1
    return float((x * w).sum())
