"""ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product package `casmvsnet_pl_amd`
(only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

CPU fp32 restatement of the reference's hot path `models/mvsnet.py::CascadeMVSNet.forward` in
eval mode, written functionally over a state dict (no nn.Module from the reference is used, so it
travels to the GPU box where /root/reference does not exist).  The arithmetic the reference
delegates to third-party code (torch: conv3d, conv_transpose3d, grid_sample, batch_norm, softmax,
interpolate, avg_pool3d; inplace_abn.ABN = batch_norm + leaky_relu(0.01); kornia 0.2.0
create_meshgrid) is restated with the same torch CPU ops, in the same order, so the restatement
is bit-comparable with the real reference wherever torch itself is deterministic.

Pinning: the reference holds no tests or golden vectors for this path ("parity
unpinned by the reference").  This oracle is therefore pinned to OUTPUTS OF THE REFERENCE ITSELF,
executed in the build container by oracle/make_golden.py (unmodified /root/reference/models/*.py
+ the two import shims under oracle/shims) and committed under tests/golden/; tests/test_oracle.py
checks the restatement against those fixtures and, when /root/reference is present, against the
live reference.

Every function cites the reference lines (relative to /root/reference) it follows.
"""
import torch
import torch.nn.functional as F

ABN_EPS = 1e-5      # inplace_abn.ABN default eps
ABN_SLOPE = 0.01    # inplace_abn.ABN default activation_param (leaky_relu)


# ---- models/modules.py --------------------------------------------------------------------------

def get_depth_values(current_depth, n_depths, depth_interval):
    """modules.py:34-49.  current_depth (B,1,H,W); depth_interval float or (B,1) -> (B,D,H,W)."""
    if not isinstance(depth_interval, float):
        depth_interval = depth_interval.reshape(-1, 1, 1, 1)                       # :43
    depth_min = torch.clamp_min(current_depth - n_depths / 2 * depth_interval, 1e-7)  # :44
    steps = torch.arange(0, n_depths, dtype=current_depth.dtype).reshape(1, -1, 1, 1)
    return depth_min + depth_interval * steps                                       # :45-48


def homo_warp(src_feat, proj_mat, depth_values):
    """modules.py:52-92.  src_feat (B,C,H,W), proj_mat (B,3,4), depth_values (B,D,H,W) -> (B,C,D,H,W)."""
    B, C, H, W = src_feat.shape
    D = depth_values.shape[1]
    R = proj_mat[:, :, :3]                                                          # :62
    T = proj_mat[:, :, 3:]                                                          # :63
    # kornia.utils.create_meshgrid(H, W, normalized_coordinates=False): [...,0]=x, [...,1]=y  :66-67
    xs = torch.linspace(0, W - 1, W, dtype=src_feat.dtype)   # float32 in the reference; float64 when the whole oracle
    ys = torch.linspace(0, H - 1, H, dtype=src_feat.dtype)   # is run in double (the "truth" of the gradient tests)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    ref_grid = torch.stack([gx, gy], 0).reshape(1, 2, H * W).expand(B, -1, -1)     # :68-69
    ref_grid = torch.cat((ref_grid, torch.ones_like(ref_grid[:, :1])), 1)           # :70
    ref_grid_d = ref_grid.repeat(1, 1, D)                                           # :71 'b c x -> b c (d x)'
    src_grid_d = R @ ref_grid_d + T / depth_values.reshape(B, 1, D * H * W)         # :72
    negative_depth_mask = src_grid_d[:, 2:] <= 1e-7                                 # :76
    src_grid_d[:, 0:1][negative_depth_mask] = W                                     # :77
    src_grid_d[:, 1:2][negative_depth_mask] = H                                     # :78
    src_grid_d[:, 2:3][negative_depth_mask] = 1                                     # :79
    src_grid = src_grid_d[:, :2] / src_grid_d[:, 2:]                                # :81
    src_grid[:, 0] = src_grid[:, 0] / ((W - 1) / 2) - 1                             # :83
    src_grid[:, 1] = src_grid[:, 1] / ((H - 1) / 2) - 1                             # :84
    src_grid = src_grid.reshape(B, 2, D, H * W).permute(0, 2, 3, 1)                 # :85
    warped = F.grid_sample(src_feat, src_grid, mode="bilinear", padding_mode="zeros",
                           align_corners=True)                                      # :87-89
    return warped.reshape(B, C, D, H, W)                                            # :90


def depth_regression(p, depth_values):
    """modules.py:95-104."""
    if depth_values.dim() == 1:
        depth_values = depth_values.reshape(1, -1, 1, 1)
    return (p * depth_values).sum(1).to(depth_values.dtype)


# ---- building blocks over a state dict ------------------------------------------------------------

def _abn(x, sd, prefix, training=False):
    """inplace_abn.ABN: batch_norm(eps 1e-5, momentum 0.1) + leaky_relu(0.01); eval mode = running statistics, train mode
    (train.py) = batch statistics + in-place update of the running ones."""
    x = F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                     sd[prefix + ".bias"], training, 0.1, ABN_EPS)
    return F.leaky_relu(x, negative_slope=ABN_SLOPE)


def _cbr2d(x, sd, prefix, stride, pad, training=False):
    """modules.py:8-18 ConvBnReLU."""
    return _abn(F.conv2d(x, sd[prefix + ".conv.weight"], None, stride, pad), sd, prefix + ".bn", training)


def _cbr3d(x, sd, prefix, stride, training=False):
    """modules.py:21-31 ConvBnReLU3D (k3, p1, no conv bias)."""
    return _abn(F.conv3d(x, sd[prefix + ".conv.weight"], None, stride, 1), sd, prefix + ".bn", training)


def _up3d(x, sd, prefix, training=False):
    """mvsnet.py:74-87: ConvTranspose3d(k3, p1, output_padding 1, s2, no bias) + ABN."""
    y = F.conv_transpose3d(x, sd[prefix + ".0.weight"], None, stride=2, padding=1, output_padding=1)
    return _abn(y, sd, prefix + ".1", training)


def feature_net(x, sd, prefix="feature", training=False):
    """mvsnet.py:7-57 FeatureNet.forward.  x (N,3,H,W) -> dict level_0/1/2."""
    p, tr = prefix, training
    c0 = _cbr2d(_cbr2d(x, sd, p + ".conv0.0", 1, 1, tr), sd, p + ".conv0.1", 1, 1, tr)    # :14-16
    c1 = _cbr2d(c0, sd, p + ".conv1.0", 2, 2, tr)                                          # :18-21
    c1 = _cbr2d(_cbr2d(c1, sd, p + ".conv1.1", 1, 1, tr), sd, p + ".conv1.2", 1, 1, tr)
    c2 = _cbr2d(c1, sd, p + ".conv2.0", 2, 2, tr)                                          # :23-26
    c2 = _cbr2d(_cbr2d(c2, sd, p + ".conv2.1", 1, 1, tr), sd, p + ".conv2.2", 1, 1, tr)

    def up_add(a, b):                                                                      # :36-38
        return F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True) + b

    feat2 = F.conv2d(c2, sd[p + ".toplayer.weight"], sd[p + ".toplayer.bias"])             # :45
    feat1 = up_add(feat2, F.conv2d(c1, sd[p + ".lat1.weight"], sd[p + ".lat1.bias"]))      # :46
    feat0 = up_add(feat1, F.conv2d(c0, sd[p + ".lat0.weight"], sd[p + ".lat0.bias"]))      # :47
    feat1 = F.conv2d(feat1, sd[p + ".smooth1.weight"], sd[p + ".smooth1.bias"], padding=1)  # :50
    feat0 = F.conv2d(feat0, sd[p + ".smooth0.weight"], sd[p + ".smooth0.bias"], padding=1)  # :51
    return {"level_0": feat0, "level_1": feat1, "level_2": feat2}


def cost_reg_net(x, sd, prefix, return_intermediates=False, training=False):
    """mvsnet.py:91-104 CostRegNet.forward.  x (B,Cin,D,h,w) -> (B,1,D,h,w)."""
    p, tr = prefix, training
    conv0 = _cbr3d(x, sd, p + ".conv0", 1, tr)                                             # :92
    conv1 = _cbr3d(conv0, sd, p + ".conv1", 2, tr)
    conv2 = _cbr3d(conv1, sd, p + ".conv2", 1, tr)                                         # :93
    conv3 = _cbr3d(conv2, sd, p + ".conv3", 2, tr)
    conv4 = _cbr3d(conv3, sd, p + ".conv4", 1, tr)                                         # :94
    conv5 = _cbr3d(conv4, sd, p + ".conv5", 2, tr)
    conv6 = _cbr3d(conv5, sd, p + ".conv6", 1, tr)                                         # :96
    up7 = conv4 + _up3d(conv6, sd, p + ".conv7", tr)                                       # :97
    up9 = conv2 + _up3d(up7, sd, p + ".conv9", tr)                                         # :99
    up11 = conv0 + _up3d(up9, sd, p + ".conv11", tr)                                       # :101
    out = F.conv3d(up11, sd[p + ".prob.weight"], sd[p + ".prob.bias"], 1, 1)               # :103
    if return_intermediates:
        return out, {"conv0": conv0, "conv1": conv1, "conv2": conv2, "conv3": conv3, "conv4": conv4,
                     "conv5": conv5, "conv6": conv6, "up7": up7, "up9": up9, "up11": up11}
    return out


def cost_volume(feats, proj_mats, depth_values, num_groups):
    """mvsnet.py:134-172 in eval mode (the arithmetic of :150-162, written out of place).
    feats (B,V,C,h,w), proj_mats (B,V-1,3,4), depth_values (B,D,h,w)
    -> (B,C,D,h,w) variance (G == 1) or (B,G,D,h,w) group-wise correlation."""
    B, V, C, H, W = feats.shape
    D = depth_values.shape[1]
    ref_volume = feats[:, 0].unsqueeze(2).expand(-1, -1, D, -1, -1)                        # :137-138
    if num_groups == 1:
        volume_sum = ref_volume                                                            # :140
        volume_sq_sum = ref_volume ** 2                                                    # :141
    else:
        ref_volume = ref_volume.reshape(B, num_groups, C // num_groups, D, H, W)           # :143
        volume_sum = 0                                                                     # :144
    for v in range(1, V):                                                                  # :147
        warped = homo_warp(feats[:, v], proj_mats[:, v - 1], depth_values)                 # :148
        if num_groups == 1:
            volume_sum = volume_sum + warped                                               # :152 / :155
            volume_sq_sum = volume_sq_sum + warped ** 2                                    # :153 / :156
        else:
            volume_sum = volume_sum + warped.reshape(B, num_groups, C // num_groups, D, H, W)  # :160 / :162
    if num_groups == 1:
        return volume_sq_sum.div(V).sub(volume_sum.div(V).pow(2))                          # :167
    return (volume_sum * ref_volume).mean(2).div(V - 1)                                    # :170-171


def softmax_regress(cost, depth_values):
    """mvsnet.py:175-193.  cost, depth_values (B,D,h,w) -> depth, confidence (B,h,w), index (B,h,w) int64."""
    D = cost.shape[1]
    prob_volume = F.softmax(cost, 1)                                                       # :175
    depth = depth_regression(prob_volume, depth_values)                                    # :177
    sum4 = 4 * F.avg_pool3d(F.pad(prob_volume.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2)),
                            (4, 1, 1), stride=1).squeeze(1)                                # :181-183
    depth_index = depth_regression(prob_volume, torch.arange(D, dtype=prob_volume.dtype)).long()  # :185-189
    depth_index = torch.clamp(depth_index, 0, D - 1)                                       # :190
    confidence = torch.gather(sum4, 1, depth_index.unsqueeze(1)).squeeze(1)                # :192-193
    return depth, confidence, depth_index


def initial_depth_values(init_depth_min, depth_interval_l, D, B, h, w, dtype=torch.float32):
    """mvsnet.py:213-229."""
    steps = torch.arange(0, D, dtype=dtype)
    if isinstance(init_depth_min, float):
        dv = init_depth_min + depth_interval_l * steps                                     # :216-219
        return dv.reshape(1, D, 1, 1).expand(B, D, h, w).contiguous()                      # :220-221
    dv = init_depth_min + depth_interval_l * steps.reshape(1, D)                           # :223-227
    return dv.reshape(B, D, 1, 1).expand(B, D, h, w).contiguous()                          # :228-229


def cascade_forward(sd, imgs, proj_mats, init_depth_min, depth_interval, n_depths=(8, 32, 48),
                    interval_ratios=(1, 2, 4), num_groups=1, return_intermediates=False):
    """mvsnet.py:197-244 CascadeMVSNet.forward (eval mode, no_grad)."""
    B, V, _, H, W = imgs.shape
    results, inter = {}, {}
    with torch.no_grad():
        feats = feature_net(imgs.reshape(B * V, 3, H, W), sd)                              # :204-205
        depth_l = None
        for l in reversed(range(3)):                                                       # :207
            feats_l = feats[f"level_{l}"]
            feats_l = feats_l.view(B, V, *feats_l.shape[1:])                               # :209
            proj_mats_l = proj_mats[:, :, l]                                               # :210
            depth_interval_l = depth_interval * interval_ratios[l]                         # :211
            D = n_depths[l]
            h, w = feats_l.shape[-2:]
            if l == 2:
                depth_values = initial_depth_values(init_depth_min, depth_interval_l, D, B, h, w)
            else:
                depth_lm1 = F.interpolate(depth_l.unsqueeze(1), scale_factor=2, mode="bilinear",
                                          align_corners=True)                              # :232-234
                depth_values = get_depth_values(depth_lm1, D, depth_interval_l)            # :235
            volume = cost_volume(feats_l, proj_mats_l, depth_values, num_groups)           # :134-172
            cost = cost_reg_net(volume, sd, f"cost_reg_{l}").squeeze(1)                    # :174
            depth_l, confidence_l, index_l = softmax_regress(cost, depth_values)           # :175-193
            results[f"depth_{l}"] = depth_l
            results[f"confidence_{l}"] = confidence_l
            if return_intermediates:
                inter[f"feats_{l}"] = feats_l
                inter[f"depth_values_{l}"] = depth_values
                inter[f"volume_{l}"] = volume
                inter[f"cost_{l}"] = cost
                inter[f"index_{l}"] = index_l
    return (results, inter) if return_intermediates else results


def cascade_forward_train(sd, imgs, proj_mats, init_depth_min, depth_interval, n_depths=(8, 32, 48),
                          interval_ratios=(1, 2, 4), num_groups=1):
    """mvsnet.py:197-244 in TRAIN mode (train.py:99-103): batch-statistics ABN (the running statistics in `sd` are
    updated in place, like the modules' buffers), an autograd graph from the six outputs back to every tensor of `sd`
    that requires grad; the hypotheses of levels 1 and 0 come from the DETACHED previous depth (:231)."""
    B, V, _, H, W = imgs.shape
    results = {}
    feats = feature_net(imgs.reshape(B * V, 3, H, W), sd, training=True)                   # :204-205
    depth_l = None
    for l in reversed(range(3)):                                                           # :207
        feats_l = feats[f"level_{l}"]
        feats_l = feats_l.view(B, V, *feats_l.shape[1:])                                   # :209
        proj_mats_l = proj_mats[:, :, l]                                                   # :210
        depth_interval_l = depth_interval * interval_ratios[l]                             # :211
        D = n_depths[l]
        h, w = feats_l.shape[-2:]
        if l == 2:
            depth_values = initial_depth_values(init_depth_min, depth_interval_l, D, B, h, w, dtype=imgs.dtype)
        else:
            depth_lm1 = F.interpolate(depth_l.detach().unsqueeze(1), scale_factor=2, mode="bilinear",
                                      align_corners=True)                                  # :231-234
            depth_values = get_depth_values(depth_lm1, D, depth_interval_l)                # :235
        volume = cost_volume(feats_l, proj_mats_l, depth_values, num_groups)               # :150-153 / :159-160
        cost = cost_reg_net(volume, sd, f"cost_reg_{l}", training=True).squeeze(1)         # :174
        prob_volume = F.softmax(cost, 1)                                                   # :175
        depth_l = depth_regression(prob_volume, depth_values)                              # :177
        with torch.no_grad():                                                              # :179-193
            _, confidence_l, _ = softmax_regress(cost, depth_values)
        results[f"depth_{l}"] = depth_l
        results[f"confidence_{l}"] = confidence_l
    return results


def sgd_train_steps(sd0, imgs, proj_mats, init_depth_min, depth_interval, targets, steps, lr=1e-3, momentum=0.9, weight_decay=1e-5,
                    dtype=torch.float64, abs_weight_eps=None):
    """train.py:99-127 + opt.py:40-47 / utils/__init__.py:12-14 in miniature: `steps` optimisation steps of torch.optim.SGD(lr, momentum, weight_decay) - the
    reference's default optimiser - on ONE fixed batch, loss = sum_l SmoothL1(depth_l, targets[l]) * 2^(1 - l) (losses.py:4-19 without masks), in `dtype`
    (float64: the truth a float32 engine's trajectory is compared with).  SGD as torch implements it: d = grad + weight_decay * p; v = d on the first
    step, v = momentum * v + d afterwards; p -= lr * v.  abs_weight_eps: train.py:41 builds the model with InPlaceABN, whose scale is |weight| + eps
    (SURVEY appendix B): the 1-d `.weight` tensors enter the forward as |w| + eps and the gradient reaches w through that expression.
    -> list of the `steps` loss values (before each update)."""
    params = {k: v.clone().to(dtype).requires_grad_(True) for k, v in sd0.items() if v.dtype.is_floating_point and "running" not in k}
    bufs = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd0.items() if k not in params}
    velocity, losses = {}, []
    for _ in range(steps):
        sd = dict(bufs)
        for k, p in params.items():
            sd[k] = (p.abs() + abs_weight_eps) if (abs_weight_eps is not None and k.endswith(".weight") and p.dim() == 1) else p
        out = cascade_forward_train(sd, imgs.to(dtype), proj_mats.to(dtype), init_depth_min, depth_interval)
        loss = sum(F.smooth_l1_loss(out[f"depth_{l}"], targets[l].to(dtype)) * 2 ** (1 - l) for l in range(3))
        grads = torch.autograd.grad(loss, list(params.values()))
        with torch.no_grad():
            for (k, p), g in zip(params.items(), grads):
                d = g + weight_decay * p
                velocity[k] = d.clone() if k not in velocity else velocity[k] * momentum + d
                p -= lr * velocity[k]
        losses.append(float(loss.detach()))
    return losses
