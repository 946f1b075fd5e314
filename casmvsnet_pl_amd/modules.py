"""Host-side mirror of the reference's `models/modules.py` (same names, argument meaning and
return shapes), with the arithmetic done by the HIP engine through the C ABI.

reference                      here
modules.py:8-18   ConvBnReLU    parameter container (folded into the 2D MFMA conv epilogue, casmvs_conv2d_*)
modules.py:21-31  ConvBnReLU3D  parameter container (folded into the MFMA conv epilogue)
modules.py:34-49  get_depth_values  -> casmvs_depth_hypotheses_f32 (on an already-upsampled map)
modules.py:52-92  homo_warp         -> casmvs_homo_warp_f32
modules.py:95-104 depth_regression  -> device-side torch reduction, API parity only (the engine fuses
                                       softmax + regression + confidence in casmvs_softmax_regress_f32)
"""
import torch
import torch.nn as nn

from . import ops
from .inplace_abn import InPlaceABN


class ConvBnReLU(nn.Module):
    """modules.py:8-18.  Inside FeatureNet the layer runs as part of casmvs_featurenet_forward_f32; called on
    its own it is one casmvs_conv2d_forward_f32 launch (eval-mode ABN folded).  No torch / CPU path."""

    _KINDS = {(3, 1, 1): ops.CONV2D_K3, (5, 2, 2): ops.CONV2D_K5S2, (1, 1, 0): ops.CONV2D_K1}

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1, norm_act=InPlaceABN):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = norm_act(out_channels)
        self._geometry = (kernel_size, stride, pad)
        self._packed = None  # (key, device tensor, slope): re-packed when a parameter / buffer changes

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("casmvsnet_pl_amd.ConvBnReLU runs on the MI355X only; there is no CPU fallback")
        if self.training:   # batch statistics, autograd graph (training.py)
            from .training import conv_bn_relu_2d
            return conv_bn_relu_2d(self, x.float())
        kind = self._KINDS.get(self._geometry)
        if kind is None:
            raise RuntimeError(f"ConvBnReLU: (kernel, stride, pad) = {self._geometry} is not one of the layer shapes of "
                               "FeatureNet (3,1,1), (5,2,2), (1,1,0)")
        key = (str(x.device),) + tuple((t.data_ptr(), t._version) for t in self.state_dict(keep_vars=True).values())
        if self._packed is None or self._packed[0] != key:
            bn = self.bn
            if hasattr(bn, "folded_scale_shift"):
                scale, shift = bn.folded_scale_shift()
            else:
                s64 = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
                scale, shift = s64.float().cpu(), (bn.bias.detach().double() - bn.running_mean.detach().double() * s64).float().cpu()
            slope = bn.leaky_slope() if hasattr(bn, "leaky_slope") else float(getattr(bn, "activation_param", 0.01))
            self._packed = (key, ops.conv2d_pack(kind, self.conv.weight, scale, shift).to(x.device), slope)
        _, packed, slope = self._packed
        return ops.conv2d_forward(kind, packed, x.float(), self.conv.out_channels, slope=slope)


def _folded(bn):
    """Eval-mode ABN -> (scale, shift, slope) on the host."""
    if hasattr(bn, "folded_scale_shift"):
        scale, shift = bn.folded_scale_shift()
    else:
        s64 = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        scale, shift = s64.float().cpu(), (bn.bias.detach().double() - bn.running_mean.detach().double() * s64).float().cpu()
    slope = bn.leaky_slope() if hasattr(bn, "leaky_slope") else float(getattr(bn, "activation_param", 0.01))
    return scale, shift, slope


class ConvBnReLU3D(nn.Module):
    """modules.py:21-31.  Inside CostRegNet the layer runs as part of casmvs_costreg_forward_f32; called on its own
    it is one casmvs_conv3d_forward_f32 launch (eval-mode ABN folded into the MFMA conv epilogue)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1, norm_act=InPlaceABN):
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = norm_act(out_channels)
        self._geometry = (kernel_size, stride, pad)
        self._packed = None

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("casmvsnet_pl_amd.ConvBnReLU3D runs on the MI355X only; there is no CPU fallback")
        if self.training:   # batch statistics, autograd graph (training.py)
            from .training import conv_bn_relu_3d
            return conv_bn_relu_3d(self, x.float())
        kind = {(3, 1, 1): ops.CONV_S1, (3, 2, 1): ops.CONV_S2}.get(self._geometry)
        if kind is None:
            raise RuntimeError(f"ConvBnReLU3D: (kernel, stride, pad) = {self._geometry} is not a CostRegNet layer shape "
                               "(3,1,1) or (3,2,1)")
        key = (str(x.device),) + tuple((t.data_ptr(), t._version) for t in self.state_dict(keep_vars=True).values())
        if self._packed is None or self._packed[0] != key:
            scale, shift, slope = _folded(self.bn)
            self._packed = (key, ops.conv3d_pack(kind, self.conv.weight, scale, shift).to(x.device), slope)
        _, packed, slope = self._packed
        with torch.no_grad():
            return ops.conv3d_forward(kind, packed, x.float(), self.conv.out_channels, None, slope)


def _per_sample(value, B, device):
    """float or (B,1)/(B,) tensor -> (B,) float32 device vector."""
    if isinstance(value, torch.Tensor):
        return value.reshape(B).to(device=device, dtype=torch.float32)
    return torch.full((B,), float(value), dtype=torch.float32, device=device)


def get_depth_values(current_depth, n_depths, depth_interval):
    """current_depth (B,1,H,W); depth_interval (B,1) or float -> (B,D,H,W)   (modules.py:34-49)."""
    B, _, H, W = current_depth.shape
    dev = current_depth.device
    if isinstance(depth_interval, torch.Tensor):
        interval_b = depth_interval.reshape(B).float()
        half_b = (n_depths / 2) * interval_b
    else:
        interval_b = _per_sample(depth_interval, B, dev)
        half_b = _per_sample(n_depths / 2 * depth_interval, B, dev)  # python-double product, then fp32
    # prev (B,H,W) with hp == H, wp == W: the align_corners x1 "upsample" is the identity
    return ops.depth_hypotheses(current_depth.reshape(B, H, W), None, interval_b, half_b, n_depths, H, W)


def homo_warp(src_feat, proj_mat, depth_values):
    """src_feat (B,C,H,W), proj_mat (B,3,4), depth_values (B,D,H,W) -> (B,C,D,H,W)  (modules.py:52-92)."""
    return ops.homo_warp(src_feat, proj_mat, depth_values)


def depth_regression(p, depth_values):
    """p (B,D,H,W); depth_values (B,D,H,W) or (D) -> (B,H,W)   (modules.py:95-104), one casmvs_depth_regression_f32
    launch.  The engine's forward never calls it (softmax, regression and confidence are one kernel,
    casmvs_softmax_regress_f32); like every op here it refuses CPU tensors."""
    if not p.is_cuda:
        raise RuntimeError("casmvsnet_pl_amd.depth_regression runs on the MI355X only; there is no CPU fallback")
    if depth_values.dim() == 1 and depth_values.shape[0] == p.shape[1]:
        pass                                                   # (D): one depth per plane
    elif depth_values.numel() == p.shape[1] and depth_values.dim() != 4:
        depth_values = depth_values.reshape(-1)
    elif tuple(depth_values.shape) != tuple(p.shape):          # anything `p * depth_values` broadcasts: (1,D,1,1), (B,D,1,1), ...
        depth_values = depth_values.expand_as(p).contiguous()
    return ops.depth_regression(p.float(), depth_values.float()).to(depth_values.dtype)
