"""Host-side mirror of the reference's `models/mvsnet.py`: same classes, constructor arguments,
attribute names, state-dict keys (206 tensors) and `forward` contract, with the hot path
(plane-sweep warp, cost volume, CostRegNet, softmax regression) executed by the hand-written
gfx950 kernels of libcasmvs_hip.so.

reference                                   here
mvsnet.py:7-57     FeatureNet                parameter container + casmvs_featurenet_forward_f32
mvsnet.py:60-104   CostRegNet                parameter container + casmvs_costreg_forward_f32
mvsnet.py:125-195  CascadeMVSNet.predict_depth  casmvs_costvol_{var,gwc}_f32 -> CostRegNet ->
                                             casmvs_softmax_regress_f32
mvsnet.py:197-244  CascadeMVSNet.forward     same loop; hypotheses by casmvs_depth_hypotheses_f32

In eval mode this is the fused inference engine (eval-mode ABN folded into the conv epilogue, no autograd graph); in
train mode (`model.train()`, train.py:99-103) every module runs its differentiable HIP form (training.py: batch-statistics
ABN, convolution gradients on the matrix cores).  CPU tensors raise instead of silently falling back.
"""
import torch
import torch.nn as nn

from . import ops
from .inplace_abn import InPlaceABN
from .modules import ConvBnReLU, ConvBnReLU3D, _per_sample
from .profiling import stage


def _fold_norm(owner, norm):
    """Eval-mode ABN -> per-channel (scale, shift) on the host and the activation slope."""
    if not hasattr(norm, "running_mean"):
        raise RuntimeError(f"{owner}: norm_act {type(norm).__name__} has no running statistics to fold")
    if hasattr(norm, "folded_scale_shift"):
        scale, shift = norm.folded_scale_shift()
    else:  # any ABN-like module: weight/bias/running_mean/running_var/eps
        var = norm.running_var.detach().double()
        scale64 = norm.weight.detach().double() / torch.sqrt(var + norm.eps)
        shift = (norm.bias.detach().double() - norm.running_mean.detach().double() * scale64).float().cpu()
        scale = scale64.float().cpu()
    slope = norm.leaky_slope() if hasattr(norm, "leaky_slope") else float(getattr(norm, "activation_param", 0.01))
    return scale, shift, slope


def compose_fpn_tail(lat_weight, lat_bias, smooth_weight, smooth_bias):
    """The FPN tail smooth(lat(x) + up(y)) (mvsnet.py:36-38,50-51,54) as ONE 3x3 convolution over [x | up(y)]: nothing
    non-linear sits between the 1x1 lateral conv, the sum and the 3x3 smoothing conv.  Returns
      weight (cout, cin + cmid, 3, 3) = [ smooth_weight o lat_weight | smooth_weight ]   (composed in float64),
      bias9  (3, 3, cout): smooth_bias + the sum over the smoothing taps INSIDE the image of smooth_weight[.., ky, kx] . lat_bias
             for the row classes (first / inner / last row) x column classes - zero padding applies to the SUM, so the
             lateral bias only arrives through taps that exist.
    lat_weight (cmid, cin, 1, 1), lat_bias (cmid), smooth_weight (cout, cmid, 3, 3), smooth_bias (cout)."""
    Ws, Wl = smooth_weight.detach().double().cpu(), lat_weight.detach().double().cpu()[:, :, 0, 0]
    bl, bs = lat_bias.detach().double().cpu(), smooth_bias.detach().double().cpu()
    composed = torch.einsum("omyx,mi->oiyx", Ws, Wl)
    weight = torch.cat([composed, Ws], dim=1).float()
    tap_bias = torch.einsum("omyx,m->oyx", Ws, bl)                       # (cout, ky, kx)
    valid = {0: (1, 2), 1: (0, 1, 2), 2: (0, 1)}                         # taps inside the image for the first / inner / last row (column)
    bias9 = torch.stack([torch.stack([bs + sum(tap_bias[:, ky, kx] for ky in valid[r] for kx in valid[c]) for c in range(3)])
                         for r in range(3)]).float()
    return weight.contiguous(), bias9.contiguous()


class _PackedWeights:
    """Mixin of the modules that keep folded + packed device images of their parameters.

    The images are rebuilt when a parameter / buffer was REPLACED (`m.weight = nn.Parameter(..)`, pruning,
    parametrisation, a swapped sub-module: the key holds the identity of every current tensor), modified in place through
    autograd-visible operations (its storage pointer or `_version` changed - optimiser steps, `copy_`,
    `load_state_dict`), when the module is moved / cast (`_apply`) and when a state dict is loaded.  An edit THROUGH
    `.data` (`p.data.mul_(..)`, common in EMA / weight-surgery code) bumps no version counter: call
    `invalidate_packed()` after it."""

    def _init_packed(self):
        self._packed = None
        self._packed_key = None
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    def invalidate_packed(self):
        self._packed_key = None   # the images are re-packed on the next use (in place where the shapes still fit)

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_packed()
        return super()._apply(fn, *args, **kwargs)

    def _store_packed(self, attr, new):
        """Keep the device images at their ADDRESSES when the shapes allow it (copy_ in place): a captured hipGraph
        (graph.py) holds raw pointers to them, so a re-pack after a weight update is what its next replay reads.
        `new`: a tensor, or a list / tuple of tensors."""
        old = getattr(self, attr, None)
        olds = list(old) if isinstance(old, (list, tuple)) else ([old] if isinstance(old, torch.Tensor) else None)
        news = list(new) if isinstance(new, (list, tuple)) else [new]
        same = lambda o, n: (o is None and n is None) or (o is not None and n is not None and o.shape == n.shape and o.device == n.device and o.dtype == n.dtype)
        if olds is not None and len(olds) == len(news) and all(same(o, n) for o, n in zip(olds, news)):   # (an entry may be None: a layer without that image)
            for o, n in zip(olds, news):
                if o is not None:
                    o.copy_(n)
            return old
        setattr(self, attr, new)
        return new

    def _split_image(self, attr, pack, wanted):
        """The split-f16 / split-bf16 image(s) `attr` of a layer set: packed only when the selected modes use them (`wanted`), kept at their
        addresses like every packed image (_store_packed).  The f16 / bf16 packers reject non-finite weights: such a model runs the layer on the
        float32 MFMA kernel, which propagates Inf / NaN through the taps that touch it as the reference does - unless an image already
        exists (a captured hipGraph may hold its address and cannot change kernels: that is an error)."""
        if not wanted:
            return None
        try:
            return self._store_packed(attr, pack())
        except RuntimeError as e:
            if "finite" not in str(e):
                raise
            if getattr(self, attr, None) is not None:
                raise RuntimeError(f"{type(self).__name__}: non-finite weights cannot be re-packed for the f16 matrix cores and a packed image "
                                   f"(possibly captured in a hipGraph) exists; select the float32 modes and re-capture") from e
            return None

    def _state_key(self, device):
        # walks the CURRENT module tree on every call (~40 us): a cached tensor list would keep answering for tensor
        # objects that are no longer the module's (round-2 advisor finding)
        return (str(device),) + tuple((id(t), t.data_ptr(), t._version) for m in self.modules()
                                      for t in (*m._parameters.values(), *m._buffers.values()) if t is not None)


class FeatureNet(_PackedWeights, nn.Module):
    """3-level FPN feature extractor (mvsnet.py:7-57): same layers / state-dict keys as the reference;
    `forward` runs the 13 layers as MFMA kernels (casmvs_featurenet_forward_f32) with eval-mode ABN
    folded into the conv epilogue and each FPN upsample-add fused into its lateral 1x1 conv."""

    # (attribute path, kind) in the order casmvs_featurenet_forward_f32 expects
    _LAYERS = (("conv0.0", ops.CONV2D_K3), ("conv0.1", ops.CONV2D_K3),
               ("conv1.0", ops.CONV2D_K5S2), ("conv1.1", ops.CONV2D_K3), ("conv1.2", ops.CONV2D_K3),
               ("conv2.0", ops.CONV2D_K5S2), ("conv2.1", ops.CONV2D_K3), ("conv2.2", ops.CONV2D_K3),
               ("toplayer", ops.CONV2D_K1), ("lat1", ops.CONV2D_K1_UP), ("lat0", ops.CONV2D_K1_UP),
               ("smooth1", ops.CONV2D_K3), ("smooth0", ops.CONV2D_K3))
    LAYER_NAMES = tuple(n for n, _ in _LAYERS)

    def __init__(self, norm_act=InPlaceABN):
        super().__init__()
        self.conv0 = nn.Sequential(
            ConvBnReLU(3, 8, 3, 1, 1, norm_act=norm_act),
            ConvBnReLU(8, 8, 3, 1, 1, norm_act=norm_act))
        self.conv1 = nn.Sequential(
            ConvBnReLU(8, 16, 5, 2, 2, norm_act=norm_act),
            ConvBnReLU(16, 16, 3, 1, 1, norm_act=norm_act),
            ConvBnReLU(16, 16, 3, 1, 1, norm_act=norm_act))
        self.conv2 = nn.Sequential(
            ConvBnReLU(16, 32, 5, 2, 2, norm_act=norm_act),
            ConvBnReLU(32, 32, 3, 1, 1, norm_act=norm_act),
            ConvBnReLU(32, 32, 3, 1, 1, norm_act=norm_act))
        self.toplayer = nn.Conv2d(32, 32, 1)
        self.lat1 = nn.Conv2d(16, 32, 1)
        self.lat0 = nn.Conv2d(8, 32, 1)
        self.smooth1 = nn.Conv2d(32, 16, 3, padding=1)
        self.smooth0 = nn.Conv2d(32, 8, 3, padding=1)
        self._init_packed()
        self._workspace = None
        self._slope = 0.01
        self._fused0 = None       # (packed 40-channel 3x3 layer, bias classes) of the fused full-resolution tail
        self.fuse_tail = True     # lat0 + upsample-add + smooth0 as one kernel (False: the reference's three steps, A/B and tests)
        self._fused0_sf = None    # the same tail as the split-f16 image
        self.fuse_conv0 = True    # conv0.0 + conv0.1 as one kernel on the f16 matrix cores (with tail_mode "splitf16"; False: two float32-MFMA layers, A/B and tests)
        self.tail_mode = "splitf16"   # arithmetic of the fused tail: "splitf16" (f16 matrix cores, fpn_fused_sf.hip) or "f32"
        self._ci2d = None         # split-f16 images of conv1.1, conv1.2, conv2.1, conv2.2, smooth1 (conv2d_ci_splitf16.hip), conv1.0, conv2.0, conv0 (follow tail_mode)
        self._split_active = False   # set by packed_layers: the split-f16 images are packed and current
        self.timer = None         # optional profiling.StageTimer (bench.py)
        self.last_channels_last = None

    def packed_layers(self, device):
        """Folded + packed parameter images on `device` (re-packed whenever a tensor changed)."""
        sf = self.fuse_tail and self.tail_mode == "splitf16"
        key = self._state_key(device) + (bool(self.fuse_tail), sf, bool(self.fuse_conv0))
        if self._packed is not None and key == self._packed_key:
            return self._packed
        packed, slopes = [], set()
        for name, kind in self._LAYERS:
            m = self.get_submodule(name)
            if isinstance(m, ConvBnReLU):
                scale, shift, slope = _fold_norm(f"FeatureNet.{name}", m.bn)
                slopes.add(slope)
                packed.append(ops.conv2d_pack(kind, m.conv.weight, scale, shift).to(device))
            else:
                packed.append(ops.conv2d_pack(kind, m.weight, None, m.bias).to(device))
        if len(slopes) > 1:
            raise RuntimeError("FeatureNet: all ABN layers must share one activation slope")
        self._slope = slopes.pop() if slopes else 0.01
        # the full-resolution tail lat0 + upsample-add + smooth0 as one 40-channel 3x3 layer (csrc/fpn_fused.hip); only the images of the
        # selected arithmetic are packed (an all-float32 replica of graph.ConcurrentForwards never pays for - or trips over - the f16 packers)
        if self.fuse_tail:
            w40, bias9 = compose_fpn_tail(self.lat0.weight, self.lat0.bias, self.smooth0.weight, self.smooth0.bias)
            self._store_packed("_fused0", (ops.conv2d_pack(ops.CONV2D_K3, w40, None, None).to(device), bias9.to(device)))
            if self._split_image("_fused0_sf", lambda: (ops.fpn_tail0_splitf16_pack(w40).to(device), bias9.to(device)), sf) is None:
                sf = False

        def pack_ci():
            ci = []
            for name in ("conv1.1", "conv1.2", "conv2.1", "conv2.2"):
                m = self.get_submodule(name)
                sc, sh, _ = _fold_norm(f"FeatureNet.{name}", m.bn)
                ci.append(ops.conv2d_ci_splitf16_pack(m.conv.weight, sc, sh).to(device))
            ci.append(ops.conv2d_ci_splitf16_pack(self.smooth1.weight, None, self.smooth1.bias).to(device))   # smooth1: Conv2d 32 -> 16 with bias
            for name in ("conv1.0", "conv2.0"):   # the 5 x 5 stride-2 layers (conv2d_k5s2_splitf16.hip)
                m = self.get_submodule(name)
                sc, sh, _ = _fold_norm(f"FeatureNet.{name}", m.bn)
                ci.append(ops.conv2d_k5s2_splitf16_pack(m.conv.weight, sc, sh).to(device))
            # conv0.0 + conv0.1 as one kernel (fnet_conv0_mm.hip): the 8-channel map between them never reaches memory
            (sc0, sh0, _), (sc1, sh1, _) = _fold_norm("FeatureNet.conv0.0", self.conv0[0].bn), _fold_norm("FeatureNet.conv0.1", self.conv0[1].bn)
            ci.append(ops.fnet_conv0_mm_pack(self.conv0[0].conv.weight, sc0, sh0, self.conv0[1].conv.weight, sc1, sh1).to(device) if self.fuse_conv0 else None)
            return ci
        self._split_image("_ci2d", pack_ci, sf)
        self._split_active = sf   # the split-f16 images exist and are current: forward may select them
        self._packed_key = key
        return self._store_packed("_packed", packed)

    def forward(self, x, pixel_major_only=False):
        """x (N, 3, H, W) -> {"level_0": (N,8,H,W), "level_1": (N,16,H/2,W/2), "level_2": (N,32,H/4,W/4)}.
        pixel_major_only (eval mode; CascadeMVSNet.forward's own call): -> {"level_l": (N,h,w,C)} - the layout the plane sweep gathers; the (N,C,h,w)
        stores of levels 0 / 1, which nothing downstream reads, are dropped."""
        if not x.is_cuda:
            raise RuntimeError("casmvsnet_pl_amd.FeatureNet runs on the MI355X only; there is no CPU fallback")
        self.last_channels_last = None   # the pixel-major maps of THIS call or none: never those of earlier images (the train-mode branch produces none)
        if self.training:   # batch statistics + autograd graph (training.py); eval mode = the fused engine below
            from .training import feature_net_train
            return feature_net_train(self, x.float())
        N, _, H, W = x.shape
        if self.tail_mode not in ("splitf16", "f32"):
            raise ValueError(f"FeatureNet.tail_mode={self.tail_mode!r} (splitf16 or f32)")
        packed = self.packed_layers(x.device)
        need = ops.featurenet_workspace_bytes(N, H, W)
        ws = self._workspace
        if ws is None or ws.device != x.device or ws.numel() < need:
            ws = self._workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        events = self.timer.layer_events("feature", 14, self.LAYER_NAMES) if self.timer is not None else None
        sf = self._split_active   # fuse_tail and tail_mode == "splitf16" and finite weights (packed_layers)
        fused0 = (self._fused0_sf if sf else self._fused0) if self.fuse_tail else None
        feat0, feat1, feat2, cl = ops.featurenet_forward(packed, x.float(), ws, slope=self._slope, layer_events=events,
                                                         channels_last_copies=True, fused0=fused0, fused0_splitf16=sf, ci_layers=self._ci2d if sf else None,
                                                         nchw_outputs=not pixel_major_only)
        # pixel-major copies of the three maps (same kernels, second store): what the cost-volume gather reads
        self.last_channels_last = {"level_0": cl[0], "level_1": cl[1], "level_2": cl[2]}
        if pixel_major_only:
            return self.last_channels_last
        return {"level_0": feat0, "level_1": feat1, "level_2": feat2}


class CostRegNet(_PackedWeights, nn.Module):
    """3D U-Net regulariser (mvsnet.py:60-104).  Parameters live in torch modules with the
    reference's names; `forward` runs the fused MFMA engine on folded, pre-packed weights."""

    # (attribute, kind, has_abn) in the order casmvs_costreg_forward_f32 expects
    _LAYERS = (("conv0", ops.CONV_S1), ("conv1", ops.CONV_S2), ("conv2", ops.CONV_S1),
               ("conv3", ops.CONV_S2), ("conv4", ops.CONV_S1), ("conv5", ops.CONV_S2),
               ("conv6", ops.CONV_S1), ("conv7", ops.CONV_T2), ("conv9", ops.CONV_T2),
               ("conv11", ops.CONV_T2), ("prob", ops.CONV_S1))

    def __init__(self, in_channels, norm_act=InPlaceABN):
        super().__init__()
        self.conv0 = ConvBnReLU3D(in_channels, 8, norm_act=norm_act)
        self.conv1 = ConvBnReLU3D(8, 16, stride=2, norm_act=norm_act)
        self.conv2 = ConvBnReLU3D(16, 16, norm_act=norm_act)
        self.conv3 = ConvBnReLU3D(16, 32, stride=2, norm_act=norm_act)
        self.conv4 = ConvBnReLU3D(32, 32, norm_act=norm_act)
        self.conv5 = ConvBnReLU3D(32, 64, stride=2, norm_act=norm_act)
        self.conv6 = ConvBnReLU3D(64, 64, norm_act=norm_act)
        self.conv7 = nn.Sequential(
            nn.ConvTranspose3d(64, 32, 3, padding=1, output_padding=1, stride=2, bias=False),
            norm_act(32))
        self.conv9 = nn.Sequential(
            nn.ConvTranspose3d(32, 16, 3, padding=1, output_padding=1, stride=2, bias=False),
            norm_act(16))
        self.conv11 = nn.Sequential(
            nn.ConvTranspose3d(16, 8, 3, padding=1, output_padding=1, stride=2, bias=False),
            norm_act(8))
        self.prob = nn.Conv3d(8, 1, 3, stride=1, padding=1)
        self._init_packed()       # _packed: list of 11 device tensors
        self._conv0_sb = None     # conv0's split-bf16 image (uint8 device tensor)
        self._conv0_sf = None     # conv0's split-f16 image
        self._ci_sf = None        # (conv2, conv4, conv6, conv9, conv11) split-f16 images
        # conv2 / conv4 / conv6 (conv_ci_splitf16.hip) and the transposed conv9 / conv11 (deconv9_splitf16.hip, deconv11_splitf16.hip) in `regress`:
        # "splitf16" (f16 matrix cores) or "f32"
        self.ci_mode = "splitf16"
        self._s2_sf = None        # (conv1, conv3) split-f16 images
        # the stride-2 conv1 / conv3 (conv_s2_splitf16.hip) in `regress`: None (= ci_mode), "splitf16" or "f32"
        self.s2_mode = None
        # conv0's arithmetic in `regress` (the engine's eval path), all float32-grade (distance to a float64 convolution at or
        # below the float32 MFMA kernel's):
        #   "splitf16":  f16 matrix cores, every float32 operand as two float16 slices behind exact power-of-two scalings (per
        #                weight tensor / per staged tile), three partial products per product, float32 accumulation; cin = 8 / 16 (cascade
        #                levels 0 / 1) on the z-marching kernel (conv0_zmarch.hip), cin = 32 on the tiled one (conv0_splitf16.hip)
        #   "splitbf16": bf16 matrix cores, three exact bf16 slices per operand, six partial products
        #   "f32":       the float32 MFMA kernel like every other layer
        self.conv0_mode = "splitf16"
        self._workspace = None
        self.timer = None         # optional profiling.StageTimer (bench.py)
        self.timer_name = "costreg"
        self._conv0_active = None   # set by packed_layers: which split image of conv0 is packed and current ("splitf16" / "splitbf16" / None)
        self._ci_active = False     # ... and whether the five images of _ci_sf are
        self._s2_active = False     # ... and the two of _s2_sf

    # -- weight folding / packing -------------------------------------------------------------
    def _layer_tensors(self, name):
        m = getattr(self, name)
        if name == "prob":
            return m.weight, None, m.bias
        if isinstance(m, ConvBnReLU3D):
            return m.conv.weight, m.bn, None
        return m[0].weight, m[1], None

    def packed_layers(self, device):
        """Folded + packed parameter images on `device` (re-packed whenever a tensor changed)."""
        if self.conv0_mode not in ("splitf16", "splitbf16", "f32"):
            raise ValueError(f"CostRegNet.conv0_mode={self.conv0_mode!r} (splitf16, splitbf16 or f32)")
        if self.ci_mode not in ("splitf16", "f32"):
            raise ValueError(f"CostRegNet.ci_mode={self.ci_mode!r} (splitf16 or f32)")
        if self.s2_mode not in (None, "splitf16", "f32"):
            raise ValueError(f"CostRegNet.s2_mode={self.s2_mode!r} (None = ci_mode, splitf16 or f32)")
        key = self._state_key(device) + (self.conv0_mode, self.ci_mode, self.s2_mode)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        packed, slopes = [], set()
        for name, kind in self._LAYERS:
            weight, norm, bias = self._layer_tensors(name)
            if norm is not None:
                scale, shift, slope = _fold_norm(f"CostRegNet.{name}", norm)
                slopes.add(slope)
            else:
                scale, shift = None, bias
            packed.append(ops.conv3d_pack(kind, weight, scale, shift).to(device))
        if len(slopes) > 1:
            raise RuntimeError("CostRegNet: all ABN layers must share one activation slope")
        self._slope = slopes.pop() if slopes else 0.01
        # the images of the layers' f16 / bf16 matrix-core forms: only those the selected modes use (an all-float32 replica of
        # graph.ConcurrentForwards packs none), float32 fallback for non-finite weights (_split_image)
        cin = self.conv0.conv.weight.shape[1]
        fold0 = lambda: _fold_norm("CostRegNet.conv0", self.conv0.bn)[:2]
        self._conv0_active = None
        if cin in (8, 16, 32) and self.conv0_mode == "splitbf16":
            if self._split_image("_conv0_sb", lambda: ops.conv0_splitbf16_pack(self.conv0.conv.weight, *fold0()).to(device), True) is not None:
                self._conv0_active = "splitbf16"
        elif cin in (8, 16, 32) and self.conv0_mode == "splitf16":
            if self._split_image("_conv0_sf", lambda: ops.conv0_splitf16_pack(self.conv0.conv.weight, *fold0()).to(device), True) is not None:
                self._conv0_active = "splitf16"

        def pack_ci():
            ci = []
            for name in ("conv2", "conv4", "conv6"):
                m = getattr(self, name)
                sc, sh, _ = _fold_norm(f"CostRegNet.{name}", m.bn)
                ci.append(ops.conv_ci_splitf16_pack(m.conv.weight, sc, sh).to(device))
            s9, b9, _ = _fold_norm("CostRegNet.conv9", self.conv9[1])
            ci.append(ops.deconv9_splitf16_pack(self.conv9[0].weight, s9, b9).to(device))
            s11, b11, _ = _fold_norm("CostRegNet.conv11", self.conv11[1])
            ci.append(ops.deconv11_splitf16_pack(self.conv11[0].weight, s11, b11).to(device))
            return ci
        self._ci_active = self._split_image("_ci_sf", pack_ci, self.ci_mode == "splitf16") is not None

        def pack_s2():
            s2 = []
            for name in ("conv1", "conv3"):
                m = getattr(self, name)
                sc, sh, _ = _fold_norm(f"CostRegNet.{name}", m.bn)
                s2.append(ops.conv_s2_splitf16_pack(m.conv.weight, sc, sh).to(device))
            return s2
        self._s2_active = self._split_image("_s2_sf", pack_s2, (self.s2_mode or self.ci_mode) == "splitf16") is not None
        self._packed_key = key
        return self._store_packed("_packed", packed)

    def forward(self, x):
        """x (B, Cin, D, h, w) -> (B, 1, D, h, w)."""
        if self.training:   # batch statistics + autograd graph (training.py); eval mode = the fused engine below
            from .training import cost_reg_net_train
            return cost_reg_net_train(self, x)
        B, _, D, h, w = x.shape
        packed = self.packed_layers(x.device)
        need = ops.costreg_workspace_bytes(B, D, h, w)
        ws = self._workspace
        if ws is None or ws.device != x.device or ws.numel() < need:
            ws = self._workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        events = self.timer.layer_events(self.timer_name) if self.timer is not None else None
        cost = ops.costreg_forward(packed, x, ws, slope=self._slope, layer_events=events)
        return cost.unsqueeze(1)

    def regress(self, x, depth_values, return_index=False):
        """CostRegNet + softmax / depth regression / confidence (mvsnet.py:174-193) in one library call: x (B,Cin,D,h,w),
        depth_values (B,D,h,w) -> cost (B,D,h,w), depth (B,h,w), confidence (B,h,w) [, index].  Eval mode only."""
        B, _, D, h, w = x.shape
        packed = self.packed_layers(x.device)
        need = ops.costreg_workspace_bytes(B, D, h, w)
        ws = self._workspace
        if ws is None or ws.device != x.device or ws.numel() < need:
            ws = self._workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        events = self.timer.layer_events(self.timer_name) if self.timer is not None else None
        split, arith = None, ops.CONV0_F32
        if self._conv0_active == "splitf16":
            split, arith = self._conv0_sf, ops.CONV0_SPLIT_F16
        elif self._conv0_active == "splitbf16":
            split, arith = self._conv0_sb, ops.CONV0_SPLIT_BF16
        c2, c4, c6, c9, c11 = self._ci_sf if self._ci_active else (None,) * 5
        c1, c3 = self._s2_sf if self._s2_active else (None, None)
        return ops.costreg_regress(packed, x, depth_values, ws, slope=self._slope, layer_events=events, return_index=return_index,
                                   conv0_split=split, conv0_arith=arith, conv2_split=c2, conv4_split=c4, conv6_split=c6, conv9_split=c9, conv11_split=c11,
                                   conv1_split=c1, conv3_split=c3)


class CascadeMVSNet(nn.Module):
    """Cascade MVSNet with the reference's constructor / forward signature (mvsnet.py:107-244)."""

    def __init__(self, n_depths=[8, 32, 48], interval_ratios=[1, 2, 4], num_groups=1, norm_act=InPlaceABN):
        super().__init__()
        self.levels = 3
        self.n_depths = n_depths
        self.interval_ratios = interval_ratios
        self.G = num_groups
        self.feature = FeatureNet(norm_act)
        for l in range(self.levels):
            cost_reg_l = CostRegNet(self.G if self.G > 1 else 8 * 2 ** l, norm_act)
            setattr(self, f"cost_reg_{l}", cost_reg_l)
        self.timer = None        # optional profiling.StageTimer: HIP events around every stage
        self.last_index = {}     # level -> (B,h,w) int32 depth index, filled when keep_index is set
        self.keep_index = False
        self.view_shard_group = None   # a torch.distributed group: split the source views over its ranks (dist.py)
        self.keep_cost = False   # parity tests: keep the regularised cost (B,D,h,w) of every level in last_cost
        self.fuse_regress = True  # eval mode: `prob` + softmax regression in one library call (CostRegNet.regress)
        self.last_cost = {}
        self._const_cache = {}

    def _const(self, value, B, device):
        """(B,) device vector filled with a python float, cached: the per-level depth ranges of a scene are
        the same call after call and a `torch.full` is a 5 us kernel launch."""
        key = (float(value), B, str(device))
        t = self._const_cache.get(key)
        if t is None:
            if len(self._const_cache) > 64:
                self._const_cache.clear()
            t = self._const_cache[key] = _per_sample(value, B, device)
        return t

    def set_timer(self, timer):
        self.timer = timer
        self.feature.timer = timer
        for l in range(self.levels):
            m = getattr(self, f"cost_reg_{l}")
            m.timer, m.timer_name = timer, f"costreg_{l}"

    def predict_depth(self, feats, proj_mats, depth_values, cost_reg, level=None, feats_channels_last=None):
        """feats (B,V,C,h,w), proj_mats (B,V-1,3,4), depth_values (B,D,h,w) -> depth, confidence (B,h,w).
        feats_channels_last: optional (B,V,h,w,C) copy of feats (FeatureNet writes one): the faster gather."""
        t = self.timer
        with stage(t, f"costvol_{level}"):                                  # mvsnet.py:134-172
            if feats is None:   # the engine's own call: FeatureNet stored the pixel-major maps only
                B, V, h, w, C = feats_channels_last.shape
            else:
                B, V, C, h, w = feats.shape
            if feats_channels_last is None and C in (8, 16, 32):
                feats_channels_last = ops.nchw_to_nhwc(feats.reshape(B * V, C, h, w)).view(B, V, h, w, C)
            if self.view_shard_group is not None:
                from .dist import view_sharded_cost_volume
                volume = view_sharded_cost_volume(feats_channels_last, proj_mats, depth_values, self.G, self.view_shard_group)
            elif feats_channels_last is not None:
                volume = ops.costvol(feats_channels_last, proj_mats, depth_values, self.G, channels_last=True)
            else:
                volume = ops.costvol(feats, proj_mats, depth_values, self.G)
        if self.fuse_regress and isinstance(cost_reg, CostRegNet) and not cost_reg.training:
            # mvsnet.py:174-193 inside the CostRegNet call: the `prob` head hands its cost values to the regression
            out = cost_reg.regress(volume, depth_values, return_index=self.keep_index)
            cost, depth, confidence = out[:3]
            if self.keep_index:
                self.last_index[level] = out[3]
            if self.keep_cost:
                self.last_cost[level] = cost
            return depth, confidence
        cost = cost_reg(volume).squeeze(1)                                  # mvsnet.py:174
        if self.keep_cost:
            self.last_cost[level] = cost
        with stage(t, f"softmax_{level}"):
            if self.keep_index:
                depth, confidence, index = ops.softmax_regress(cost, depth_values, return_index=True)
                self.last_index[level] = index
            else:
                depth, confidence = ops.softmax_regress(cost, depth_values)  # mvsnet.py:175-193
        return depth, confidence

    def forward(self, imgs, proj_mats, init_depth_min, depth_interval):
        """imgs (B,V,3,H,W); proj_mats (B,V-1,levels,3,4) fine->coarse; init_depth_min,
        depth_interval: float or (B,1) tensor.  Returns {"depth_l", "confidence_l"} for l in 0..2."""
        if self.training:
            # train mode (train.py:99-103): batch-statistics ABN and an autograd graph, every tensor-sized operation a HIP
            # kernel (casmvsnet_pl_amd/training.py).  The fused inference engine below is the eval-mode path.
            if not imgs.is_cuda:
                raise RuntimeError("casmvsnet_pl_amd.CascadeMVSNet runs on the MI355X only; there is no CPU fallback")
            from .training import cascade_forward_train
            return cascade_forward_train(self, imgs, proj_mats, init_depth_min, depth_interval)
        if not imgs.is_cuda:
            raise RuntimeError("casmvsnet_pl_amd.CascadeMVSNet runs on the MI355X only: move the model and inputs "
                               "to 'cuda' (ROCm). There is no CPU fallback.")
        B, V, _, H, W = imgs.shape
        dev = imgs.device
        results = {}
        imgs = imgs.reshape(B * V, 3, H, W).float()
        proj_mats = proj_mats.float().permute(2, 0, 1, 3, 4).contiguous()  # (levels, B, V-1, 3, 4): one copy, not one per level
        t = self.timer
        with torch.no_grad():
            # (a user's replacement module, or FeatureNet left in train mode, returns the reference's (N,C,h,w) dict)
            engine_feats = type(self.feature) is FeatureNet and not self.feature.training
            with stage(t, "feature"):
                feats = self.feature(imgs, pixel_major_only=True) if engine_feats else self.feature(imgs)
            depth_l = None
            for l in reversed(range(self.levels)):
                feats_l = feats[f"level_{l}"]
                if engine_feats:
                    (h, w, C), cl, feats_l = feats_l.shape[1:], feats_l.view(B, V, *feats_l.shape[1:]), None
                else:
                    C, h, w = feats_l.shape[1:]
                    feats_l = feats_l.reshape(B, V, C, h, w)
                    cl = getattr(self.feature, "last_channels_last", None)
                    cl = None if cl is None else cl[f"level_{l}"].view(B, V, h, w, C)
                proj_mats_l = proj_mats[l]
                D = self.n_depths[l]
                ratio = self.interval_ratios[l]
                if isinstance(depth_interval, torch.Tensor):
                    interval_b = depth_interval.reshape(B).to(dev, torch.float32) * ratio  # mvsnet.py:211
                    half_b = (D / 2) * interval_b                                          # modules.py:44
                else:
                    depth_interval_l = depth_interval * ratio
                    interval_b = self._const(depth_interval_l, B, dev)
                    half_b = self._const(D / 2 * depth_interval_l, B, dev)
                with stage(t, f"hypotheses_{l}"):
                    if l == self.levels - 1:
                        dmin_b = (_per_sample(init_depth_min, B, dev) if isinstance(init_depth_min, torch.Tensor)
                                  else self._const(init_depth_min, B, dev))
                        depth_values = ops.depth_hypotheses(None, dmin_b, interval_b, None, D, h, w)
                    else:
                        depth_values = ops.depth_hypotheses(depth_l, None, interval_b, half_b, D, h, w)
                depth_l, confidence_l = self.predict_depth(feats_l, proj_mats_l, depth_values,
                                                           getattr(self, f"cost_reg_{l}"), level=l,
                                                           feats_channels_last=cl)
                results[f"depth_{l}"] = depth_l
                results[f"confidence_{l}"] = confidence_l
        return results
