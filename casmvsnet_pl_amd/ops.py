"""Tensor-level wrappers over the C ABI (include/casmvs.h): torch is used only for device memory
and the current HIP stream; every op below is one call into libcasmvs_hip.so.

All inputs must be float32 tensors on a ROCm device ("cuda" in torch); nothing here runs on CPU.
"""
import ctypes

import torch

from . import _lib, streams
from ._lib import (CONV_S1, CONV_S2, CONV_T2, CONV2D_K3, CONV2D_K5S2, CONV2D_K1,  # noqa: F401  (re-exported)
                   CONV2D_K1_UP)


def _dev(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: casmvsnet_pl_amd ops run on the MI355X only (got a {t.device} tensor); "
                           "there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t.contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t, f16=False):
    """hipStream_t of the launch: torch's current stream, after the cross-stream guard of streams.py (f16: the call launches kernels with
    f16 / bf16 matrix instructions)."""
    return streams.launch_stream(t, f16)


def homo_warp(src_feat, proj_mat, depth_values, impl="auto"):
    """modules.py:52-92.  (B,C,H,W), (B,3,4), (B,D,H,W) -> (B,C,D,H,W).
    impl: "lds" = casmvs_homo_warp_lds_f32 (the source box staged in LDS straight from the channel planes of `src_feat`: no layout pass),
    "lds_copy" = pixel-major copy of src + casmvs_homo_warp_nhwc_f32 (round 2-3's form, kept for A/B), "gather" = casmvs_homo_warp_f32
    (NCHW gathers), "auto" = "lds" when the shape has an LDS plan.  Bit-identical results."""
    src_feat, proj_mat, depth_values = _dev(src_feat, "src_feat"), _dev(proj_mat, "proj_mat"), _dev(depth_values, "depth_values")
    B, C, H, W = src_feat.shape
    D = depth_values.shape[1]
    if proj_mat.shape != (B, 3, 4) or depth_values.shape != (B, D, H, W):
        raise ValueError(f"homo_warp: shapes {tuple(src_feat.shape)} {tuple(proj_mat.shape)} {tuple(depth_values.shape)}")
    lib = _lib.load()
    if impl == "auto":
        impl = "lds" if C in (8, 16, 32) and lib.casmvs_homo_warp_lds_supported(C, W, D) else "gather"
    out = torch.empty((B, C, D, H, W), dtype=torch.float32, device=src_feat.device)
    with torch.cuda.device(src_feat.device):
        if impl == "lds":
            rc = lib.casmvs_homo_warp_lds_f32(_ptr(src_feat), _ptr(proj_mat), _ptr(depth_values), _ptr(out),
                                              B, C, H, W, D, _stream(src_feat))
            _lib.check(rc, "casmvs_homo_warp_lds_f32")
        elif impl == "lds_copy":
            nhwc = nchw_to_nhwc(src_feat)
            rc = lib.casmvs_homo_warp_nhwc_f32(_ptr(nhwc), _ptr(proj_mat), _ptr(depth_values), _ptr(out),
                                               B, C, H, W, D, _stream(src_feat))
            _lib.check(rc, "casmvs_homo_warp_nhwc_f32")
        else:
            rc = lib.casmvs_homo_warp_f32(_ptr(src_feat), _ptr(proj_mat), _ptr(depth_values), _ptr(out),
                                          B, C, H, W, D, _stream(src_feat))
            _lib.check(rc, "casmvs_homo_warp_f32")
    return out


def homo_warp_nhwc(src_nhwc, proj_mat, depth_values):
    """modules.py:52-92 on a pixel-major source map (B,H,W,C) -> (B,C,D,H,W): casmvs_homo_warp_nhwc_f32
    (LDS-staged).  Raises for shapes without an LDS plan (use homo_warp)."""
    src, proj_mat, depth_values = _dev(src_nhwc, "src_nhwc"), _dev(proj_mat, "proj_mat"), _dev(depth_values, "depth_values")
    B, H, W, C = src.shape
    D = depth_values.shape[1]
    if proj_mat.shape != (B, 3, 4) or depth_values.shape != (B, D, H, W):
        raise ValueError(f"homo_warp_nhwc: shapes {tuple(src.shape)} {tuple(proj_mat.shape)} {tuple(depth_values.shape)}")
    out = torch.empty((B, C, D, H, W), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = _lib.load().casmvs_homo_warp_nhwc_f32(_ptr(src), _ptr(proj_mat), _ptr(depth_values), _ptr(out), B, C, H, W, D, _stream(src))
    _lib.check(rc, "casmvs_homo_warp_nhwc_f32")
    return out


def nchw_to_nhwc(x):
    """(N, C, h, w) -> (N, h, w, C) device copy (casmvs_nchw_to_nhwc_f32), C in {8, 16, 32}."""
    x = _dev(x, "x")
    N, C, h, w = x.shape
    out = torch.empty((N, h, w, C), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_nchw_to_nhwc_f32(_ptr(x), _ptr(out), N, C, h, w, _stream(x))
    _lib.check(rc, "casmvs_nchw_to_nhwc_f32")
    return out


def costvol(feats, proj_mats, depth_values, num_groups=1, channels_last=False, impl="auto"):
    """Fused plane sweep + aggregation (mvsnet.py:134-172).
    feats (B,V,C,h,w) - or (B,V,h,w,C) with channels_last=True, the faster kernels -,
    proj_mats (B,V-1,3,4), depth_values (B,D,h,w) ->
    (B,C,D,h,w) variance volume (num_groups == 1) or (B,G,D,h,w) group-wise correlation.
    impl (channels_last only): "lds" = source boxes staged in LDS (casmvs_costvol_*_lds_f32), "gather" = the
    texture-path gather kernels (casmvs_costvol_*_nhwc_f32), "auto" = whichever was measured faster for the shape (casmvs_costvol_lds_preferred).
    All three kernel families give bit-identical volumes."""
    feats, proj_mats, depth_values = _dev(feats, "feats"), _dev(proj_mats, "proj_mats"), _dev(depth_values, "depth_values")
    if channels_last:
        B, V, h, w, C = feats.shape
    else:
        B, V, C, h, w = feats.shape
    D = depth_values.shape[1]
    if proj_mats.shape != (B, V - 1, 3, 4) or depth_values.shape != (B, D, h, w):
        raise ValueError(f"costvol: shapes {tuple(feats.shape)} {tuple(proj_mats.shape)} {tuple(depth_values.shape)}")
    lib = _lib.load()
    if not channels_last:
        sfx = ""
    else:
        if impl == "auto":
            impl = "lds" if lib.casmvs_costvol_lds_preferred(C, w, D, V - 1, num_groups) else "gather"
        sfx = "_lds" if impl == "lds" else "_nhwc"
    with torch.cuda.device(feats.device):
        if num_groups == 1:
            out = torch.empty((B, C, D, h, w), dtype=torch.float32, device=feats.device)
            rc = getattr(lib, f"casmvs_costvol_var{sfx}_f32")(_ptr(feats), _ptr(proj_mats), _ptr(depth_values), _ptr(out),
                                                              B, V, C, h, w, D, _stream(feats))
            _lib.check(rc, f"casmvs_costvol_var{sfx}_f32")
        else:
            out = torch.empty((B, num_groups, D, h, w), dtype=torch.float32, device=feats.device)
            rc = getattr(lib, f"casmvs_costvol_gwc{sfx}_f32")(_ptr(feats), _ptr(proj_mats), _ptr(depth_values), _ptr(out),
                                                              B, V, C, num_groups, h, w, D, _stream(feats))
            _lib.check(rc, f"casmvs_costvol_gwc{sfx}_f32")
    return out


def costvol_partial(feats_nhwc, proj_mats, depth_values, view_begin, view_end, num_groups=1, include_ref=True):
    """Partial sums of the view-sharded cost volume: the source views [view_begin, view_end)
    (1-based; view 0 is the reference) of feats_nhwc (B,V,h,w,C).
    num_groups == 1: -> (sum, sq), each (B,C,D,h,w), `include_ref` adds ref / ref^2 (exactly one rank does);
    num_groups  > 1: -> (B,G,D,h,w) group means of (sum of the warped views) * ref.
    Finalise (after the all-reduce) with costvol_finalize."""
    feats, proj_mats, depth_values = _dev(feats_nhwc, "feats"), _dev(proj_mats, "proj_mats"), _dev(depth_values, "depth_values")
    B, V, h, w, C = feats.shape
    D = depth_values.shape[1]
    if proj_mats.shape != (B, V - 1, 3, 4) or depth_values.shape != (B, D, h, w):
        raise ValueError(f"costvol_partial: shapes {tuple(feats.shape)} {tuple(proj_mats.shape)} {tuple(depth_values.shape)}")
    lib = _lib.load()
    # The kernels stage every view of a call in LDS: a range that does not fit (many views x wide channel splits) is
    # processed in chunks whose partial sums are added - they are sums, the order of the additions is the only difference.
    nv = view_end - view_begin
    while nv > 1 and not lib.casmvs_costvol_lds_supported(C, w, D, nv, num_groups):
        nv -= 1
    total = None
    with torch.cuda.device(feats.device):
        for vb in range(view_begin, view_end, nv):
            ve = min(vb + nv, view_end)
            if num_groups == 1:
                part = torch.empty((2, B, C, D, h, w), dtype=torch.float32, device=feats.device)  # one buffer: one all-reduce
                rc = lib.casmvs_costvol_partial_var_f32(_ptr(feats), _ptr(proj_mats), _ptr(depth_values), _ptr(part[0]), _ptr(part[1]),
                                                        B, V, C, h, w, D, vb, ve, int(bool(include_ref) and vb == view_begin), _stream(feats))
                _lib.check(rc, "casmvs_costvol_partial_var_f32")
            else:
                part = torch.empty((B, num_groups, D, h, w), dtype=torch.float32, device=feats.device)
                rc = lib.casmvs_costvol_partial_gwc_f32(_ptr(feats), _ptr(proj_mats), _ptr(depth_values), _ptr(part),
                                                        B, V, C, num_groups, h, w, D, vb, ve, _stream(feats))
                _lib.check(rc, "casmvs_costvol_partial_gwc_f32")
            total = part if total is None else total.add_(part)
    return total


def costvol_finalize(partial, V, num_groups=1):
    """(sum, sq) stacked as (2,B,C,D,h,w) -> variance (B,C,D,h,w) (mvsnet.py:167); or the all-reduced correlation
    (B,G,D,h,w) -> / (V-1) (mvsnet.py:171), in place."""
    partial = _dev(partial, "partial")
    lib = _lib.load()
    with torch.cuda.device(partial.device):
        if num_groups == 1:
            out = partial[0]
            rc = lib.casmvs_costvol_var_finalize_f32(_ptr(partial[0]), _ptr(partial[1]), _ptr(out), out.numel(), V, _stream(partial))
            _lib.check(rc, "casmvs_costvol_var_finalize_f32")
            return out
        rc = lib.casmvs_costvol_gwc_finalize_f32(_ptr(partial), _ptr(partial), partial.numel(), V, _stream(partial))
        _lib.check(rc, "casmvs_costvol_gwc_finalize_f32")
        return partial


def depth_hypotheses(prev_depth, depth_min_b, interval_b, half_range_b, D, h, w):
    """mvsnet.py:213-235 + modules.py:34-49.  Per-sample (B,) float32 device vectors; prev_depth
    (B,hp,wp) or None (coarsest level) -> (B,D,h,w)."""
    interval_b = _dev(interval_b, "interval_b")
    B = interval_b.shape[0]
    if prev_depth is not None:
        prev_depth = _dev(prev_depth, "prev_depth")
        half_range_b = _dev(half_range_b, "half_range_b")
        hp, wp = prev_depth.shape[-2:]
    else:
        depth_min_b = _dev(depth_min_b, "depth_min_b")
        hp = wp = 0
    out = torch.empty((B, D, h, w), dtype=torch.float32, device=interval_b.device)
    with torch.cuda.device(interval_b.device):
        rc = _lib.load().casmvs_depth_hypotheses_f32(_ptr(prev_depth), _ptr(depth_min_b), _ptr(interval_b),
                                                     _ptr(half_range_b), _ptr(out), B, D, h, w, hp, wp,
                                                     _stream(interval_b))
    _lib.check(rc, "casmvs_depth_hypotheses_f32")
    return out


def softmax_regress(cost, depth_values, return_index=False):
    """mvsnet.py:174-193.  cost, depth_values (B,D,h,w) -> depth (B,h,w), confidence (B,h,w)
    [, index (B,h,w) int32]."""
    cost, depth_values = _dev(cost, "cost"), _dev(depth_values, "depth_values")
    if cost.shape != depth_values.shape or cost.dim() != 4:
        raise ValueError(f"softmax_regress: shapes {tuple(cost.shape)} {tuple(depth_values.shape)}")
    B, D, h, w = cost.shape
    depth = torch.empty((B, h, w), dtype=torch.float32, device=cost.device)
    conf = torch.empty((B, h, w), dtype=torch.float32, device=cost.device)
    index = torch.empty((B, h, w), dtype=torch.int32, device=cost.device) if return_index else None
    with torch.cuda.device(cost.device):
        rc = _lib.load().casmvs_softmax_regress_f32(_ptr(cost), _ptr(depth_values), _ptr(depth), _ptr(conf),
                                                    _ptr(index), B, D, h, w, _stream(cost))
    _lib.check(rc, "casmvs_softmax_regress_f32")
    return (depth, conf, index) if return_index else (depth, conf)


def depth_regression(p, depth_values):
    """modules.py:95-104.  p (B,D,h,w); depth_values (B,D,h,w) or (D,) -> (B,h,w)  (casmvs_depth_regression_f32)."""
    p = _dev(p, "p")
    depth_values = _dev(depth_values.to(p.device), "depth_values")
    B, D, h, w = p.shape
    per_plane = depth_values.dim() == 1
    if (per_plane and depth_values.shape[0] != D) or (not per_plane and tuple(depth_values.shape) != (B, D, h, w)):
        raise ValueError(f"depth_regression: shapes {tuple(p.shape)} {tuple(depth_values.shape)}")
    out = torch.empty((B, h, w), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        rc = _lib.load().casmvs_depth_regression_f32(_ptr(p), _ptr(depth_values), _ptr(out), B, D, h, w, int(per_plane), _stream(p))
    _lib.check(rc, "casmvs_depth_regression_f32")
    return out


def conv3d_pack(kind, weight, scale=None, shift=None):
    """Host-side packing of one layer (casmvs_conv3d_pack_f32).  CPU tensors in, CPU tensor out."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    if kind == CONV_T2:
        cin, cout = weight.shape[:2]
    else:
        cout, cin = weight.shape[:2]
    if tuple(weight.shape[2:]) != (3, 3, 3):
        raise ValueError(f"conv3d_pack: kernel {tuple(weight.shape[2:])}, only 3x3x3 is supported")
    lib = _lib.load()
    n = lib.casmvs_conv3d_packed_floats(kind, cin, cout)
    if n == 0:
        raise RuntimeError(f"conv3d_pack: unsupported layer kind={kind} cin={cin} cout={cout}")
    packed = torch.empty(n, dtype=torch.float32)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_conv3d_pack_f32(kind, cin, cout, _ptr(weight), _ptr(sc), _ptr(sh), _ptr(packed))
    _lib.check(rc, "casmvs_conv3d_pack_f32")
    return packed


def costreg_pack(weights, scales=None, shifts=None):
    """Host-side packing of a whole CostRegNet in one call (casmvs_costreg_pack_f32): `weights` = the eleven torch-layout weights of conv0..conv6, conv7,
    conv9, conv11, prob; scales / shifts = per-layer folded ABN (entries may be None).  -> (CPU blob, [float offset of every layer image])."""
    import ctypes
    if len(weights) != 11:
        raise ValueError(f"costreg_pack: {len(weights)} weights, CostRegNet has 11 layers")
    ws = [w.detach().to("cpu", torch.float32).contiguous() for w in weights]
    cin = ws[0].shape[1]
    vec = lambda seq: [None if (seq is None or t is None) else t.detach().to("cpu", torch.float32).contiguous() for t in (seq if seq is not None else [None] * 11)]
    scs, shs = vec(scales), vec(shifts)
    lib = _lib.load()
    offs = (ctypes.c_size_t * 11)()
    n = lib.casmvs_costreg_packed_floats(cin, ctypes.cast(offs, ctypes.c_void_p))
    if n == 0:
        raise RuntimeError(f"costreg_pack: unsupported cin={cin}")
    arr = lambda ts: (ctypes.c_void_p * 11)(*[None if t is None else t.data_ptr() for t in ts])
    packed = torch.empty(n, dtype=torch.float32)
    rc = lib.casmvs_costreg_pack_f32(cin, ctypes.cast(arr(ws), ctypes.c_void_p), ctypes.cast(arr(scs), ctypes.c_void_p), ctypes.cast(arr(shs), ctypes.c_void_p), _ptr(packed))
    _lib.check(rc, "casmvs_costreg_pack_f32")
    return packed, [int(o) for o in offs]


def conv3d_forward(kind, packed, x, cout, skip=None, slope=0.01):
    """One CostRegNet layer on the matrix cores (casmvs_conv3d_forward_f32)."""
    x, packed = _dev(x, "x"), _dev(packed, "packed")
    B, cin, D, H, W = x.shape
    if kind == CONV_S1:
        oshape = (B, cout, D, H, W)
    elif kind == CONV_S2:
        oshape = (B, cout, D // 2, H // 2, W // 2)
    else:
        oshape = (B, cout, 2 * D, 2 * H, 2 * W)
    if skip is not None:
        skip = _dev(skip, "skip")
        if tuple(skip.shape) != oshape:
            raise ValueError(f"conv3d_forward: skip shape {tuple(skip.shape)} != {oshape}")
    out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv3d_forward_f32(kind, _ptr(packed), _ptr(x), _ptr(skip), _ptr(out), B, cin, cout,
                                                   D, H, W, float(slope), _stream(x))
    _lib.check(rc, "casmvs_conv3d_forward_f32")
    return out


def conv0_splitbf16_pack(weight, scale=None, shift=None):
    """Host-side packing of CostRegNet.conv0 for the split-bf16 kernel (casmvs_conv0_splitbf16_pack): weight (8, cin, 3,3,3),
    cin in {8, 16, 32} -> uint8 CPU tensor (the three bf16 slices of every weight as MFMA lane images + scale / shift)."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    n = lib.casmvs_conv0_splitbf16_packed_bytes(cin)
    if cout != 8 or tuple(weight.shape[2:]) != (3, 3, 3) or n == 0:
        raise ValueError(f"conv0_splitbf16_pack: weight {tuple(weight.shape)} (need (8, 8|16|32, 3, 3, 3))")
    packed = torch.empty(n, dtype=torch.uint8)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_conv0_splitbf16_pack(cin, _ptr(weight), _ptr(sc), _ptr(sh), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_conv0_splitbf16_pack")
    return packed


def conv0_splitbf16_forward(packed, x, slope=0.01, terms=0):
    """conv0 on the bf16 matrix cores with float32-grade arithmetic (casmvs_conv0_splitbf16_forward_f32): x (B,cin,D,H,W) ->
    (B,8,D,H,W).  terms: 0 / 6 = six partial products per product, 9 = all nine."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("conv0_splitbf16_forward: `packed` must be the uint8 image on the MI355X")
    B, cin, D, H, W = x.shape
    out = torch.empty((B, 8, D, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv0_splitbf16_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(out), B, cin, D, H, W,
                                                            float(slope), int(terms), _stream(x, f16=True))
    _lib.check(rc, "casmvs_conv0_splitbf16_forward_f32")
    return out


def selftest_mfma_bf16():
    """Lane-semantics probe of v_mfma_f32_16x16x32_bf16 -> (rc, dump (4 regs, 64 lanes), message)."""
    dump = torch.zeros(4 * 64, dtype=torch.float32)
    rc = _lib.load().casmvs_selftest_mfma_bf16(_ptr(dump))
    return rc, dump.view(4, 64), _lib.load().casmvs_last_error().decode()


def conv0_splitf16_pack(weight, scale=None, shift=None):
    """Host-side packing of CostRegNet.conv0 for the split-f16 kernel (casmvs_conv0_splitf16_pack): weight (8, cin, 3,3,3),
    cin in {8, 16, 32} -> uint8 CPU tensor (2^kw w as two float16 slices per weight, MFMA lane images + scale 2^-kw / shift)."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    n = lib.casmvs_conv0_splitf16_packed_bytes(cin)
    if cout != 8 or tuple(weight.shape[2:]) != (3, 3, 3) or n == 0:
        raise ValueError(f"conv0_splitf16_pack: weight {tuple(weight.shape)} (need (8, 8|16|32, 3, 3, 3))")
    packed = torch.empty(n, dtype=torch.uint8)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_conv0_splitf16_pack(cin, _ptr(weight), _ptr(sc), _ptr(sh), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_conv0_splitf16_pack")
    return packed


def conv0_splitf16_forward(packed, x, slope=0.01, terms=0):
    """conv0 on the f16 matrix cores with float32-grade arithmetic (casmvs_conv0_splitf16_forward_f32): x (B,cin,D,H,W) ->
    (B,8,D,H,W).  terms: 0 / 3 = three partial products per product, 4 = all four."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("conv0_splitf16_forward: `packed` must be the uint8 image on the MI355X")
    B, cin, D, H, W = x.shape
    out = torch.empty((B, 8, D, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv0_splitf16_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(out), B, cin, D, H, W,
                                                           float(slope), int(terms), _stream(x, f16=True))
    _lib.check(rc, "casmvs_conv0_splitf16_forward_f32")
    return out


def deconv9_splitf16_pack(weight, scale=None, shift=None):
    """Host-side packing of conv9 (ConvTranspose3d 32 -> 16, weight (32, 16, 3, 3, 3)) for casmvs_deconv9_splitf16_forward_f32 -> uint8 CPU tensor."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    if tuple(weight.shape) != (32, 16, 3, 3, 3):
        raise ValueError(f"deconv9_splitf16_pack: weight {tuple(weight.shape)} (need (32, 16, 3, 3, 3))")
    lib = _lib.load()
    packed = torch.empty(lib.casmvs_deconv9_splitf16_packed_bytes(), dtype=torch.uint8)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_deconv9_splitf16_pack(_ptr(weight), _ptr(sc), _ptr(sh), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_deconv9_splitf16_pack")
    return packed


def deconv9_splitf16_forward(packed, x, skip=None, slope=0.01):
    """conv9 (+ ABN + leaky-relu + skip) on the f16 matrix cores (casmvs_deconv9_splitf16_forward_f32): x (B,32,Di,Hi,Wi), skip (B,16,2Di,2Hi,2Wi) or
    None -> (B,16,2Di,2Hi,2Wi).  Opt-in (added without a GPU run at the end of round 3)."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("deconv9_splitf16_forward: `packed` must be the uint8 image on the MI355X")
    B, cin, Di, Hi, Wi = x.shape
    if cin != 32:
        raise ValueError("deconv9_splitf16_forward: 32 input channels")
    if skip is not None:
        skip = _dev(skip, "skip")
    out = torch.empty((B, 16, 2 * Di, 2 * Hi, 2 * Wi), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_deconv9_splitf16_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(skip), _ptr(out), B, Di, Hi, Wi, float(slope), _stream(x, f16=True))
    _lib.check(rc, "casmvs_deconv9_splitf16_forward_f32")
    return out


def deconv11_splitf16_pack(weight, scale=None, shift=None):
    """Host-side packing of conv11 (ConvTranspose3d 16 -> 8, weight (16, 8, 3, 3, 3)) for casmvs_deconv11_splitf16_forward_f32 -> uint8 CPU tensor."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    if tuple(weight.shape) != (16, 8, 3, 3, 3):
        raise ValueError(f"deconv11_splitf16_pack: weight {tuple(weight.shape)} (need (16, 8, 3, 3, 3))")
    lib = _lib.load()
    packed = torch.empty(lib.casmvs_deconv11_splitf16_packed_bytes(), dtype=torch.uint8)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_deconv11_splitf16_pack(_ptr(weight), _ptr(sc), _ptr(sh), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_deconv11_splitf16_pack")
    return packed


def deconv11_splitf16_forward(packed, x, skip=None, slope=0.01):
    """conv11 (+ ABN + leaky-relu + skip) on the f16 matrix cores (casmvs_deconv11_splitf16_forward_f32): x (B,16,Di,Hi,Wi), skip (B,8,2Di,2Hi,2Wi) or
    None -> (B,8,2Di,2Hi,2Wi)."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("deconv11_splitf16_forward: `packed` must be the uint8 image on the MI355X")
    B, cin, Di, Hi, Wi = x.shape
    if cin != 16:
        raise ValueError("deconv11_splitf16_forward: 16 input channels")
    if skip is not None:
        skip = _dev(skip, "skip")
    out = torch.empty((B, 8, 2 * Di, 2 * Hi, 2 * Wi), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_deconv11_splitf16_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(skip), _ptr(out), B, Di, Hi, Wi, float(slope), _stream(x, f16=True))
    _lib.check(rc, "casmvs_deconv11_splitf16_forward_f32")
    return out


def conv0_zmarch_forward(packed, x, slope=0.01):
    """conv0 in split-f16 arithmetic, input-stationary along z (casmvs_conv0_zmarch_forward_f32, csrc/conv0_zmarch.hip): `packed` is the
    image of conv0_splitf16_pack, x (B,cin,D,H,W) with cin 8 / 16 / 32 -> (B,8,D,H,W).  The regulariser uses it at cin = 16."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("conv0_zmarch_forward: `packed` must be the uint8 image on the MI355X")
    B, cin, D, H, W = x.shape
    out = torch.empty((B, 8, D, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv0_zmarch_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(out), B, cin, D, H, W, float(slope), _stream(x, f16=True))
    _lib.check(rc, "casmvs_conv0_zmarch_forward_f32")
    return out


def selftest_mfma_f16():
    """Lane-semantics probe of v_mfma_f32_16x16x32_f16 -> (rc, dump (4 regs, 64 lanes), message)."""
    dump = torch.zeros(4 * 64, dtype=torch.float32)
    rc = _lib.load().casmvs_selftest_mfma_f16(_ptr(dump))
    return rc, dump.view(4, 64), _lib.load().casmvs_last_error().decode()


def conv_ci_splitf16_pack(weight, scale=None, shift=None):
    """Host-side packing of a stride-1 equal-channel CostRegNet layer (conv2 16 -> 16, conv4 32 -> 32) for the split-f16 channel-inner
    kernel (casmvs_conv_ci_splitf16_pack): weight (c, c, 3,3,3) -> uint8 CPU tensor."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    n = lib.casmvs_conv_ci_splitf16_packed_bytes(cin, cout)
    if tuple(weight.shape[2:]) != (3, 3, 3) or n == 0:
        raise ValueError(f"conv_ci_splitf16_pack: weight {tuple(weight.shape)} (need (c, c, 3, 3, 3) with c in 16, 32, 64)")
    packed = torch.empty(n, dtype=torch.uint8)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_conv_ci_splitf16_pack(cin, cout, _ptr(weight), _ptr(sc), _ptr(sh), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_conv_ci_splitf16_pack")
    return packed


def conv_ci_splitf16_forward(packed, x, cout, slope=0.01):
    """conv2 / conv4 on the f16 matrix cores with float32-grade arithmetic (casmvs_conv_ci_splitf16_forward_f32):
    x (B,cin,D,H,W) -> (B,cout,D,H,W)."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("conv_ci_splitf16_forward: `packed` must be the uint8 image on the MI355X")
    B, cin, D, H, W = x.shape
    out = torch.empty((B, cout, D, H, W), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv_ci_splitf16_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(out), B, cin, cout, D, H, W,
                                                             float(slope), _stream(x, f16=True))
    _lib.check(rc, "casmvs_conv_ci_splitf16_forward_f32")
    return out


def conv_s2_splitf16_pack(weight, scale=None, shift=None):
    """Host-side packing of a stride-2 CostRegNet layer (conv1 8 -> 16, conv3 16 -> 32) for the split-f16 z-marching kernel
    (casmvs_conv_s2_splitf16_pack): weight (cout, cin, 3,3,3) -> uint8 CPU tensor."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    n = lib.casmvs_conv_s2_splitf16_packed_bytes(cin, cout)
    if tuple(weight.shape[2:]) != (3, 3, 3) or n == 0:
        raise ValueError(f"conv_s2_splitf16_pack: weight {tuple(weight.shape)} (need (16, 8, 3, 3, 3) or (32, 16, 3, 3, 3))")
    packed = torch.empty(n, dtype=torch.uint8)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_conv_s2_splitf16_pack(cin, cout, _ptr(weight), _ptr(sc), _ptr(sh), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_conv_s2_splitf16_pack")
    return packed


def conv_s2_splitf16_forward(packed, x, cout, slope=0.01):
    """conv1 / conv3 (Conv3d k3 s2 p1 + ABN) on the f16 matrix cores with float32-grade arithmetic (casmvs_conv_s2_splitf16_forward_f32):
    x (B,cin,D,H,W), W % 4 == 0 -> (B,cout,ceil(D/2),ceil(H/2),ceil(W/2))."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("conv_s2_splitf16_forward: `packed` must be the uint8 image on the MI355X")
    B, cin, D, H, W = x.shape
    out = torch.empty((B, cout, (D - 1) // 2 + 1, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv_s2_splitf16_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(out), B, cin, cout, D, H, W,
                                                             float(slope), _stream(x, f16=True))
    _lib.check(rc, "casmvs_conv_s2_splitf16_forward_f32")
    return out


def costreg_workspace_bytes(B, D, h, w):
    n = _lib.load().casmvs_costreg_workspace_bytes(B, D, h, w)
    if n == 0:
        raise ValueError(f"CostRegNet: D, h, w must be positive multiples of 8 (got {D}, {h}, {w})")
    return n


def costreg_forward(packed_layers, vol, workspace, slope=0.01, layer_events=None):
    """Whole CostRegNet (mvsnet.py:91-104).  packed_layers: 11 device tensors (conv0..conv6, conv7,
    conv9, conv11, prob); vol (B,cin,D,h,w) -> cost (B,D,h,w).  layer_events: optional list of 12
    already-created torch.cuda.Event(enable_timing=True) recorded around the 11 launches."""
    vol = _dev(vol, "vol")
    B, cin, D, h, w = vol.shape
    if len(packed_layers) != 11:
        raise ValueError("costreg_forward: need 11 packed layers")
    arr = (ctypes.c_void_p * 11)(*[p.data_ptr() for p in packed_layers])
    cost = torch.empty((B, D, h, w), dtype=torch.float32, device=vol.device)
    need = costreg_workspace_bytes(B, D, h, w)
    if workspace.numel() * workspace.element_size() < need:
        raise ValueError("costreg_forward: workspace too small")
    ev = None
    if layer_events is not None:
        if len(layer_events) != 12:
            raise ValueError("costreg_forward: need 12 events")
        ev = (ctypes.c_void_p * 12)(*[e.cuda_event for e in layer_events])
    with torch.cuda.device(vol.device):
        rc = _lib.load().casmvs_costreg_forward_f32(arr, _ptr(vol), _ptr(cost), ctypes.c_void_p(workspace.data_ptr()),
                                                    B, cin, D, h, w, float(slope), ev, _stream(vol))
    _lib.check(rc, "casmvs_costreg_forward_f32")
    return cost


CONV0_F32, CONV0_SPLIT_BF16, CONV0_SPLIT_F16 = 0, 1, 2   # casmvs.h: CASMVS_CONV0_*


def costreg_regress(packed_layers, vol, depth_values, workspace, slope=0.01, layer_events=None, return_index=False, conv0_split=None,
                    conv0_arith=CONV0_F32, conv2_split=None, conv4_split=None, conv6_split=None, conv9_split=None, conv11_split=None, conv1_split=None,
                    conv3_split=None):
    """CostRegNet + softmax / depth regression / confidence in one library call (mvsnet.py:91-104 + :174-193): the `prob`
    head walks the depth axis and, when the whole depth range is one chunk, runs the regression on the cost values it has
    just produced (casmvs_costreg_regress_f32).  -> cost (B,D,h,w), depth (B,h,w), confidence (B,h,w) [, index int32].
    conv0_arith: CONV0_F32 (conv0 on the float32 MFMA kernel), CONV0_SPLIT_BF16 / CONV0_SPLIT_F16 with conv0_split = the device
    image of conv0_splitbf16_pack / conv0_splitf16_pack (conv0 on the bf16 / f16 matrix cores with float32-grade arithmetic).
    conv2_split / conv4_split / conv6_split: device images of conv_ci_splitf16_pack (those layers on the f16 matrix cores) or None;
    conv9_split / conv11_split: device images of deconv9_splitf16_pack / deconv11_splitf16_pack (the transposed layers there) or None;
    conv1_split / conv3_split: device images of conv_s2_splitf16_pack (the stride-2 layers there) or None."""
    vol, depth_values = _dev(vol, "vol"), _dev(depth_values, "depth_values")
    B, cin, D, h, w = vol.shape
    if tuple(depth_values.shape) != (B, D, h, w):
        raise ValueError(f"costreg_regress: shapes {tuple(vol.shape)} {tuple(depth_values.shape)}")
    if len(packed_layers) != 11:
        raise ValueError("costreg_regress: need 11 packed layers")
    arr = (ctypes.c_void_p * 11)(*[p.data_ptr() for p in packed_layers])
    dev = vol.device
    cost = torch.empty((B, D, h, w), dtype=torch.float32, device=dev)
    depth = torch.empty((B, h, w), dtype=torch.float32, device=dev)
    conf = torch.empty((B, h, w), dtype=torch.float32, device=dev)
    index = torch.empty((B, h, w), dtype=torch.int32, device=dev) if return_index else None
    if workspace.numel() * workspace.element_size() < costreg_workspace_bytes(B, D, h, w):
        raise ValueError("costreg_regress: workspace too small")
    ev = None
    if layer_events is not None:
        if len(layer_events) != 12:
            raise ValueError("costreg_regress: need 12 events")
        ev = (ctypes.c_void_p * 12)(*[e.cuda_event for e in layer_events])
    split = None
    images = (conv0_split, conv2_split, conv4_split, conv6_split, conv9_split, conv11_split, conv1_split, conv3_split)
    if any(t is not None for t in images):
        split = (ctypes.c_void_p * 8)(*[None if t is None else t.data_ptr() for t in images])
    with torch.cuda.device(dev):
        rc = _lib.load().casmvs_costreg_regress_f32(arr, split, int(conv0_arith), _ptr(vol), _ptr(depth_values), _ptr(cost), _ptr(depth), _ptr(conf),
                                                    _ptr(index), ctypes.c_void_p(workspace.data_ptr()), B, cin, D, h, w,
                                                    float(slope), ev, _stream(vol, f16=split is not None))
    _lib.check(rc, "casmvs_costreg_regress_f32")
    return (cost, depth, conf, index) if return_index else (cost, depth, conf)


def prob_regress(packed, x, depth_values=None, slope=1.0, zchunk=0, return_index=False):
    """The `prob` head on its own (casmvs_prob_regress_f32): x (B,8,D,h,w) -> cost (B,D,h,w); with depth_values (B,D,h,w)
    also depth, confidence (B,h,w) [, index].  zchunk: output planes per workgroup along D (0 = library's choice)."""
    x, packed = _dev(x, "x"), _dev(packed, "packed")
    B, cin, D, h, w = x.shape
    dev = x.device
    cost = torch.empty((B, D, h, w), dtype=torch.float32, device=dev)
    depth = conf = index = None
    if depth_values is not None:
        depth_values = _dev(depth_values, "depth_values")
        if tuple(depth_values.shape) != (B, D, h, w):
            raise ValueError(f"prob_regress: shapes {tuple(x.shape)} {tuple(depth_values.shape)}")
        depth = torch.empty((B, h, w), dtype=torch.float32, device=dev)
        conf = torch.empty((B, h, w), dtype=torch.float32, device=dev)
        index = torch.empty((B, h, w), dtype=torch.int32, device=dev) if return_index else None
    with torch.cuda.device(dev):
        rc = _lib.load().casmvs_prob_regress_f32(_ptr(packed), _ptr(x), _ptr(depth_values), _ptr(cost), _ptr(depth), _ptr(conf),
                                                 _ptr(index), B, cin, D, h, w, float(slope), int(zchunk), _stream(x))
    _lib.check(rc, "casmvs_prob_regress_f32")
    if depth_values is None:
        return cost
    return (cost, depth, conf, index) if return_index else (cost, depth, conf)


def conv11_prob_zfused(deconv11_packed, prob_packed, x, skip, depth_values, slope=0.01, prob_slope=1.0, return_index=False):
    """CostRegNet's tail as one depth-walking kernel (casmvs_conv11_prob_zfused_f32): conv11 + ABN + leaky-relu + skip, `prob`, softmax / regression /
    confidence.  deconv11_packed: device uint8 image of deconv11_splitf16_pack; prob_packed: device image of conv3d_pack(CONV_S1, prob weight (1,8,3,3,3), None, bias);
    x (B,16,Di,Hi,Wi), skip (B,8,2Di,2Hi,2Wi), depth_values (B,2Di,2Hi,2Wi) -> cost (B,2Di,2Hi,2Wi), depth, confidence (B,2Hi,2Wi) [, index]."""
    x, skip, depth_values, prob_packed = _dev(x, "x"), _dev(skip, "skip"), _dev(depth_values, "depth_values"), _dev(prob_packed, "prob_packed")
    if not deconv11_packed.is_cuda or deconv11_packed.dtype != torch.uint8:
        raise RuntimeError("conv11_prob_zfused: `deconv11_packed` must be the uint8 image on the MI355X")
    B, cin, Di, Hi, Wi = x.shape
    D, h, w = 2 * Di, 2 * Hi, 2 * Wi
    if cin != 16 or tuple(skip.shape) != (B, 8, D, h, w) or tuple(depth_values.shape) != (B, D, h, w):
        raise ValueError(f"conv11_prob_zfused: shapes {tuple(x.shape)} {tuple(skip.shape)} {tuple(depth_values.shape)}")
    dev = x.device
    cost = torch.empty((B, D, h, w), dtype=torch.float32, device=dev)
    depth = torch.empty((B, h, w), dtype=torch.float32, device=dev)
    conf = torch.empty((B, h, w), dtype=torch.float32, device=dev)
    index = torch.empty((B, h, w), dtype=torch.int32, device=dev) if return_index else None
    with torch.cuda.device(dev):
        rc = _lib.load().casmvs_conv11_prob_zfused_f32(ctypes.c_void_p(deconv11_packed.data_ptr()), _ptr(prob_packed), _ptr(x), _ptr(skip), _ptr(depth_values),
                                                       _ptr(cost), _ptr(depth), _ptr(conf), _ptr(index), B, Di, Hi, Wi, float(slope), float(prob_slope),
                                                       _stream(x, f16=True))
    _lib.check(rc, "casmvs_conv11_prob_zfused_f32")
    return (cost, depth, conf, index) if return_index else (cost, depth, conf)


_CONV2D_KSIZE = {CONV2D_K3: 3, CONV2D_K5S2: 5, CONV2D_K1: 1, CONV2D_K1_UP: 1}


def conv2d_pack(kind, weight, scale=None, shift=None):
    """Host-side packing of one FeatureNet layer (casmvs_conv2d_pack_f32).  CPU tensors in / out."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    cout, cin = weight.shape[:2]
    k = _CONV2D_KSIZE.get(kind)
    if k is None or tuple(weight.shape[2:]) != (k, k):
        raise ValueError(f"conv2d_pack: kind {kind} with kernel {tuple(weight.shape[2:])}")
    lib = _lib.load()
    n = lib.casmvs_conv2d_packed_floats(kind, cin, cout)
    if n == 0:
        raise RuntimeError(f"conv2d_pack: unsupported layer kind={kind} cin={cin} cout={cout}")
    packed = torch.empty(n, dtype=torch.float32)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_conv2d_pack_f32(kind, cin, cout, _ptr(weight), _ptr(sc), _ptr(sh), _ptr(packed))
    _lib.check(rc, "casmvs_conv2d_pack_f32")
    return packed


def conv2d_forward(kind, packed, x, cout, up=None, slope=0.01):
    """One FeatureNet layer on the matrix cores (casmvs_conv2d_forward_f32).  `up`: the coarser FPN
    level (N, cout, H/2, W/2) whose bilinear x2 upsampling is added (CONV2D_K1_UP only)."""
    x, packed = _dev(x, "x"), _dev(packed, "packed")
    N, cin, H, W = x.shape
    oshape = (N, cout, H // 2, W // 2) if kind == CONV2D_K5S2 else (N, cout, H, W)
    if up is not None:
        up = _dev(up, "up")
        if tuple(up.shape) != (N, cout, H // 2, W // 2):
            raise ValueError(f"conv2d_forward: up shape {tuple(up.shape)} != {(N, cout, H // 2, W // 2)}")
    out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv2d_forward_f32(kind, _ptr(packed), _ptr(x), _ptr(up), _ptr(out), N, cin, cout,
                                                   H, W, float(slope), _stream(x))
    _lib.check(rc, "casmvs_conv2d_forward_f32")
    return out


def featurenet_workspace_bytes(N, H, W):
    n = _lib.load().casmvs_featurenet_workspace_bytes(N, H, W)
    if n == 0:
        raise ValueError(f"FeatureNet: H, W must be positive multiples of 4 (got {H}, {W})")
    return n


def fpn_tail0(packed40, bias9, conv0, feat1_sum, channels_last_copy=False):
    """feat0 = smooth0(lat0(conv0) + upsample2x(feat1_sum)) as one kernel (casmvs_fpn_tail0_f32); packed40 / bias9 from
    mvsnet.compose_fpn_tail + conv2d_pack.  conv0 (N,8,H,W), feat1_sum (N,32,H/2,W/2) -> (N,8,H,W) [, (N,H,W,8)]."""
    conv0, feat1_sum, packed40, bias9 = _dev(conv0, "conv0"), _dev(feat1_sum, "feat1_sum"), _dev(packed40, "packed40"), _dev(bias9, "bias9")
    N, c, H, W = conv0.shape
    if c != 8 or tuple(feat1_sum.shape) != (N, 32, H // 2, W // 2) or tuple(bias9.shape) != (3, 3, 8):
        raise ValueError(f"fpn_tail0: shapes {tuple(conv0.shape)} {tuple(feat1_sum.shape)} {tuple(bias9.shape)}")
    out = torch.empty((N, 8, H, W), dtype=torch.float32, device=conv0.device)
    out2 = torch.empty((N, H, W, 8), dtype=torch.float32, device=conv0.device) if channels_last_copy else None
    with torch.cuda.device(conv0.device):
        rc = _lib.load().casmvs_fpn_tail0_f32(_ptr(packed40), _ptr(bias9), _ptr(conv0), _ptr(feat1_sum), _ptr(out), _ptr(out2), N, H, W, _stream(conv0))
    _lib.check(rc, "casmvs_fpn_tail0_f32")
    return (out, out2) if channels_last_copy else out


def conv2d_ci_splitf16_pack(weight, scale=None, shift=None):
    """Host-side packing of a 3x3 stride-1 FeatureNet layer (16 -> 16, 32 -> 32 or 32 -> 16) for the split-f16 kernel
    (casmvs_conv2d_ci_splitf16_pack): weight (cout, cin, 3, 3) -> uint8 CPU tensor."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    n = lib.casmvs_conv2d_ci_splitf16_packed_bytes(cin, cout)
    if tuple(weight.shape[2:]) != (3, 3) or n == 0:
        raise ValueError(f"conv2d_ci_splitf16_pack: weight {tuple(weight.shape)} (need (16, 16, 3, 3), (32, 32, 3, 3) or (16, 32, 3, 3))")
    packed = torch.empty(n, dtype=torch.uint8)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_conv2d_ci_splitf16_pack(cin, cout, _ptr(weight), _ptr(sc), _ptr(sh), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_conv2d_ci_splitf16_pack")
    return packed


def conv2d_ci_splitf16_forward(packed, x, cout=None, slope=0.01, channels_last_copy=False):
    """conv1.1 / conv1.2 / conv2.1 / conv2.2 / smooth1 of FeatureNet on the f16 matrix cores (casmvs_conv2d_ci_splitf16_forward_f32):
    x (N,cin,H,W) -> (N,cout,H,W) [, (N,H,W,cout)]."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("conv2d_ci_splitf16_forward: `packed` must be the uint8 image on the MI355X")
    N, cin, H, W = x.shape
    cout = cin if cout is None else int(cout)
    out = torch.empty((N, cout, H, W), dtype=torch.float32, device=x.device)
    out2 = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device) if channels_last_copy else None
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv2d_ci_splitf16_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(out), _ptr(out2), N, cin, cout, H, W,
                                                               float(slope), _stream(x, f16=True))
    _lib.check(rc, "casmvs_conv2d_ci_splitf16_forward_f32")
    return (out, out2) if channels_last_copy else out


def conv2d_k5s2_splitf16_pack(weight, scale=None, shift=None):
    """Host-side packing of a 5x5 stride-2 FeatureNet layer (conv1.0: 8 -> 16, conv2.0: 16 -> 32) for the split-f16 kernel
    (casmvs_conv2d_k5s2_splitf16_pack): weight (cout, cin, 5, 5) -> uint8 CPU tensor."""
    weight = weight.detach().to("cpu", torch.float32).contiguous()
    cout, cin = weight.shape[:2]
    lib = _lib.load()
    n = lib.casmvs_conv2d_k5s2_splitf16_packed_bytes(cin, cout)
    if tuple(weight.shape[2:]) != (5, 5) or n == 0:
        raise ValueError(f"conv2d_k5s2_splitf16_pack: weight {tuple(weight.shape)} (need (16, 8, 5, 5) or (32, 16, 5, 5))")
    packed = torch.empty(n, dtype=torch.uint8)
    sc = None if scale is None else scale.detach().to("cpu", torch.float32).contiguous()
    sh = None if shift is None else shift.detach().to("cpu", torch.float32).contiguous()
    rc = lib.casmvs_conv2d_k5s2_splitf16_pack(cin, cout, _ptr(weight), _ptr(sc), _ptr(sh), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_conv2d_k5s2_splitf16_pack")
    return packed


def conv2d_k5s2_splitf16_forward(packed, x, cout, slope=0.01):
    """conv1.0 / conv2.0 of FeatureNet on the f16 matrix cores (casmvs_conv2d_k5s2_splitf16_forward_f32): x (N,cin,H,W), H even, W % 4 == 0 -> (N,cout,H/2,W/2)."""
    x = _dev(x, "x")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("conv2d_k5s2_splitf16_forward: `packed` must be the uint8 image on the MI355X")
    N, cin, H, W = x.shape
    out = torch.empty((N, cout, H // 2, W // 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().casmvs_conv2d_k5s2_splitf16_forward_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(x), _ptr(out), N, cin, cout, H, W, float(slope),
                                                                 _stream(x, f16=True))
    _lib.check(rc, "casmvs_conv2d_k5s2_splitf16_forward_f32")
    return out


def fnet_conv0_mm_pack(w0, scale0, shift0, w1, scale1, shift1):
    """Host-side packing of FeatureNet.conv0's two layers for the fused kernel (casmvs_fnet_conv0_mm_pack): w0 (8, 3, 3, 3), w1 (8, 8, 3, 3), the layers'
    folded eval-mode ABN scale / shift (8) or None -> uint8 CPU tensor."""
    w0, w1 = w0.detach().to("cpu", torch.float32).contiguous(), w1.detach().to("cpu", torch.float32).contiguous()
    if tuple(w0.shape) != (8, 3, 3, 3) or tuple(w1.shape) != (8, 8, 3, 3):
        raise ValueError(f"fnet_conv0_mm_pack: weights {tuple(w0.shape)} {tuple(w1.shape)} (need (8, 3, 3, 3) and (8, 8, 3, 3))")
    vec = [None if v is None else v.detach().to("cpu", torch.float32).contiguous() for v in (scale0, shift0, scale1, shift1)]
    for v in vec:
        if v is not None and tuple(v.shape) != (8,):
            raise ValueError("fnet_conv0_mm_pack: scale / shift must hold 8 values")
    lib = _lib.load()
    packed = torch.empty(lib.casmvs_fnet_conv0_mm_packed_bytes(), dtype=torch.uint8)
    rc = lib.casmvs_fnet_conv0_mm_pack(_ptr(w0), _ptr(vec[0]), _ptr(vec[1]), _ptr(w1), _ptr(vec[2]), _ptr(vec[3]), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_fnet_conv0_mm_pack")
    return packed


def fnet_conv0_mm(packed, imgs, slope=0.01):
    """FeatureNet.conv0 (ConvBnReLU 3 -> 8 -> 8, mvsnet.py:14-16) as one kernel on the f16 matrix cores (casmvs_fnet_conv0_mm_f32); packed: device uint8
    image of fnet_conv0_mm_pack.  imgs (N, 3, H, W) -> (N, 8, H, W)."""
    imgs = _dev(imgs, "imgs")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("fnet_conv0_mm: `packed` must be the uint8 image on the MI355X")
    N, c, H, W = imgs.shape
    if c != 3 or not _lib.load().casmvs_fnet_conv0_mm_supported(W):
        raise ValueError(f"fnet_conv0_mm: imgs {tuple(imgs.shape)} (3 channels, W even)")
    out = torch.empty((N, 8, H, W), dtype=torch.float32, device=imgs.device)
    with torch.cuda.device(imgs.device):
        rc = _lib.load().casmvs_fnet_conv0_mm_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(imgs), _ptr(out), N, H, W, float(slope), _stream(imgs, f16=True))
    _lib.check(rc, "casmvs_fnet_conv0_mm_f32")
    return out


def fpn_tail0_splitf16_pack(weight40):
    """Host-side packing of the composed 40-channel 3x3 tail (mvsnet.compose_fpn_tail) for the split-f16 kernel
    (casmvs_fpn_tail0_splitf16_pack): weight40 (8, 40, 3, 3) -> uint8 CPU tensor."""
    weight40 = weight40.detach().to("cpu", torch.float32).contiguous()
    if tuple(weight40.shape) != (8, 40, 3, 3):
        raise ValueError(f"fpn_tail0_splitf16_pack: weight {tuple(weight40.shape)} (need (8, 40, 3, 3))")
    lib = _lib.load()
    packed = torch.empty(lib.casmvs_fpn_tail0_splitf16_packed_bytes(), dtype=torch.uint8)
    rc = lib.casmvs_fpn_tail0_splitf16_pack(_ptr(weight40), ctypes.c_void_p(packed.data_ptr()))
    _lib.check(rc, "casmvs_fpn_tail0_splitf16_pack")
    return packed


def fpn_tail0_splitf16(packed, bias9, conv0, feat1_sum, channels_last_copy=False):
    """fpn_tail0 on the f16 matrix cores with float32-grade arithmetic (casmvs_fpn_tail0_splitf16_f32); packed: device uint8 image of
    fpn_tail0_splitf16_pack."""
    conv0, feat1_sum, bias9 = _dev(conv0, "conv0"), _dev(feat1_sum, "feat1_sum"), _dev(bias9, "bias9")
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("fpn_tail0_splitf16: `packed` must be the uint8 image on the MI355X")
    N, c, H, W = conv0.shape
    if c != 8 or tuple(feat1_sum.shape) != (N, 32, H // 2, W // 2) or tuple(bias9.shape) != (3, 3, 8):
        raise ValueError(f"fpn_tail0_splitf16: shapes {tuple(conv0.shape)} {tuple(feat1_sum.shape)} {tuple(bias9.shape)}")
    out = torch.empty((N, 8, H, W), dtype=torch.float32, device=conv0.device)
    out2 = torch.empty((N, H, W, 8), dtype=torch.float32, device=conv0.device) if channels_last_copy else None
    with torch.cuda.device(conv0.device):
        rc = _lib.load().casmvs_fpn_tail0_splitf16_f32(ctypes.c_void_p(packed.data_ptr()), _ptr(bias9), _ptr(conv0), _ptr(feat1_sum), _ptr(out), _ptr(out2),
                                                       N, H, W, _stream(conv0, f16=True))
    _lib.check(rc, "casmvs_fpn_tail0_splitf16_f32")
    return (out, out2) if channels_last_copy else out


def featurenet_forward(packed_layers, imgs, workspace, slope=0.01, layer_events=None, channels_last_copies=False, fused0=None, fused0_splitf16=False,
                       ci_layers=None, nchw_outputs=True):
    """Whole FeatureNet (mvsnet.py:40-57).  packed_layers: 13 device tensors (conv0.0 .. conv2.2,
    toplayer, lat1, lat0, smooth1, smooth0); imgs (N,3,H,W) -> feat0 (N,8,H,W), feat1 (N,16,H/2,W/2),
    feat2 (N,32,H/4,W/4).  layer_events: optional 14 recorded torch.cuda.Event.
    channels_last_copies: also return the three maps pixel-major (N,h,w,C) (written by the same
    kernels) -> (feat0, feat1, feat2, (nhwc0, nhwc1, nhwc2)).  fused0: (packed40, bias9) device tensors - the
    full-resolution tail as one kernel (casmvs_featurenet_forward_fused_f32); fused0_splitf16: packed40 is the uint8 image of
    fpn_tail0_splitf16_pack (the tail on the f16 matrix cores) instead of the float32 conv2d_pack image.  ci_layers (with fused0 only):
    8 device images - conv2d_ci_splitf16_pack's for conv1.1, conv1.2, conv2.1, conv2.2, smooth1, conv2d_k5s2_splitf16_pack's for conv1.0, conv2.0,
    fnet_conv0_mm_pack's for conv0 (conv0.0 + conv0.1 as one kernel) (entries may be None) - those layers on the f16 cores.
    nchw_outputs=False (with channels_last_copies; the engine's own call): feat0 / feat1 are not stored - nothing downstream reads their (N,C,h,w) layout -
    and come back as None."""
    imgs = _dev(imgs, "imgs")
    N, c, H, W = imgs.shape
    if c != 3 or len(packed_layers) != 13:
        raise ValueError("featurenet_forward: need (N,3,H,W) images and 13 packed layers")
    need = featurenet_workspace_bytes(N, H, W)
    if workspace.numel() * workspace.element_size() < need:
        raise ValueError("featurenet_forward: workspace too small")
    arr = (ctypes.c_void_p * 13)(*[p.data_ptr() for p in packed_layers])
    dev = imgs.device
    if not nchw_outputs and not channels_last_copies:
        raise ValueError("featurenet_forward: nchw_outputs=False needs channels_last_copies=True")
    feat0 = torch.empty((N, 8, H, W), dtype=torch.float32, device=dev) if nchw_outputs else None
    feat1 = torch.empty((N, 16, H // 2, W // 2), dtype=torch.float32, device=dev) if nchw_outputs else None
    feat2 = torch.empty((N, 32, H // 4, W // 4), dtype=torch.float32, device=dev)
    cl = (None, None, None)
    if channels_last_copies:
        cl = (torch.empty((N, H, W, 8), dtype=torch.float32, device=dev),
              torch.empty((N, H // 2, W // 2, 16), dtype=torch.float32, device=dev),
              torch.empty((N, H // 4, W // 4, 32), dtype=torch.float32, device=dev))
    ev = None
    if layer_events is not None:
        if len(layer_events) != 14:
            raise ValueError("featurenet_forward: need 14 events")
        ev = (ctypes.c_void_p * 14)(*[e.cuda_event for e in layer_events])
    with torch.cuda.device(dev):
        if fused0 is not None:   # (packed 40-channel tail, bias classes): lat0 + upsample-add + smooth0 in one kernel
            ci = None
            if ci_layers is not None:
                if len(ci_layers) != 8:
                    raise ValueError("featurenet_forward: ci_layers needs 8 entries (conv1.1, conv1.2, conv2.1, conv2.2, smooth1, conv1.0, conv2.0, conv0)")
                ci = (ctypes.c_void_p * 8)(*[None if t is None else t.data_ptr() for t in ci_layers])
            rc = _lib.load().casmvs_featurenet_forward_fused_f32(arr, ctypes.c_void_p(fused0[0].data_ptr()), 1 if fused0_splitf16 else 0, _ptr(fused0[1]),
                                                                 ci, _ptr(imgs), _ptr(feat0), _ptr(feat1),
                                                                 _ptr(feat2), _ptr(cl[0]), _ptr(cl[1]), _ptr(cl[2]),
                                                                 ctypes.c_void_p(workspace.data_ptr()), N, H, W, float(slope), ev,
                                                                 _stream(imgs, f16=bool(fused0_splitf16) or ci is not None))
        else:
            rc = _lib.load().casmvs_featurenet_forward_f32(arr, _ptr(imgs), _ptr(feat0), _ptr(feat1), _ptr(feat2),
                                                           _ptr(cl[0]), _ptr(cl[1]), _ptr(cl[2]),
                                                           ctypes.c_void_p(workspace.data_ptr()), N, H, W, float(slope),
                                                           ev, _stream(imgs))
    _lib.check(rc, "casmvs_featurenet_forward_f32")
    return (feat0, feat1, feat2, cl) if channels_last_copies else (feat0, feat1, feat2)


def selftest_mfma():
    """MFMA lane-mapping probe; returns the raw (4 variants, 4 regs, 64 lanes) dump."""
    dump = torch.zeros(16 * 64, dtype=torch.float32)
    rc = _lib.load().casmvs_selftest_mfma(_ptr(dump))
    return rc, dump.view(4, 4, 64), _lib.load().casmvs_last_error().decode()


def selftest_mfma_rate(shape=1, blocks=2048, iters=4096):
    """Measured TFLOP/s of back-to-back fp32 MFMAs (shape 0: 4x4x1_16b, 1: 16x16x4, 2: 32x32x2,
    3: 16x16x1_4b) - the ceilings the conv kernels are judged against."""
    out = ctypes.c_float(0.0)
    rc = _lib.load().casmvs_selftest_mfma_rate(shape, blocks, iters, ctypes.byref(out))
    _lib.check(rc, "casmvs_selftest_mfma_rate")
    return out.value
