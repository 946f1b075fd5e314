"""Training support, op by op: torch.autograd.Function wrappers whose forward AND backward are HIP
kernels of libcasmvs_hip.so.  `CascadeMVSNet.forward` itself stays the inference engine (eval-mode ABN folded into
the conv epilogues); these are the differentiable forms of the two ops of the hot path whose gradients are not plain
convolution gradients:

  homo_warp(src_feat, proj_mat, depth_values)      models/modules.py:52-92.  Gradient with respect to src_feat only:
      the sampling grid depends on proj_mat and on depth hypotheses that the reference detaches (mvsnet.py:231).
  softmax_depth_regression(cost, depth_values)     models/mvsnet.py:175-177 + modules.py:95-104: depth = sum_k
      softmax(cost)_k d_k, gradient with respect to the cost volume (the hypotheses are detached; the confidence is
      computed under torch.no_grad() in the reference, mvsnet.py:179-193, and is returned without a graph here).

A variance / correlation cost volume built from `homo_warp` outputs with torch's elementwise ops (the reference's own
training-mode code, mvsnet.py:150-153) is therefore differentiable end to end down to the feature maps.
"""
import ctypes

import torch

from . import _lib, ops


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream(t):
    from . import streams
    return streams.launch_stream(t)


class _HomoWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src_feat, proj_mat, depth_values):
        ctx.save_for_backward(proj_mat, depth_values)
        ctx.src_shape = tuple(src_feat.shape)
        return ops.homo_warp(src_feat.detach(), proj_mat.detach(), depth_values.detach())

    @staticmethod
    def backward(ctx, grad_out):
        proj_mat, depth_values = ctx.saved_tensors
        B, C, H, W = ctx.src_shape
        D = depth_values.shape[1]
        grad_out = grad_out.contiguous().float()
        proj_mat, depth_values = proj_mat.contiguous().float(), depth_values.contiguous().float()
        grad_src = torch.empty((B, C, H, W), dtype=torch.float32, device=grad_out.device)
        # caller-owned scratch: the 64-bit fixed-point map of the order-independent scatter (same bits run to run) + the channels' largest magnitudes
        ws = torch.empty(_lib.load().casmvs_homo_warp_backward_workspace_bytes(B, C, D, H, W) // 8 + 1, dtype=torch.int64, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            rc = _lib.load().casmvs_homo_warp_backward_f32(_ptr(grad_out), _ptr(proj_mat), _ptr(depth_values), _ptr(grad_src), _ptr(ws),
                                                           B, C, H, W, D, _stream(grad_out))
        _lib.check(rc, "casmvs_homo_warp_backward_f32")
        return grad_src, None, None


class _SoftmaxDepthRegression(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, depth_values):
        cost, depth_values = cost.detach().contiguous().float(), depth_values.detach().contiguous().float()
        ctx.save_for_backward(cost, depth_values)
        depth, confidence = ops.softmax_regress(cost, depth_values)
        ctx.mark_non_differentiable(confidence)
        return depth, confidence

    @staticmethod
    def backward(ctx, grad_depth, _grad_confidence):
        cost, depth_values = ctx.saved_tensors
        B, D, h, w = cost.shape
        grad_depth = grad_depth.contiguous().float()
        grad_cost = torch.empty_like(cost)
        with torch.cuda.device(cost.device):
            rc = _lib.load().casmvs_softmax_regress_backward_f32(_ptr(cost), _ptr(depth_values), _ptr(grad_depth), _ptr(grad_cost),
                                                                 B, D, h, w, _stream(cost))
        _lib.check(rc, "casmvs_softmax_regress_backward_f32")
        return grad_cost, None


def homo_warp(src_feat, proj_mat, depth_values):
    """Differentiable models/modules.py:52-92: (B,C,H,W), (B,3,4), (B,D,H,W) -> (B,C,D,H,W); d/d src_feat by HIP scatter-add."""
    if not src_feat.is_cuda:
        raise RuntimeError("casmvsnet_pl_amd.autograd.homo_warp runs on the MI355X only; there is no CPU fallback")
    return _HomoWarp.apply(src_feat, proj_mat, depth_values)


def softmax_depth_regression(cost, depth_values):
    """Differentiable models/mvsnet.py:175-193: cost, depth_values (B,D,h,w) -> depth (B,h,w) [graph], confidence (B,h,w) [no graph]."""
    if not cost.is_cuda:
        raise RuntimeError("casmvsnet_pl_amd.autograd.softmax_depth_regression runs on the MI355X only; there is no CPU fallback")
    return _SoftmaxDepthRegression.apply(cost, depth_values)
