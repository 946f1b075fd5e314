// Backward kernels of the plane sweep and of the softmax depth regression (training support op by op).
//
// Reference semantics: the autograd graph of models/modules.py:52-92 (homo_warp: F.grid_sample's gradient with
// respect to its INPUT - the sampling grid depends on the detached depth hypotheses only, mvsnet.py:231, so no
// gradient flows into it) and of models/mvsnet.py:175-177 + modules.py:95-104 (softmax over depth, then
// depth = sum_k p_k d_k; the confidence is computed under torch.no_grad(), mvsnet.py:179-193).
#include "common.h"
#include "plane_sweep.h"

namespace {

using namespace casmvs_dev;

constexpr int kThreads = 256;

// grad_src[b, c, tap] += grad_out[b, c, d, y, x] * w_tap: the transpose of the forward gather is a scatter-add.
// One thread per (reference pixel, plane), lanes along the row: the taps (same arithmetic as the forward, plane_sweep.h)
// are computed once and reused for every channel; fp32 hardware atomics (global_atomic_add_f32) - like ATen's own
// grid_sampler backward the result depends on the order of the atomic adds in the last bits.
__global__ __launch_bounds__(kThreads) void homo_warp_bwd_kernel(const float *__restrict__ grad_out, const float *__restrict__ proj,
                                                                const float *__restrict__ depth, float *__restrict__ grad_src,
                                                                int C, int H, int W, int D) {
  const int b = blockIdx.z, d = blockIdx.y;
  const int hw = H * W;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const int y = p / W, x = p - y * W;
  const float dv = depth[((size_t)b * D + d) * hw + p];
  const Taps t = plane_sweep_taps(proj + (size_t)b * 12, (float)x, (float)y, dv, W, H);
  if (!taps_live(t)) return;
  const float *go = grad_out + ((size_t)b * C * D + d) * hw + p;
  float *gs = grad_src + (size_t)b * C * hw;
  const int on = t.yn * W + t.xl, os = t.ys * W + t.xl;
  for (int c = 0; c < C; ++c) {
    const float g = go[(size_t)c * D * hw];
    float *gc = gs + (size_t)c * hw;
    if (t.w_nl != 0.0f) unsafeAtomicAdd(gc + on, g * t.w_nl);
    if (t.w_nr != 0.0f) unsafeAtomicAdd(gc + on + 1, g * t.w_nr);
    if (t.w_sl != 0.0f) unsafeAtomicAdd(gc + os, g * t.w_sl);
    if (t.w_sr != 0.0f) unsafeAtomicAdd(gc + os + 1, g * t.w_sr);
  }
}

// depth = sum_k softmax(cost)_k d_k  =>  d depth / d cost_k = p_k (d_k - depth).  One thread per pixel.
__global__ __launch_bounds__(kThreads) void softmax_regress_bwd_kernel(const float *__restrict__ cost, const float *__restrict__ dvals,
                                                                      const float *__restrict__ grad_depth, float *__restrict__ grad_cost,
                                                                      int D, int hw) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const float *cp = cost + (size_t)b * D * hw + p;
  const float *dp = dvals + (size_t)b * D * hw + p;
  float mx = -INFINITY;
  for (int k = 0; k < D; ++k) mx = fmaxf(mx, cp[(size_t)k * hw]);
  float sum = 0.0f, dsum = 0.0f;
  for (int k = 0; k < D; ++k) {
    const float e = expf(cp[(size_t)k * hw] - mx);
    sum = sum + e;
    dsum = dsum + e * dp[(size_t)k * hw];
  }
  const float depth = dsum / sum, g = grad_depth[(size_t)b * hw + p];
  float *gc = grad_cost + (size_t)b * D * hw + p;
  for (int k = 0; k < D; ++k) {
    const float pk = expf(cp[(size_t)k * hw] - mx) / sum;
    gc[(size_t)k * hw] = g * (pk * (dp[(size_t)k * hw] - depth));
  }
}

}  // namespace

extern "C" int casmvs_homo_warp_backward_f32(const float *grad_out, const float *proj, const float *depth, float *grad_src,
                                             int B, int C, int H, int W, int D, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(grad_out && proj && depth && grad_src, "homo_warp_backward: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 1 && W > 1 && D > 0 && D <= 65535, "homo_warp_backward: bad shape B=%d C=%d H=%d W=%d D=%d", B, C, H, W, D);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(grad_src, 0, (size_t)B * C * H * W * sizeof(float), st);
  if (e != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "homo_warp_backward: hipMemsetAsync: %s", hipGetErrorString(e));
  dim3 grid((unsigned)casmvs::ceil_div(H * W, kThreads), (unsigned)D, (unsigned)B);
  hipLaunchKernelGGL(homo_warp_bwd_kernel, grid, dim3(kThreads), 0, st, grad_out, proj, depth, grad_src, C, H, W, D);
  return casmvs::check_launch("homo_warp_bwd_kernel");
}

extern "C" int casmvs_softmax_regress_backward_f32(const float *cost, const float *depth_values, const float *grad_depth,
                                                   float *grad_cost, int B, int D, int h, int w, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(cost && depth_values && grad_depth && grad_cost, "softmax_regress_backward: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && D > 0 && h > 0 && w > 0, "softmax_regress_backward: bad shape B=%d D=%d h=%d w=%d", B, D, h, w);
  dim3 grid((unsigned)casmvs::ceil_div(h * w, kThreads), (unsigned)B);
  hipLaunchKernelGGL(softmax_regress_bwd_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, cost, depth_values, grad_depth,
                     grad_cost, D, h * w);
  return casmvs::check_launch("softmax_regress_bwd_kernel");
}
