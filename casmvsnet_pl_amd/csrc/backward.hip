// Backward kernels of the plane sweep and of the softmax depth regression (training support op by op).
//
// Reference semantics: the autograd graph of models/modules.py:52-92 (homo_warp: F.grid_sample's gradient with
// respect to its INPUT - the sampling grid depends on the detached depth hypotheses only, mvsnet.py:231, so no
// gradient flows into it) and of models/mvsnet.py:175-177 + modules.py:95-104 (softmax over depth, then
// depth = sum_k p_k d_k; the confidence is computed under torch.no_grad(), mvsnet.py:179-193).
#include "common.h"
#include "plane_sweep.h"
#include "fixed_accum.h"

namespace {

using namespace casmvs_dev;

constexpr int kThreads = 256;

// grad_src[b, c, tap] += grad_out[b, c, d, y, x] * w_tap: the transpose of the forward gather is a scatter-add.
// One thread per (reference pixel, plane), lanes along the row: the taps (same arithmetic as the forward, plane_sweep.h)
// are computed once and reused for every channel.  The sums are 64-bit fixed point (fixed_accum.h): unlike ATen's own grid_sampler backward (float
// atomics) the result does NOT depend on the order of the adds - the same bits run to run.  Scale per (sample, channel): a contribution is g w with
// w <= 1, so 2^be >= 2 G with G the channel's largest finite |grad_out|.
__global__ __launch_bounds__(kThreads) void homo_warp_bwd_kernel(const float *__restrict__ grad_out, const float *__restrict__ proj,
                                                                const float *__restrict__ depth, float *__restrict__ grad_src,
                                                                unsigned long long *__restrict__ acc, const unsigned *__restrict__ gmax, int C, int H, int W,
                                                                int D, int U) {
  const int b = blockIdx.z, d = blockIdx.y;
  const int hw = H * W;
  // the scale of a (sample, channel) does not depend on the pixel: once per workgroup into LDS (round-5 advisor finding: every thread rebuilt it - two
  // float-to-double conversions, two double multiplies, a power of two - for every channel inside the scatter loop)
  extern __shared__ double ch_scale[];   // [C]
  for (int c = threadIdx.x; c < C; c += kThreads) {
    int be;
    ch_scale[c] = fixed_exponent(gmax[b * C + c], 0x3f800000u, 2.0, be) ? pow2_double(U - be) : 0.0;   // 1.0f: fixed_exponent(G, 1, 2) = the exponent of 2 G
  }
  __syncthreads();
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const int y = p / W, x = p - y * W;
  const float dv = depth[((size_t)b * D + d) * hw + p];
  const Taps t = plane_sweep_taps(proj + (size_t)b * 12, (float)x, (float)y, dv, W, H);
  if (!taps_live(t)) return;
  const float *go = grad_out + ((size_t)b * C * D + d) * hw + p;
  float *gs = grad_src + (size_t)b * C * hw;
  unsigned long long *as = acc + (size_t)b * C * hw;
  const int on = t.yn * W + t.xl, os = t.ys * W + t.xl;
  for (int c = 0; c < C; ++c) {
    const float g = go[(size_t)c * D * hw];
    const double scale = ch_scale[c];
    float *gc = gs + (size_t)c * hw;
    unsigned long long *ac = as + (size_t)c * hw;
    auto add = [&](int o, float w) {
      if (w == 0.0f) return;
      const float val = g * w;
      if (is_finite(val)) atomicAdd(ac + o, to_fixed_point(val, scale));
      else unsafeAtomicAdd(gc + o, val);
    };
    add(on, t.w_nl);
    add(on + 1, t.w_nr);
    add(os, t.w_sl);
    add(os + 1, t.w_sr);
  }
}

// grad_src (zero, or the non-finite contributions) += the fixed-point sums in float32, one rounding per element
__global__ __launch_bounds__(kThreads) void homo_warp_bwd_finish_kernel(const unsigned long long *__restrict__ acc, const unsigned *__restrict__ gmax,
                                                                       float *__restrict__ grad_src, int hw, int U) {
  const int row = blockIdx.y;   // (b, c)
  int be;
  const double from_fixed = fixed_exponent(gmax[row], 0x3f800000u, 2.0, be) ? pow2_double(be - U) : 0.0;
  const size_t base = (size_t)row * hw;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < hw; p += gridDim.x * kThreads)
    grad_src[base + p] += (float)((double)(long long)acc[base + p] * from_fixed);
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

// workspace: [acc: B C H W 64-bit fixed-point sums][gmax: B C uint32][partial maxima of volume_absmax_kernel]
extern "C" size_t casmvs_homo_warp_backward_workspace_bytes(int B, int C, int D, int H, int W) {
  if (B < 1 || C < 1 || D < 1 || H < 1 || W < 1) return 0;
  auto pad = [](size_t n) { return (n + 15) & ~(size_t)15; };
  return (size_t)B * C * H * W * sizeof(unsigned long long) + pad((size_t)B * C * sizeof(unsigned)) + pad((size_t)B * C * absmax_chunks((size_t)D * H * W) * sizeof(unsigned));
}

extern "C" int casmvs_homo_warp_backward_f32(const float *grad_out, const float *proj, const float *depth, float *grad_src, void *workspace,
                                             int B, int C, int H, int W, int D, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(grad_out && proj && depth && grad_src && workspace, "homo_warp_backward: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && C > 0 && H > 1 && W > 1 && D > 0 && D <= 65535 && (size_t)B * C <= 65535,
                 "homo_warp_backward: bad shape B=%d C=%d H=%d W=%d D=%d", B, C, H, W, D);
  CASMVS_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "homo_warp_backward: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int hw = H * W, U = casmvs::fixed_point_bits(D, H, W);
  const size_t acc_bytes = (size_t)B * C * hw * sizeof(unsigned long long);
  hipError_t e = hipMemsetAsync(grad_src, 0, (size_t)B * C * hw * sizeof(float), st);
  if (e == hipSuccess) e = hipMemsetAsync(workspace, 0, acc_bytes, st);
  if (e != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "homo_warp_backward: hipMemsetAsync: %s", hipGetErrorString(e));
  unsigned long long *acc = static_cast<unsigned long long *>(workspace);
  unsigned *gmax = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + acc_bytes);
  unsigned *partial = gmax + ((((size_t)B * C * sizeof(unsigned) + 15) & ~(size_t)15) / sizeof(unsigned));
  if (int rc = launch_volume_absmax(grad_out, nullptr, partial, gmax, nullptr, B * C, (size_t)D * hw, B, 0, C, 0, st)) return rc;
  dim3 grid((unsigned)casmvs::ceil_div(hw, kThreads), (unsigned)D, (unsigned)B);
  CASMVS_REQUIRE((size_t)C * sizeof(double) <= 48 * 1024, "homo_warp_backward: C=%d (the per-channel scales are kept in LDS)", C);
  hipLaunchKernelGGL(homo_warp_bwd_kernel, grid, dim3(kThreads), (size_t)C * sizeof(double), st, grad_out, proj, depth, grad_src, acc, gmax, C, H, W, D, U);
  if (int rc = casmvs::check_launch("homo_warp_bwd_kernel")) return rc;
  dim3 fgrid((unsigned)std::min(casmvs::ceil_div(hw, kThreads), 64), (unsigned)(B * C));
  hipLaunchKernelGGL(homo_warp_bwd_finish_kernel, fgrid, dim3(kThreads), 0, st, acc, gmax, grad_src, hw, U);
  return casmvs::check_launch("homo_warp_bwd_finish_kernel");
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
