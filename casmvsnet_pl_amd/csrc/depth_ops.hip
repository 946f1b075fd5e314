// Depth-hypothesis generation and softmax / soft-argmin regression / confidence kernels.
//
// Reference semantics: models/mvsnet.py:213-235 + models/modules.py:34-49 (hypotheses),
// models/mvsnet.py:174-193 + models/modules.py:95-104 (softmax, regression, confidence).
// Both are HBM-bound streaming kernels: one thread per pixel, lanes along the image row so every
// (D, h, w) plane access is a coalesced 256 B wavefront transaction; the D values of a pixel stay
// in registers between the max / exp-sum / regression passes (one read of the cost volume).
// Built with -ffp-contract=off: every * and + below is a separately rounded fp32 operation, in
// the reference's (torch's) order.
#include "common.h"
#include "softmax_regress.h"

namespace {

constexpr int kThreads = 256;

// ---- hypotheses ------------------------------------------------------------------------------
// Coarsest level (prev == nullptr): d_k = depth_min[b] + interval[b] * k      (mvsnet.py:216-229)
// Finer levels: u = bilinear x2 upsample (align_corners=True) of prev (mvsnet.py:232-234, ATen
// UpSample.h area_pixel_compute_source_index / compute_source_index_and_lambda), then
// d_min = max(u - half_range[b], 1e-7), d_k = d_min + interval[b] * k          (modules.py:44-48)
__global__ __launch_bounds__(kThreads) void hypotheses_kernel(
    const float *__restrict__ prev, const float *__restrict__ depth_min_b,
    const float *__restrict__ interval_b, const float *__restrict__ half_range_b,
    float *__restrict__ out, int D, int h, int w, int hp, int wp) {
  const int b = blockIdx.y;
  const int hw = h * w;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const float delta = interval_b[b];
  float dmin;
  if (prev == nullptr) {
    dmin = depth_min_b[b];
  } else {
    const int y = p / w, x = p - y * w;
    // ATen: scale = (in - 1) / (out - 1) computed in float; src = scale * dst
    const float sy = (h > 1) ? (float)(hp - 1) / (float)(h - 1) : 0.0f;
    const float sx = (w > 1) ? (float)(wp - 1) / (float)(w - 1) : 0.0f;
    const float fy = sy * (float)y, fx = sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;  // fy, fx >= 0: truncation == floor
    const int y1 = y0 + ((y0 < hp - 1) ? 1 : 0), x1 = x0 + ((x0 < wp - 1) ? 1 : 0);
    const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
    const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
    const float *pp = prev + (size_t)b * hp * wp;
    const float v00 = pp[y0 * wp + x0], v01 = pp[y0 * wp + x1];
    const float v10 = pp[y1 * wp + x0], v11 = pp[y1 * wp + x1];
    // ATen upsample_bilinear2d: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (...)
    const float top = lx0 * v00 + lx1 * v01;
    const float bot = lx0 * v10 + lx1 * v11;
    const float u = ly0 * top + ly1 * bot;
    dmin = fmaxf(u - half_range_b[b], 1e-7f);  // torch.clamp_min (NaN propagates below)
    if (u != u) dmin = u;
  }
  float *op = out + (size_t)b * D * hw + p;
  for (int k = 0; k < D; ++k) op[(size_t)k * hw] = dmin + delta * (float)k;
}

// ---- softmax + regression + confidence ---------------------------------------------------------
template <int DT>  // DT > 0: D == DT, values cached in registers; DT == 0: generic 3-pass (softmax_regress.h)
__global__ __launch_bounds__(kThreads) void softmax_regress_kernel(
    const float *__restrict__ cost, const float *__restrict__ dvals, float *__restrict__ depth,
    float *__restrict__ conf, int32_t *__restrict__ index, int Drt, int hw) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const int D = DT > 0 ? DT : Drt;
  float dsum, c4;
  int idx;
  casmvs::softmax_regress_pixel<DT>(cost + (size_t)b * D * hw + p, dvals + (size_t)b * D * hw + p, (size_t)hw, D, dsum, c4, idx);
  depth[(size_t)b * hw + p] = dsum;
  conf[(size_t)b * hw + p] = c4;
  if (index) index[(size_t)b * hw + p] = idx;
}

// ---- depth_regression (modules.py:95-104): sum_k p_k * d_k, d per voxel (B, D, h, w) or per plane (D) -------
__global__ __launch_bounds__(kThreads) void depth_regression_kernel(const float *__restrict__ prob,
                                                                   const float *__restrict__ dvals, float *__restrict__ out,
                                                                   int D, int hw, int per_plane) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const float *pp = prob + (size_t)b * D * hw + p;
  const float *dp = per_plane ? dvals : dvals + (size_t)b * D * hw + p;
  const size_t ds = per_plane ? 1 : (size_t)hw;
  float acc = 0.0f;
  for (int k = 0; k < D; ++k) acc = acc + pp[(size_t)k * hw] * dp[(size_t)k * ds];
  out[(size_t)b * hw + p] = acc;
}

// ---- input images: uint8 HWC -> normalised float CHW ---------------------------------------------------------
// datasets/dtu.py:134-137 (T.ToTensor + T.Normalize): x = u8 / 255, then (x - mean[c]) / std[c], all float32 - done on
// the device so that the host uploads 1 byte per sample instead of 4.  Thread per pixel, three coalesced plane stores.
__global__ __launch_bounds__(kThreads) void normalize_u8_kernel(const unsigned char *__restrict__ in, float *__restrict__ out,
                                                               int hw, float m0, float m1, float m2, float s0, float s1, float s2) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= hw) return;
  const unsigned char *ip = in + ((size_t)n * hw + p) * 3;
  float *op = out + (size_t)n * 3 * hw + p;
  op[0] = ((float)ip[0] / 255.0f - m0) / s0;
  op[hw] = ((float)ip[1] / 255.0f - m1) / s1;
  op[2 * (size_t)hw] = ((float)ip[2] / 255.0f - m2) / s2;
}

}  // namespace

extern "C" int casmvs_normalize_images_u8(const unsigned char *images, float *out, int N, int H, int W, const float *mean3,
                                          const float *std3, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(images && out && mean3 && std3, "normalize_images: null pointer");
  CASMVS_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0, "normalize_images: bad shape N=%d H=%d W=%d", N, H, W);
  dim3 grid((unsigned)casmvs::ceil_div(H * W, kThreads), (unsigned)N);
  hipLaunchKernelGGL(normalize_u8_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, images, out, H * W, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2]);
  return casmvs::check_launch("normalize_u8_kernel");
}

extern "C" int casmvs_depth_regression_f32(const float *prob, const float *depth_values, float *out, int B, int D,
                                           int h, int w, int depth_values_per_plane, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(prob && depth_values && out, "depth_regression: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && D > 0 && h > 0 && w > 0, "depth_regression: bad shape B=%d D=%d h=%d w=%d", B, D, h, w);
  dim3 grid((unsigned)casmvs::ceil_div(h * w, kThreads), (unsigned)B);
  hipLaunchKernelGGL(depth_regression_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, prob, depth_values, out, D,
                     h * w, depth_values_per_plane ? 1 : 0);
  return casmvs::check_launch("depth_regression_kernel");
}

extern "C" int casmvs_depth_hypotheses_f32(const float *prev_depth, const float *depth_min_b,
                                           const float *interval_b, const float *half_range_b,
                                           float *out, int B, int D, int h, int w, int hp, int wp,
                                           void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(interval_b && out, "depth_hypotheses: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && D > 0 && h > 0 && w > 0, "depth_hypotheses: bad shape B=%d D=%d h=%d w=%d", B, D, h, w);
  if (prev_depth) {
    CASMVS_REQUIRE(half_range_b, "depth_hypotheses: half_range_b is required with prev_depth");
    CASMVS_REQUIRE(hp > 0 && wp > 0, "depth_hypotheses: bad previous shape hp=%d wp=%d", hp, wp);
  } else {
    CASMVS_REQUIRE(depth_min_b, "depth_hypotheses: depth_min_b is required without prev_depth");
  }
  dim3 grid((unsigned)casmvs::ceil_div(h * w, kThreads), (unsigned)B);
  hipLaunchKernelGGL(hypotheses_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, prev_depth,
                     depth_min_b, interval_b, half_range_b, out, D, h, w, hp, wp);
  return casmvs::check_launch("hypotheses_kernel");
}

extern "C" int casmvs_softmax_regress_f32(const float *cost, const float *depth_values,
                                          float *depth, float *confidence, int32_t *index, int B,
                                          int D, int h, int w, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(cost && depth_values && depth && confidence, "softmax_regress: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && D > 0 && h > 0 && w > 0, "softmax_regress: bad shape B=%d D=%d h=%d w=%d", B, D, h, w);
  const int hw = h * w;
  dim3 grid((unsigned)casmvs::ceil_div(hw, kThreads), (unsigned)B), blk(kThreads);
  hipStream_t st = (hipStream_t)stream;
#define CASMVS_SR(DT)                                                                            \
  hipLaunchKernelGGL((softmax_regress_kernel<DT>), grid, blk, 0, st, cost, depth_values, depth, \
                     confidence, index, D, hw)
  switch (D) {
    case 8: CASMVS_SR(8); break;
    case 16: CASMVS_SR(16); break;
    case 32: CASMVS_SR(32); break;
    case 48: CASMVS_SR(48); break;
    case 64: CASMVS_SR(64); break;
    default: CASMVS_SR(0); break;
  }
#undef CASMVS_SR
  return casmvs::check_launch("softmax_regress_kernel");
}
