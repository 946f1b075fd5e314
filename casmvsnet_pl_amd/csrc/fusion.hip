// Depth filtering / fusion of one reference view against its source views.
//
// Reference semantics: eval.py:113-182 (xy_ref2src, xy_src2ref, check_geo_consistency: numba + cv2.remap) and
// eval.py:273-326 (confidence mask via cv2.resize x4, geometric mask count, depth / colour averaging, world points).
// The two OpenCV functions are restated (oracle/fusion_restatement.py states from what): remap = 5-bit fixed-point
// coordinates + a 32 x 32 weight table (float weights for the depth map, 15-bit integer weights for the 8-bit image),
// constant-0 border; resize = half-pixel-centre bilinear, horizontal then vertical.
//
// One thread per reference pixel, lanes along the row: the reference maps are read / written coalesced; each source
// view costs 4 depth taps + 4 x 3 byte taps around one projected point - neighbouring pixels project to neighbouring
// points, so the gathers of a wavefront fall into a few cache lines.  HBM-bound (algorithmic bytes per reference
// view: H W (4 + 3 + S (4 + 3)) read, H W (4 + 24 + 4 + 1 + 12) written).  Built with -ffp-contract=off: every
// float operation below is separately rounded, in the oracle's order; masks and counts are integer work and must
// match the oracle bit for bit.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int INTER_BITS = 5, INTER_TAB = 32;

struct FuseArgs {
  const float *depth_ref;          // (H, W)
  const unsigned char *image_ref;  // (H, W, 3)
  const float *proba_quarter;      // (H/4, W/4) or nullptr (no confidence mask)
  const float *depth_src;          // (S, H, W)
  const unsigned char *image_src;  // (S, H, W, 3)
  const float *m_ref2src;          // (S, 3, 4)  (P_src @ inv(P_ref))[:3]
  const float *m_src2ref;          // (S, 3, 4)  (P_ref @ inv(P_src))[:3]
  const float *m_ref2world;        // (3, 4) rows of inv(P_ref), or nullptr (no world points)
  float *depth_refined;            // (H, W)
  double *image_refined;           // (H, W, 3)
  int *mask_geo_sum;               // (H, W)
  unsigned char *mask_final;       // (H, W)
  float *xyz_world;                // (H, W, 3) or nullptr
  unsigned char *mask_geo;         // (S, H, W) or nullptr: the per-view masks of check_geo_consistency
  float *depth_reproj;             // (S, H, W) or nullptr: the per-view masked reprojected depths
  unsigned char *image_s2r;        // (S, H, W, 3) or nullptr: the per-view masked warped source images
  int S, H, W;
  float conf;
  int min_geo_consistent;
};

// cvRound(v * 32): round half to even; NaN / beyond int32 -> INT_MIN like x86's cvtss2si
__device__ __forceinline__ int fixed_coord(float v) {
  const float r = rintf(v * (float)INTER_TAB);
  return (fabsf(r) < 2147483648.0f) ? (int)r : (int)0x80000000;
}

__device__ __forceinline__ float project_row(const float *m, float X, float Y, float Z) {
  return ((m[0] * X + m[1] * Y) + m[2] * Z) + m[3];
}

// cv2.resize(proba, fx = 4, fy = 4, INTER_LINEAR) at destination pixel (x, y): hresize of the two rows, then vresize
__device__ __forceinline__ float resize_x4(const float *__restrict__ p, int hq, int wq, int x, int y) {
  auto coef = [](int d, int n, int &s0, int &s1, float &a) {
    const float f = (float)(((double)d + 0.5) * 0.25 - 0.5);
    int s = (int)floorf(f);
    a = f - (float)s;
    if (s < 0) { s = 0; a = 0.0f; }
    if (s >= n - 1) { s = n - 1; a = 0.0f; }
    s0 = s;
    s1 = min(s + 1, n - 1);
  };
  int x0, x1, y0, y1;
  float ax, ay;
  coef(x, wq, x0, x1, ax);
  coef(y, hq, y0, y1, ay);
  const float r0 = p[y0 * wq + x0] * (1.0f - ax) + p[y0 * wq + x1] * ax;
  const float r1 = p[y1 * wq + x0] * (1.0f - ax) + p[y1 * wq + x1] * ax;
  return r0 * (1.0f - ay) + r1 * ay;
}

__global__ __launch_bounds__(kThreads) void fuse_view_kernel(const FuseArgs a) {
  const int x = blockIdx.x * kThreads + threadIdx.x, y = blockIdx.y;
  if (x >= a.W) return;
  const int H = a.H, W = a.W, hw = H * W, p = y * W + x;
  const float d = a.depth_ref[p];
  const float xf = (float)x, yf = (float)y;
  const float X = xf * d, Y = yf * d;                       // eval.py:117: (x, y, 1) * depth_ref
  float dsum = d;                                           // np.sum([depth_ref, reproj_1, ...], 0): sequential float32
  unsigned isum[3] = {a.image_ref[3 * p], a.image_ref[3 * p + 1], a.image_ref[3 * p + 2]};
  int nsum = 0;
  for (int s = 0; s < a.S; ++s) {
    const float *M = a.m_ref2src + s * 12;
    const float q0 = project_row(M, X, Y, d), q1 = project_row(M + 4, X, Y, d), q2 = project_row(M + 8, X, Y, d);
    const float u = q0 / q2, v = q1 / q2;                   // eval.py:123
    // cv2.remap: fixed-point coordinates, 2x2 taps with constant-0 border
    const int sx = fixed_coord(u), sy = fixed_coord(v);
    const int ix = min(max(sx >> INTER_BITS, -32768), 32767), iy = min(max(sy >> INTER_BITS, -32768), 32767);
    const int fx = sx & (INTER_TAB - 1), fy = sy & (INTER_TAB - 1);
    const float ax = (float)fx / (float)INTER_TAB, ay = (float)fy / (float)INTER_TAB;
    const float w0 = (1.0f - ay) * (1.0f - ax), w1 = (1.0f - ay) * ax, w2 = ay * (1.0f - ax), w3 = ay * ax;
    const bool x0in = ix >= 0 && ix < W, x1in = ix + 1 >= 0 && ix + 1 < W;
    const bool y0in = iy >= 0 && iy < H, y1in = iy + 1 >= 0 && iy + 1 < H;
    const int cx0 = min(max(ix, 0), W - 1), cx1 = min(max(ix + 1, 0), W - 1);
    const int cy0 = min(max(iy, 0), H - 1), cy1 = min(max(iy + 1, 0), H - 1);
    // Every tap is loaded from its CLAMPED (always valid) address and zeroed by a select afterwards: 4 + 12 independent loads
    // in flight per source view.  (Written as `inside ? load : 0` the compiler made 16 exec-masked branches whose loads
    // completed one after the other: 16 serial memory latencies per view, the kernel ran at 0.13 of the HBM roof.)
    const float *ds = a.depth_src + (size_t)s * hw;
    const int o00 = cy0 * W + cx0, o01 = cy0 * W + cx1, o10 = cy1 * W + cx0, o11 = cy1 * W + cx1;
    const bool in00 = x0in && y0in, in01 = x1in && y0in, in10 = x0in && y1in, in11 = x1in && y1in;
    const float l0 = ds[o00], l1 = ds[o01], l2 = ds[o10], l3 = ds[o11];
    const unsigned char *is = a.image_src + (size_t)s * hw * 3;
    int bt[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bt[0][c] = is[o00 * 3 + c]; bt[1][c] = is[o01 * 3 + c]; bt[2][c] = is[o10 * 3 + c]; bt[3][c] = is[o11 * 3 + c];
    }
    const float t0 = in00 ? l0 : 0.0f, t1 = in01 ? l1 : 0.0f, t2 = in10 ? l2 : 0.0f, t3 = in11 ? l3 : 0.0f;
    const float d_s2r = ((t0 * w0 + t1 * w1) + t2 * w2) + t3 * w3;   // remapBilinear, CV_32F
    // 8-bit image: OpenCV's integer weight table, saturate_cast<short>(w * 2^15) with the rounding error of the four
    // entries pushed onto the largest / smallest one.  With 5-bit fractions the weights are (32 - fy)(32 - fx) / 1024 ...
    // fy fx / 1024: exact in float32, and w * 2^15 = 32 x an integer product - the rounding is the identity, the four
    // entries sum to exactly 2^15 and the correction never fires.  (The float64 rint / argmax / argmin form of the
    // oracle, oracle/fusion_restatement.py, gives the same integers; it cost 16 fp64 instructions per source view.)
    const int wi[4] = {(INTER_TAB - fy) * (INTER_TAB - fx) * 32, (INTER_TAB - fy) * fx * 32, fy * (INTER_TAB - fx) * 32, fy * fx * 32};
    int col[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int b0 = in00 ? bt[0][c] : 0, b1 = in01 ? bt[1][c] : 0, b2 = in10 ? bt[2][c] : 0, b3 = in11 ? bt[3][c] : 0;
      const int acc = b0 * wi[0] + b1 * wi[1] + b2 * wi[2] + b3 * wi[3];
      col[c] = min(max((acc + (1 << 14)) >> 15, 0), 255);
    }
    // back to the reference view with the sampled depth            (eval.py:130-153)
    const float X2 = u * d_s2r, Y2 = v * d_s2r;
    const float *Q = a.m_src2ref + s * 12;
    const float r0 = project_row(Q, X2, Y2, d_s2r), r1 = project_row(Q + 4, X2, Y2, d_s2r), r2 = project_row(Q + 8, X2, Y2, d_s2r);
    const float ddx = r0 / r2 - xf, ddy = r1 / r2 - yf;
    const bool m_pix = (ddx * ddx + ddy * ddy) < 1.0f;            // |p_reproj - p| < 1
    const bool m_dep = fabsf((r2 - d) / d) < 0.01f;               // |d_reproj - d| / d < 0.01   (NaN -> false)
    const bool m = m_pix && m_dep;
    const float dr = m ? r2 : 0.0f;                               // depth_ref_reproj[~mask_geo] = 0
    dsum = dsum + dr;
    nsum += m ? 1 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) isum[c] += m ? (unsigned)col[c] : 0u;
    if (a.mask_geo) a.mask_geo[(size_t)s * hw + p] = m ? 1 : 0;
    if (a.depth_reproj) a.depth_reproj[(size_t)s * hw + p] = dr;
    if (a.image_s2r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.image_s2r[((size_t)s * hw + p) * 3 + c] = m ? (unsigned char)col[c] : 0;
    }
  }
  // float32 sum / int count in float64, rounded to float32          (eval.py:312-313)
  const float drf = (float)((double)dsum / (double)(nsum + 1));
  a.depth_refined[p] = drf;
#pragma unroll
  for (int c = 0; c < 3; ++c) a.image_refined[(size_t)3 * p + c] = (double)isum[c] / (double)(nsum + 1);
  a.mask_geo_sum[p] = nsum;
  bool m_conf = true;
  if (a.proba_quarter) m_conf = resize_x4(a.proba_quarter, H / 4, W / 4, x, y) > a.conf;   // eval.py:281-283
  a.mask_final[p] = (m_conf && nsum >= a.min_geo_consistent) ? 1 : 0;
  if (a.xyz_world) {   // inv(P_ref) @ (x d, y d, d, 1): int64 * float32 -> float64 products (eval.py:322-327)
    const double Xw = (double)x * (double)drf, Yw = (double)y * (double)drf, Zw = (double)drf;
    const float *Mi = a.m_ref2world;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      a.xyz_world[(size_t)3 * p + i] = (float)((((double)Mi[4 * i] * Xw + (double)Mi[4 * i + 1] * Yw) + (double)Mi[4 * i + 2] * Zw) + (double)Mi[4 * i + 3]);
  }
}

// ---- the same arithmetic with a quarter of the memory instructions (round 3; measured: bit-identical, 95 -> 83 us) ----------
// fuse_view_kernel issues per source view 4 dword + 8 byte / short tap loads and 6 x 16-byte loads of the two 3 x 4 matrices (the
// per-view outputs it may store forbid scalar loads): ~18 vector-memory instructions per (pixel, view), each a 64-address
// gather for the CU's address unit - 105 us at 0.14 of the HBM roof.  Here the two taps of a row come from ONE load -
// 8 bytes of depth (bx, bx + 1) and 8 bytes holding the six colour bytes, bx = min(x0, W - 2) and selects for the clamped
// border columns - and the matrices are staged in LDS once per workgroup: 4 vector-memory instructions per (pixel, view).
// What bounds both kernels is the vector ALU: 255 instructions per (pixel, view) here - five correctly rounded divisions (55), six
// 3 x 4 projections (36), the fixed-point coordinates and weights (~30), the colour interpolation (~60), the border selects (31) -
// at 4 cycles per wave64 instruction: 67 us for 995 k pixels x 10 views on 256 CUs, 83 measured; the HBM floor is 15 us.
// Every float operation is the same expression in the same order as above; results must be bit-identical.
constexpr int kMaxViewsLds = 64;

__device__ __forceinline__ uint64_t load_u64_bytes(const unsigned char *p) {
  uint64_t v;
  __builtin_memcpy(&v, p, 8);
  return v;
}

__global__ __launch_bounds__(kThreads) void fuse_view_paired_kernel(const FuseArgs a) {
  __shared__ float mats[kMaxViewsLds * 24];   // [view][ref2src 12 | src2ref 12]
  for (int i = threadIdx.x; i < a.S * 12; i += kThreads) {
    const int s = i / 12, k = i - 12 * s;
    mats[s * 24 + k] = a.m_ref2src[i];
    mats[s * 24 + 12 + k] = a.m_src2ref[i];
  }
  __syncthreads();
  const int x = blockIdx.x * kThreads + threadIdx.x, y = blockIdx.y;
  if (x >= a.W) return;
  const int H = a.H, W = a.W, hw = H * W, p = y * W + x;
  const float d = a.depth_ref[p];
  const float xf = (float)x, yf = (float)y;
  const float X = xf * d, Y = yf * d;
  float dsum = d;
  unsigned isum[3] = {a.image_ref[3 * p], a.image_ref[3 * p + 1], a.image_ref[3 * p + 2]};
  int nsum = 0;
  for (int s = 0; s < a.S; ++s) {
    const float *M = mats + s * 24;
    const float q0 = project_row(M, X, Y, d), q1 = project_row(M + 4, X, Y, d), q2 = project_row(M + 8, X, Y, d);
    const float u = q0 / q2, v = q1 / q2;
    const int sx = fixed_coord(u), sy = fixed_coord(v);
    const int ix = min(max(sx >> INTER_BITS, -32768), 32767), iy = min(max(sy >> INTER_BITS, -32768), 32767);
    const int fx = sx & (INTER_TAB - 1), fy = sy & (INTER_TAB - 1);
    const float ax = (float)fx / (float)INTER_TAB, ay = (float)fy / (float)INTER_TAB;
    const float w0 = (1.0f - ay) * (1.0f - ax), w1 = (1.0f - ay) * ax, w2 = ay * (1.0f - ax), w3 = ay * ax;
    const bool x0in = ix >= 0 && ix < W, x1in = ix + 1 >= 0 && ix + 1 < W;
    const bool y0in = iy >= 0 && iy < H, y1in = iy + 1 >= 0 && iy + 1 < H;
    const int cx0 = min(max(ix, 0), W - 1), cx1 = min(max(ix + 1, 0), W - 1);
    const int cy0 = min(max(iy, 0), H - 1), cy1 = min(max(iy + 1, 0), H - 1);
    const bool in00 = x0in && y0in, in01 = x1in && y0in, in10 = x0in && y1in, in11 = x1in && y1in;
    // cx1 is cx0 or cx0 + 1: both lie in the pair (bx, bx + 1)
    const int bx = min(cx0, W - 2);
    const bool hi0 = cx0 != bx, hi1 = cx1 != bx;
    const int r0 = cy0 * W + bx, r1 = cy1 * W + bx;
    const float *ds = a.depth_src + (size_t)s * hw;
    float2 pa, pb;
    __builtin_memcpy(&pa, ds + r0, 8);
    __builtin_memcpy(&pb, ds + r1, 8);
    // six colour bytes of a pair inside one 8-byte load; the last pair of a view would read 2 bytes past it: start 2 bytes earlier
    const unsigned char *is = a.image_src + (size_t)s * hw * 3;
    const int back0 = r0 >= hw - 2 ? 2 : 0, back1 = r1 >= hw - 2 ? 2 : 0;
    const uint64_t ca = load_u64_bytes(is + 3 * r0 - back0) >> (8 * back0), cb = load_u64_bytes(is + 3 * r1 - back1) >> (8 * back1);
    const float l0 = hi0 ? pa.y : pa.x, l1 = hi1 ? pa.y : pa.x, l2 = hi0 ? pb.y : pb.x, l3 = hi1 ? pb.y : pb.x;
    const unsigned a0 = (unsigned)(hi0 ? ca >> 24 : ca), a1 = (unsigned)(hi1 ? ca >> 24 : ca);
    const unsigned b0w = (unsigned)(hi0 ? cb >> 24 : cb), b1w = (unsigned)(hi1 ? cb >> 24 : cb);
    const float t0 = in00 ? l0 : 0.0f, t1 = in01 ? l1 : 0.0f, t2 = in10 ? l2 : 0.0f, t3 = in11 ? l3 : 0.0f;
    const float d_s2r = ((t0 * w0 + t1 * w1) + t2 * w2) + t3 * w3;
    const int wi[4] = {(INTER_TAB - fy) * (INTER_TAB - fx) * 32, (INTER_TAB - fy) * fx * 32, fy * (INTER_TAB - fx) * 32, fy * fx * 32};
    int col[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int b0 = in00 ? (int)((a0 >> (8 * c)) & 255u) : 0, b1 = in01 ? (int)((a1 >> (8 * c)) & 255u) : 0;
      const int b2 = in10 ? (int)((b0w >> (8 * c)) & 255u) : 0, b3 = in11 ? (int)((b1w >> (8 * c)) & 255u) : 0;
      const int acc = b0 * wi[0] + b1 * wi[1] + b2 * wi[2] + b3 * wi[3];
      col[c] = min(max((acc + (1 << 14)) >> 15, 0), 255);
    }
    const float X2 = u * d_s2r, Y2 = v * d_s2r;
    const float *Q = M + 12;
    const float r0f = project_row(Q, X2, Y2, d_s2r), r1f = project_row(Q + 4, X2, Y2, d_s2r), r2 = project_row(Q + 8, X2, Y2, d_s2r);
    const float ddx = r0f / r2 - xf, ddy = r1f / r2 - yf;
    const bool m_pix = (ddx * ddx + ddy * ddy) < 1.0f;
    const bool m_dep = fabsf((r2 - d) / d) < 0.01f;
    const bool m = m_pix && m_dep;
    const float dr = m ? r2 : 0.0f;
    dsum = dsum + dr;
    nsum += m ? 1 : 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) isum[c] += m ? (unsigned)col[c] : 0u;
    if (a.mask_geo) a.mask_geo[(size_t)s * hw + p] = m ? 1 : 0;
    if (a.depth_reproj) a.depth_reproj[(size_t)s * hw + p] = dr;
    if (a.image_s2r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.image_s2r[((size_t)s * hw + p) * 3 + c] = m ? (unsigned char)col[c] : 0;
    }
  }
  const float drf = (float)((double)dsum / (double)(nsum + 1));
  a.depth_refined[p] = drf;
#pragma unroll
  for (int c = 0; c < 3; ++c) a.image_refined[(size_t)3 * p + c] = (double)isum[c] / (double)(nsum + 1);
  a.mask_geo_sum[p] = nsum;
  bool m_conf = true;
  if (a.proba_quarter) m_conf = resize_x4(a.proba_quarter, H / 4, W / 4, x, y) > a.conf;
  a.mask_final[p] = (m_conf && nsum >= a.min_geo_consistent) ? 1 : 0;
  if (a.xyz_world) {
    const double Xw = (double)x * (double)drf, Yw = (double)y * (double)drf, Zw = (double)drf;
    const float *Mi = a.m_ref2world;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      a.xyz_world[(size_t)3 * p + i] = (float)((((double)Mi[4 * i] * Xw + (double)Mi[4 * i + 1] * Yw) + (double)Mi[4 * i + 2] * Zw) + (double)Mi[4 * i + 3]);
  }
}

}  // namespace

namespace {
int fuse_launch(bool paired, const float *depth_ref, const unsigned char *image_ref, const float *proba_quarter, const float *depth_src,
                const unsigned char *image_src, const float *m_ref2src, const float *m_src2ref, const float *m_ref2world, float *depth_refined,
                double *image_refined, int32_t *mask_geo_sum, unsigned char *mask_final, float *xyz_world, unsigned char *mask_geo,
                float *depth_reproj, unsigned char *image_s2r, int S, int H, int W, float conf, int min_geo_consistent, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(depth_ref && image_ref && depth_refined && image_refined && mask_geo_sum && mask_final, "fuse_reference_view: null pointer");
  CASMVS_REQUIRE(S >= 0 && H > 0 && W > 0 && H <= 65535, "fuse_reference_view: bad shape S=%d H=%d W=%d", S, H, W);
  CASMVS_REQUIRE(S == 0 || (depth_src && image_src && m_ref2src && m_src2ref), "fuse_reference_view: null source-view pointer");
  CASMVS_REQUIRE(!proba_quarter || (H % 4 == 0 && W % 4 == 0), "fuse_reference_view: the confidence map is (H/4, W/4): H, W must be multiples of 4");
  CASMVS_REQUIRE(!xyz_world || m_ref2world, "fuse_reference_view: xyz_world needs m_ref2world");
  CASMVS_REQUIRE(!paired || (W >= 2 && H >= 2 && S <= kMaxViewsLds), "fuse_reference_view_paired: needs W, H >= 2 and at most %d source views (got S=%d H=%d W=%d)", kMaxViewsLds, S, H, W);
  FuseArgs a{depth_ref, image_ref, proba_quarter, depth_src, image_src, m_ref2src, m_src2ref, m_ref2world, depth_refined,
             image_refined, mask_geo_sum, mask_final, xyz_world, mask_geo, depth_reproj, image_s2r, S, H, W, conf, min_geo_consistent};
  dim3 grid((unsigned)casmvs::ceil_div(W, kThreads), (unsigned)H);
  if (paired) {
    hipLaunchKernelGGL(fuse_view_paired_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
    return casmvs::check_launch("fuse_view_paired_kernel");
  }
  hipLaunchKernelGGL(fuse_view_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, a);
  return casmvs::check_launch("fuse_view_kernel");
}
}  // namespace

extern "C" int casmvs_fuse_reference_view(const float *depth_ref, const unsigned char *image_ref, const float *proba_quarter,
                                          const float *depth_src, const unsigned char *image_src, const float *m_ref2src,
                                          const float *m_src2ref, const float *m_ref2world, float *depth_refined,
                                          double *image_refined, int32_t *mask_geo_sum, unsigned char *mask_final,
                                          float *xyz_world, unsigned char *mask_geo, float *depth_reproj, unsigned char *image_s2r,
                                          int S, int H, int W, float conf, int min_geo_consistent, void *stream) {
  return fuse_launch(false, depth_ref, image_ref, proba_quarter, depth_src, image_src, m_ref2src, m_src2ref, m_ref2world, depth_refined, image_refined,
                     mask_geo_sum, mask_final, xyz_world, mask_geo, depth_reproj, image_s2r, S, H, W, conf, min_geo_consistent, stream);
}

extern "C" int casmvs_fuse_reference_view_paired(const float *depth_ref, const unsigned char *image_ref, const float *proba_quarter,
                                                 const float *depth_src, const unsigned char *image_src, const float *m_ref2src,
                                                 const float *m_src2ref, const float *m_ref2world, float *depth_refined,
                                                 double *image_refined, int32_t *mask_geo_sum, unsigned char *mask_final,
                                                 float *xyz_world, unsigned char *mask_geo, float *depth_reproj, unsigned char *image_s2r,
                                                 int S, int H, int W, float conf, int min_geo_consistent, void *stream) {
  return fuse_launch(true, depth_ref, image_ref, proba_quarter, depth_src, image_src, m_ref2src, m_src2ref, m_ref2world, depth_refined, image_refined,
                     mask_geo_sum, mask_final, xyz_world, mask_geo, depth_reproj, image_s2r, S, H, W, conf, min_geo_consistent, stream);
}
