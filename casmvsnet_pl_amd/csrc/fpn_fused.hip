// FeatureNet's full-resolution FPN tail as ONE kernel: feat0 = smooth0( lat0(conv0) + upsample2x(feat1') ).
//
// Reference semantics: models/mvsnet.py:36-38,50-51,54 - `lat0` = Conv2d(8, 32, 1), `F.interpolate(scale_factor = 2,
// bilinear, align_corners = True)` of the half-resolution 32-channel map, their sum, `smooth0` = Conv2d(32, 8, 3, pad 1).
//
// Why: as two kernels (fpn_lateral_kernel<8> + the 3x3 MFMA layer) the 32-channel full-resolution sum is written to
// and re-read from memory - 2 x 251 MB of the FeatureNet's ~1.2 GB per step at batch 2 - for a map that only ever feeds
// an 8-channel output; both kernels ran at their bandwidth / MFMA bound (88 + 116 us).  Nothing non-linear sits between
// the three operators, so
//     feat0 = (smooth0 o lat0)(conv0) + smooth0(up(feat1')) + [smooth0's valid taps] . lat0.bias + smooth0.bias
// and the kernel is a 3x3 convolution with 40 input channels of which
//   * channels 0..7 are conv0 itself, with the COMPOSED weights W_s[co, :, ky, kx] . W_l[:, ci] (host, float64);
//   * channels 8..39 are up(feat1'), never materialised: the staging step of a chunk of 4 channels interpolates its halo
//     tile from the half-resolution map (ATen's upsample_bilinear2d index / weight rule, horizontal then vertical) while
//     the matrix cores work on the previous chunk's tile;
//   * lat0's bias reaches an output pixel through the taps of smooth0 that lie inside the image (zero padding applies
//     to the SUM): nine constant vectors (top / middle / bottom x left / middle / right), added in the epilogue.
// MFMA form: PX of conv3d_mfma.hip (rows = 8 output channels x 2 x-phases, K = 4 input x-offsets of one (ci, ky);
// 3 of 4 K steps useful), weights packed by casmvs_conv2d_pack_f32(CASMVS_CONV2D_K3, cin = 40, cout = 8).
// Bound: fp32 MFMA (11.3 GFLOP at 6 x 512 x 640 incl. the composed 8-channel term: ~95 us at the PX form's 75 %);
// traffic 8 + 8 (quarter-size x 32) input channel-equivalents + 2 x 8 output channels per pixel.
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;

struct FpnCfg {
  static constexpr int TY = 8, TX = 64;          // output tile; wave w owns rows 2 w, 2 w + 1; a column tile = 32 x
  static constexpr int NT = 4;                   // column tiles per wave
  static constexpr int CK = 4;                   // input channels per chunk
  static constexpr int IY = TY + 2, IX = TX + 8; // staged rows y0 - 1 .. y0 + 8, columns x0 - 4 .. x0 + 67 (16-byte groups)
  static constexpr int SY = IX, SC = IY * IX;
  static constexpr int GROUPS = CK * IY * (IX / 4);                 // 16-byte groups per chunk: 720
  static constexpr int NK = (GROUPS + kThreads - 1) / kThreads;     // 3
  static constexpr int NW = CK * 3 * 64;                            // weight floats per chunk: [channel][ky][lane]
  static constexpr int CD = 8, CU = 32;                             // direct (conv0) / upsampled (feat1') input channels
  static constexpr int NCHUNK = (CD + CU) / CK;
  static constexpr size_t LDS_BYTES = 2 * (size_t)(CK * SC + NW + 4) * sizeof(float);   // two buffers (+ a dump unit each)
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// grid: x = tiles (XCD-major, x fastest), y = image.  c0 (N, 8, H, W), f1 (N, 32, H/2, W/2), wpk: PX image of the 40-channel
// 3x3 layer, bias9 (3, 3, 8): [row class][column class][co].  out (N, 8, H, W); out2: NULL or (N, H, W, 8) pixel-major.
__global__ __launch_bounds__(kThreads, 2) void fpn_tail0_kernel(const float *__restrict__ c0, const float *__restrict__ f1,
                                                               const float *__restrict__ wpk, const float *__restrict__ bias9,
                                                               float *__restrict__ out, float *__restrict__ out2, int H, int W,
                                                               int tiles_x) {
  using Cfg = FpnCfg;
  constexpr int NK = Cfg::NK, SY = Cfg::SY, SC = Cfg::SC, CK = Cfg::CK, NT = Cfg::NT, NW = Cfg::NW, IX = Cfg::IX, IY = Cfg::IY;
  extern __shared__ float smem[];
  // two buffers of { tile [CK][IY][IX], weights [CK][3][64] }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jcol = lane & 15, kq = lane >> 4;
  const int bid = xcd_major(blockIdx.x, gridDim.x);
  const int tx0 = (bid % tiles_x) * Cfg::TX, ty0 = (bid / tiles_x) * Cfg::TY;
  const int n = blockIdx.y;
  const int hc = H >> 1, wc = W >> 1;
  const int hw = H * W, hwc = hc * wc;
  const rsrc_t src0 = make_rsrc(c0 + (size_t)n * Cfg::CD * hw, (size_t)Cfg::CD * hw * 4);
  const rsrc_t src1 = make_rsrc(f1 + (size_t)n * Cfg::CU * hwc, (size_t)Cfg::CU * hwc * 4);
  const rsrc_t wsrc = make_rsrc(wpk, (size_t)Cfg::NCHUNK * NW * 4);

  // ---- staging plan (tile constants): group e = tid + 256 k -> (local channel, staged row, 16-byte group of the row) ----
  int voff_d[NK];            // direct chunk: byte offset in c0 (chunk channel 0), or kOOB
  int voff_u0[NK], voff_u1[NK];   // upsampled chunk: byte offsets of the 4-column windows of source rows y0, y1 in f1
  float ly0[NK], ly1[NK];    // vertical weights
  float T[NK][4][4];         // horizontal "tent" matrix: pixel j of the group = sum_m T[j][m] * window[m] (two non-zeros per
                             // row: adding exact zeros keeps ATen's lambda0 * v0 + lambda1 * v1; as fpn_lateral_kernel)
  int loff[NK];              // LDS float offset of the group (the dump unit for items past the chunk)
  const float sy = H > 1 ? (float)(hc - 1) / (float)(H - 1) : 0.0f;   // ATen: scale = (in - 1) / (out - 1) in float
  const float sx = W > 1 ? (float)(wc - 1) / (float)(W - 1) : 0.0f;
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int e = threadIdx.x + k * kThreads;
    const int c = e / (IY * (IX / 4)), r = e - c * (IY * (IX / 4));
    const int iy = r / (IX / 4), g = r - iy * (IX / 4);
    const int gy = ty0 - 1 + iy, gx = tx0 - 4 + 4 * g;
    const bool ok = e < Cfg::GROUPS && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 4 == 0: a group is inside or outside
    loff[k] = e < Cfg::GROUPS ? c * SC + iy * SY + 4 * g : CK * SC + NW;   // items past the chunk write a dump unit: no branch in the MFMA stream
    voff_d[k] = ok ? (c * hw + gy * W + gx) * 4 : kOOB;
    // ATen upsample_bilinear2d, align_corners: src = dst * scale; i0 = (int)src; i1 = i0 + (i0 < in - 1); l1 = src - i0
    const float fy = sy * (float)(ok ? gy : 0);
    const int y0 = (int)fy, y1 = y0 + (y0 < hc - 1 ? 1 : 0);
    ly1[k] = fy - (float)y0;
    ly0[k] = 1.0f - ly1[k];
    int xb = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float fx = sx * (float)((ok ? gx : 0) + j);
      const int x0 = (int)fx, x1 = x0 + (x0 < wc - 1 ? 1 : 0);
      const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
      if (j == 0) xb = x0 < wc - 4 ? x0 : wc - 4;   // the 4-column window [xb, xb + 4) holds every column the 4 pixels use
#pragma unroll
      for (int m = 0; m < 4; ++m) T[k][j][m] = (m == x0 - xb ? lx0 : 0.0f) + (m == x1 - xb ? lx1 : 0.0f);
    }
    voff_u0[k] = ok ? (c * hwc + y0 * wc + xb) * 4 : kOOB;
    voff_u1[k] = ok ? (c * hwc + y1 * wc + xb) * 4 : kOOB;
  }

  // lane's B-operand bases (PX form): column tile t of this wave -> (cy, cx); word = cy * SY + cx * 32 + 2 j + u + 3
  int base[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ct = wave * NT + t, cx = ct & 1, cy = ct >> 1;
    base[t] = cy * SY + cx * 32 + 2 * jcol + kq + 3;
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Two LDS buffers and two staging register sets, ONE barrier per chunk: while the matrix cores multiply chunk s out of
  // buffer s & 1, the registers that hold chunk s + 1 (loaded during chunk s - 1) are interpolated and written to the other
  // buffer - one item after each group of 12 MFMAs, so that the VALU work sits between matrix instructions instead of in
  // front of them - and the loads of chunk s + 2 go out into the set that chunk s occupied.  (First version: one buffer,
  // barrier - commit - barrier - MFMAs: 188 us, the matrix pipe 51 % busy.)  Every load is issued unconditionally (a chunk
  // that does not exist reads through an empty descriptor) so that the wait-count pass emits counted vmcnt waits.
  constexpr int BUF = CK * SC + NW + 4;   // floats per LDS buffer
  f32x4v ra[2][NK], rb[2][NK], wreg[2];
  const rsrc_t none = make_rsrc(c0, 0);
  auto issue = [&](auto set_, int s) {   // every load of chunk s into register set S; nothing here waits
    constexpr int S = decltype(set_)::value;
    const bool exists = s < Cfg::NCHUNK, direct = s < Cfg::CD / CK;
    wreg[S] = buf_load4(exists ? wsrc : none, threadIdx.x < NW / 4 ? threadIdx.x * 16 : kOOB, exists ? s * NW * 4 : 0);
    const rsrc_t r_a = !exists ? none : (direct ? src0 : src1), r_b = (exists && !direct) ? src1 : none;
    const int soff = !exists ? 0 : (direct ? s * CK * hw * 4 : (s - Cfg::CD / CK) * CK * hwc * 4);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      ra[S][k] = buf_load4(r_a, direct ? voff_d[k] : voff_u0[k], soff);   // upsampled: 4-byte aligned 16-byte loads (the window starts at any column)
      rb[S][k] = buf_load4(r_b, voff_u1[k], soff);
    }
  };
  auto commit_item = [&](auto set_, int k, int s, float *buf) {  // item k of chunk s (register set S) -> LDS: the interpolation happens here
    constexpr int S = decltype(set_)::value;
    f32x4v v = ra[S][k];
    if (s >= Cfg::CD / CK) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // ATen: h0lambda * (w0lambda * v00 + w1lambda * v01) + h1lambda * (w0lambda * v10 + w1lambda * v11)
        const float top = fmaf(T[k][j][3], ra[S][k][3], fmaf(T[k][j][2], ra[S][k][2], fmaf(T[k][j][1], ra[S][k][1], T[k][j][0] * ra[S][k][0])));
        const float bot = fmaf(T[k][j][3], rb[S][k][3], fmaf(T[k][j][2], rb[S][k][2], fmaf(T[k][j][1], rb[S][k][1], T[k][j][0] * rb[S][k][0])));
        v[j] = ly0[k] * top + ly1[k] * bot;
      }
    }
    *reinterpret_cast<f32x4v *>(buf + loff[k]) = v;
  };
  auto commit_weights = [&](auto set_, float *buf) {
    constexpr int S = decltype(set_)::value;
    if (threadIdx.x < NW / 4) *reinterpret_cast<f32x4v *>(buf + CK * SC + 4 * threadIdx.x) = wreg[S];
  };
  static_assert(NK <= CK, "one staging item after each of the first NK MFMA groups");
  // chunk s out of buffer s & 1 (register-set parity PAR = s & 1 held it; set 1 - PAR holds chunk s + 1)
  auto chunk = [&](int s, auto par_) {
    constexpr int PAR = decltype(par_)::value;
    using Mine = std::integral_constant<int, PAR>;
    using Other = std::integral_constant<int, 1 - PAR>;
    const float *cur = smem + PAR * BUF;
    float *nxt = smem + (1 - PAR) * BUF;
    issue(Mine{}, s + 2);   // set PAR was written to LDS during chunk s - 1: free
#pragma unroll
    for (int c = 0; c < CK; ++c) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float a = cur[CK * SC + (c * 3 + ky) * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma16(a, cur[base[t] + c * SC + ky * SY], acc[t]);
      }
      if (c < NK) commit_item(Other{}, c, s + 1, nxt);     // (chunk NCHUNK: zeros into the buffer nobody reads)
      if (c == CK - 1) commit_weights(Other{}, nxt);
    }
    __syncthreads();   // the other buffer is published, this one is free
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  issue(S0{}, 0);
  issue(S1{}, 1);
#pragma unroll
  for (int k = 0; k < NK; ++k) commit_item(S0{}, k, 0, smem);
  commit_weights(S0{}, smem);
  __syncthreads();
  for (int s = 0; s < Cfg::NCHUNK; s += 2) {
    chunk(s, S0{});
    chunk(s + 1, S1{});
  }
  static_assert(Cfg::NCHUNK % 2 == 0, "chunks are processed in pairs");

  // ---- epilogue: + bias class of the pixel; lane holds rows 4 kq + r = (co = 2 kq + (r >> 1), x phase r & 1) of column jcol ----
  const rsrc_t dst = make_rsrc(out + (size_t)n * 8 * hw, (size_t)8 * hw * 4);
  const rsrc_t dst2 = make_rsrc(out2 ? out2 + (size_t)n * 8 * hw : out, (size_t)8 * hw * 4);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ct = wave * NT + t, cx = ct & 1, cy = ct >> 1;
    const int oy = ty0 + cy, ox = tx0 + cx * 32 + 2 * jcol;
    const bool ok = oy < H && ox < W;   // W even: the pixel pair is inside or outside
    const int rcls = oy == 0 ? 0 : (oy == H - 1 ? 2 : 1);
    const int c0cls = ox == 0 ? 0 : 1, c1cls = ox + 1 == W - 1 ? 2 : 1;   // ox even: never the last column; ox + 1 odd: never the first
    float o[2][2];   // [x phase][channel 2 kq + h]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int co = 2 * kq + h;
      o[0][h] = acc[t][2 * h] + bias9[(rcls * 3 + c0cls) * 8 + co];
      o[1][h] = acc[t][2 * h + 1] + bias9[(rcls * 3 + c1cls) * 8 + co];
      buf_store2(f32x2{o[0][h], o[1][h]}, dst, ok ? (co * hw + oy * W + ox) * 4 : kOOB, 0);
    }
    if (out2) {   // pixel-major copy: channels (2 kq, 2 kq + 1) of pixels ox, ox + 1
      const int pbase = ((oy * W + ox) * 8 + 2 * kq) * 4;
      buf_store2(f32x2{o[0][0], o[0][1]}, dst2, ok ? pbase : kOOB, 0);
      buf_store2(f32x2{o[1][0], o[1][1]}, dst2, ok ? pbase + 32 : kOOB, 0);
    }
  }
}

}  // namespace

extern "C" int casmvs_fpn_tail0_supported(int H, int W) { return H >= 4 && W >= 8 && H % 2 == 0 && W % 4 == 0; }

extern "C" int casmvs_fpn_tail0_f32(const float *packed40, const float *bias9, const float *conv0, const float *feat1_sum,
                                    float *feat0, float *feat0_nhwc, int N, int H, int W, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed40 && bias9 && conv0 && feat1_sum && feat0, "fpn_tail0: null pointer");
  CASMVS_REQUIRE(N > 0 && N <= 65535 && casmvs_fpn_tail0_supported(H, W), "fpn_tail0: N=%d H=%d W=%d (H even, W %% 4 == 0, W >= 8)", N, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(conv0) | reinterpret_cast<size_t>(feat0) | reinterpret_cast<size_t>(packed40)) & 15) == 0 &&
                 (reinterpret_cast<size_t>(feat0_nhwc) & 15) == 0, "fpn_tail0: conv0 / feat0 / packed40 / feat0_nhwc must be 16-byte aligned");
  CASMVS_REQUIRE((size_t)32 * (H / 2) * (W / 2) < ((size_t)1 << 29) && (size_t)8 * H * W < ((size_t)1 << 29), "fpn_tail0: image too large");
  using Cfg = FpnCfg;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY);
  hipLaunchKernelGGL(fpn_tail0_kernel, dim3((unsigned)(tiles_x * tiles_y), (unsigned)N), dim3(kThreads), Cfg::LDS_BYTES, (hipStream_t)stream,
                     conv0, feat1_sum, packed40, bias9, feat0, feat0_nhwc, H, W, tiles_x);
  return casmvs::check_launch("fpn_tail0_kernel");
}
