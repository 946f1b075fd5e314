// FeatureNet's 3x3 stride-1 layers with 16 / 32 channels (conv1.1 / conv1.2: 16 -> 16, conv2.1 / conv2.2: 32 -> 32: ConvBnReLU,
// models/modules.py:8-18, models/mvsnet.py:19-20,24-25; smooth1: Conv2d 32 -> 16 with bias, mvsnet.py:32,53, which also writes the
// pixel-major copy of its output) on the f16 matrix cores in the float32-grade split arithmetic
// of conv0_splitf16.hip: the 2D sibling of conv_ci_splitf16.hip.
//
// Formulation: D[16 x 16] += A[16 x 32] B[32 x 16]; rows = 16 output channels (C / 16 row blocks share every B operand), columns = 16
// consecutive output x of one row, K = 32 = two TAPS x 16 input channels: 5 steps for the 9 taps (tap t = ky * 3 + kx; the 10th is
// zero weights).  Lane (j, kb) reads the 8 channels 8 (kb & 1) .. of tap 2 m + (kb >> 1) at column j: one 16-byte LDS read from the
// planes [slice][channel half][row][x]; the upper lane half reads one x or one row further (two per-lane base sets).
//
// Workgroup = 256 threads, output tile 16 x 16 pixels, wave w = rows 4 w .. 4 w + 3; per chunk of 16 input channels the halo tile
// 18 x 20 pixels (x0 - 2 .. x0 + 17: 8-byte aligned pairs) x 2 slices x 32 B = 23 KiB (planes padded to a multiple of 256 B: both
// channel halves start on the same bank); the lane images of ALL chunks (10 KiB / 40 KiB) stay in LDS: persistent workgroups, tiles
// XCD-major.  180 staging items on 256 threads: one round.  Only integer address arithmetic sits between the matrix instructions
// (DESIGN.md 2.0).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"
#include "split_f16.h"

#ifndef CASMVS_C2_PAIR
#define CASMVS_C2_PAIR 0
#endif

#ifndef CASMVS_CI_SWP
#define CASMVS_CI_SWP 1   // A/B builds: 0 = the two voxels of a staging item written in plain order (2-way conflicted 16-byte writes, no selects)
#endif

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int CIN, int COUT>
struct C2Cfg {
  static constexpr int THREADS = 256, NT = 4;
  static constexpr int TY = 16, TX = 16;
  static constexpr int IY = TY + 2, IX = TX + 4;                     // rows y0 - 1 .. y0 + 16, columns x0 - 2 .. x0 + 17
  static constexpr int RS = IX;                                      // 16-byte units per staged row of one plane
  static constexpr int NVOX = ((IY * RS + 15) / 16) * 16;            // units per plane, padded to 256 B: 368
  static constexpr int RB = COUT / 16, NCH = CIN / 16, STEPS = 5;
  static constexpr int ITEMS = IY * (IX / 2);                        // 180
  static constexpr int WUNITS = NCH * STEPS * RB * 2 * 64;           // [chunk][step][row block][slice][lane]: 640 / 2560 16-byte units
  static constexpr int NWL = (WUNITS + THREADS - 1) / THREADS;       // 3 / 10
  static constexpr size_t ACT_BYTES = (size_t)4 * NVOX * 16, W_BYTES = (size_t)WUNITS * 16;
  // At most TWO workgroups per CU (= two waves per SIMD): the requested LDS is padded to 56 KiB.  A SIMD that holds one staging wave
  // (packed float32 arithmetic) and TWO waves with f16 matrix instructions in flight is the configuration in which float32 results
  // came out wrong on this GPU; with two waves per SIMD, and no floating-point work between a wave's own matrix
  // instructions, a staging wave never meets more than one stream of them.
  static constexpr size_t LDS_USED = ACT_BYTES + W_BYTES + 16;       // 33 808 / 43 936 / 64 528
  static constexpr size_t LDS_BYTES = LDS_USED < CASMVS_SF_LDS_FLOOR ? CASMVS_SF_LDS_FLOOR : LDS_USED;
};

__device__ __forceinline__ f32x4 mfma_f16_c2(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// in (N, CIN, H, W) float32, W % 2 == 0, 8-byte aligned; wpk: [chunk][step][row block][slice][lane] 16-byte lane images, then scale[COUT]
// (ABN scale x 2^-kw), shift[COUT]; out (N, COUT, H, W) or NULL; out2: NULL or (N, H, W, COUT) pixel-major (16-byte aligned); at least one.
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv2d_ci_sf_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                             float *__restrict__ out, float *__restrict__ out2, int N, int H, int W, int tiles_x,
                                                             int tiles_y, float slope) {
  using Cfg = C2Cfg<CIN, COUT>;
  constexpr int NCH = Cfg::NCH, RB = Cfg::RB, NT = Cfg::NT, NWL = Cfg::NWL, IX = Cfg::IX, NVOX = Cfg::NVOX, RS = Cfg::RS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                          // [slice][half][NVOX]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                          // [chunk][step][rb][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, kb = lane >> 4, half = kb & 1, hi_tap = kb >> 1;
  const int total = tiles_x * tiles_y * N;
  if ((int)blockIdx.x >= total) return;
  const int hw = H * W;
  const size_t ss = (size_t)CIN * hw, oss = (size_t)COUT * hw;
  {   // the lane images of all chunks: once per workgroup
    const rsrc_t wsrc = make_rsrc(reinterpret_cast<const float *>(wpk), Cfg::W_BYTES);
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int unit = tid + i * Cfg::THREADS;
      const u32x4 v = __builtin_bit_cast(u32x4, buf_load4(wsrc, unit < Cfg::WUNITS ? unit * 16 : kOOB, 0));
      if (unit < Cfg::WUNITS) wl[unit] = v;
    }
  }
  const float *tail = reinterpret_cast<const float *>(wpk + Cfg::W_BYTES);
  float sc[RB][4], sh[RB][4];   // lane holds rows 4 kb + r of every row block
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[rb][r] = tail[rb * 16 + 4 * kb + r];
      sh[rb][r] = tail[COUT + rb * 16 + 4 * kb + r];
    }
  const rsrc_t none = make_rsrc(in, 0);

  // lane's B unit (slice 0) of output row t of this wave at tap 0: plane `half`, voxel (4 wave + t, j + 1); the upper lane half
  // (taps 2 m + 1) adds the step's tap distance: next x or next row
  int vbx[NT], vby[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int v = half * NVOX + (4 * wave + t) * RS + jcol + 1;
    vbx[t] = v + hi_tap * 1;
    vby[t] = v + hi_tap * (RS - 2);
  }
  // staging plan (integer arithmetic only): item e = tid -> (staged row, pair of x)
  int voff, vox;
  auto plan = [&](int ty0, int tx0) {
    const int e = tid;
    const int iy = e / (IX / 2), g = e - iy * (IX / 2);
    const int gy = ty0 - 1 + iy, gx = tx0 - 2 + 2 * g;
    const bool ok = e < Cfg::ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 2 == 0
    voff = ok ? (gy * W + gx) * 4 : kOOB;
    vox = e < Cfg::ITEMS ? iy * RS + 2 * g : -1;
  };
  f32x2 R[16];
  auto prefetch = [&](int n, int chunk, bool exists) {
    const rsrc_t src = exists ? make_rsrc(in + (size_t)n * ss, ss * 4) : none;
#pragma unroll
    for (int c = 0; c < 16; ++c) R[c] = buf_load2(src, voff, (chunk * 16 + c) * hw * 4);
  };
  auto decode = [&](int v, int &n, int &ty0, int &tx0) {
#if CASMVS_C2_PAIR   // A/B builds: the two workgroups of a CU on neighbouring tiles (buffer_ops.h: cu_pair_remap)
    v = cu_pair_remap(v, total);
#endif
    int item = xcd_major(v, total);   // x fastest, then y, then image
    tx0 = (item % tiles_x) * Cfg::TX;
    item /= tiles_x;
    ty0 = (item % tiles_y) * Cfg::TY;
    n = item / tiles_y;
  };

  f32x4 acc[NT][RB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int swp = CASMVS_CI_SWP ? (lane >> 2) & 1 : 0;   // write order of an item's two voxels (conv_ci_splitf16.hip): conflict-free staging writes

  int item = blockIdx.x, n, ty0, tx0;
  decode(item, n, ty0, tx0);
  plan(ty0, tx0);
  prefetch(n, 0, true);
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    int nn = n, nty0 = ty0, ntx0 = tx0;
    if (have_next) decode(next_item, nn, nty0, ntx0);
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      float m = 0.0f;
#pragma unroll
      for (int c = 0; c < 16; ++c) m = casmvs::absmax3(m, R[c][0], R[c][1]);
      const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
      if (lane == 0) wmax[wave] = wm;
      __syncthreads();   // every wave is done with the previous chunk's LDS; the four maxima (first time: the lane images) are visible
      float mult, inv;   // max |x| 2^kx in [2^14, 2^15); 2^-kx
      casmvs::tile_scale(wmax, mult, inv);
      if (vox >= 0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          u32x4 o[2][2];   // [voxel of the pair][slice]
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            float x[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = R[hf * 8 + c][p];
            casmvs::split8_f16(x, mult, o[p]);
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            u32x4 first_v, second_v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              first_v[q] = swp ? o[1][s][q] : o[0][s][q];
              second_v[q] = swp ? o[0][s][q] : o[1][s][q];
            }
            u32x4 *pl = act + (s * 2 + hf) * NVOX + vox;
            pl[swp] = first_v;
            pl[1 - swp] = second_v;
          }
        }
      }
      __syncthreads();
      if (ch + 1 < NCH) {
        prefetch(n, ch + 1, true);
      } else {
        plan(nty0, ntx0);
        prefetch(nn, 0, have_next);
      }
      // ---- matrix phase: 5 steps (tap pairs) x 4 output rows x RB row blocks x 3 partial products ----
      f32x4 part[NT][RB];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) part[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < Cfg::STEPS; ++st) {
        const int t0 = 2 * st, ky0 = t0 / 3, kx0 = t0 % 3;
        const int off = ky0 * RS + kx0;
        u32x4 bv[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int base = (kx0 != 2 || t0 == 8) ? vbx[t] : vby[t];   // the 10th tap (zero weights) reads the next x: staged, finite data
#pragma unroll
          for (int s = 0; s < 2; ++s) bv[t][s] = act[s * 2 * NVOX + base + off];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          u32x4 a[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) a[s] = wl[(((ch * Cfg::STEPS + st) * RB + rb) * 2 + s) * 64 + lane];
          constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t) part[t][rb] = mfma_f16_c2(a[PA[p]], bv[t][PB[p]], part[t][rb]);
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[t][rb][q] = NCH > 1 ? fmaf(part[t][rb][q], inv, acc[t][rb][q]) : part[t][rb][q] * inv;
    }
    // ---- epilogue: y = lrelu(acc * scale + shift); lane holds rows 4 kb + r (output channel 16 rb + 4 kb + r), column j ----
    const rsrc_t dst = out ? make_rsrc(out + (size_t)n * oss, oss * 4) : make_rsrc(out2, 0);   // out == NULL: an empty range drops the stores
    const rsrc_t dst2 = make_rsrc(out2 ? out2 + (size_t)n * oss : out, oss * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int oy = ty0 + 4 * wave + t, ox = tx0 + jcol;
      const bool ok = oy < H && ox < W;
      const int o0 = ok ? (4 * kb * hw + oy * W + ox) * 4 : kOOB;   // the lane's first channel row is part of the lane offset
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        f32x4v o4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[t][rb][r], sc[rb][r], sh[rb][r]);
          v = v > 0.0f ? v : v * slope;
          o4[r] = v;
          buf_store(v, dst, o0, (rb * 16 + r) * hw * 4);
        }
        if (out2)   // pixel-major copy: this lane's 4 consecutive channels of pixel (oy, ox) in one store
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), dst2, ok ? ((oy * W + ox) * COUT + 4 * kb) * 4 : kOOB, rb * 64, 0);
        acc[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (!have_next) break;
    item = next_item;
    n = nn;
    ty0 = nty0;
    tx0 = ntx0;
  }
}

inline uint16_t f16_bits_c2(float x) {
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

template <int CIN, int COUT>
int launch_c2(const void *packed, const float *in, float *out, float *out2, int N, int H, int W, float slope, hipStream_t st) {
  using Cfg = C2Cfg<CIN, COUT>;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY);
  const long total = (long)tiles_x * tiles_y * N;
  CASMVS_REQUIRE(total < (1L << 31), "conv2d_ci_splitf16_forward: too many tiles");
  auto kernel = conv2d_ci_sf_kernel<CIN, COUT>;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "conv2d_ci_sf_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, in,
                     reinterpret_cast<const unsigned char *>(packed), out, out2, N, H, W, tiles_x, tiles_y, slope);
  return casmvs::check_launch("conv2d_ci_sf_kernel");
}

}  // namespace

namespace {
inline bool c2_shape_ok(int cin, int cout) { return (cin == 16 && cout == 16) || (cin == 32 && cout == 32) || (cin == 32 && cout == 16); }
}  // namespace

extern "C" size_t casmvs_conv2d_ci_splitf16_packed_bytes(int cin, int cout) {
  if (!c2_shape_ok(cin, cout)) return 0;
  return (size_t)(cin / 16) * 5 * (cout / 16) * 2 * 64 * 16 + (size_t)2 * cout * sizeof(float);
}

// HOST-side packing: weight (cout, cin, 3, 3) float32 -> w' = 2^kw w (max |w'| in [2^13, 2^14)); per chunk of 16 input channels, per step m
// (taps 2 m, 2 m + 1), per row block, per slice, per lane the 8 float16 values
// A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[co = 16 rb + i][ci = 16 chunk + 8 ((lane >> 4) & 1) + e][tap 2 m + (lane >> 5)]), zero for tap 9;
// then scale[cout] * 2^-kw, shift[cout].
extern "C" int casmvs_conv2d_ci_splitf16_pack(int cin, int cout, const float *weight, const float *scale, const float *shift, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "conv2d_ci_splitf16_pack: null pointer");
  CASMVS_REQUIRE(c2_shape_ok(cin, cout), "conv2d_ci_splitf16_pack: cin=%d cout=%d (16 -> 16, 32 -> 32 or 32 -> 16)", cin, cout);
  float wmax = 0.0f;
  for (size_t i = 0; i < (size_t)cout * cin * 9; ++i) {
    CASMVS_REQUIRE(std::isfinite(weight[i]), "conv2d_ci_splitf16_pack: weight %zu is not finite", i);
    wmax = std::fmax(wmax, std::fabs(weight[i]));
  }
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
  const int kw = 14 - ex;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int ch = 0; ch < cin / 16; ++ch)
    for (int st = 0; st < 5; ++st)
      for (int rb = 0; rb < cout / 16; ++rb) {
        uint16_t img[2][64][8];
        for (int l = 0; l < 64; ++l) {
          const int i = l & 15, kb = l >> 4, tap = 2 * st + (kb >> 1), co = 16 * rb + i;
          for (int e = 0; e < 8; ++e) {
            const int ci = 16 * ch + 8 * (kb & 1) + e;
            const float w = tap < 9 ? std::ldexp(weight[((size_t)co * cin + ci) * 9 + tap], kw) : 0.0f;
            const float a = (float)(_Float16)w;
            img[0][l][e] = f16_bits_c2(w);
            img[1][l][e] = f16_bits_c2(w - a);
          }
        }
        std::memcpy(p, img, sizeof(img));
        p += 2 * 64 * 8;
      }
  float *tail = reinterpret_cast<float *>(p);
  for (int k = 0; k < cout; ++k) tail[k] = std::ldexp(scale ? scale[k] : 1.0f, -kw);
  for (int k = 0; k < cout; ++k) tail[cout + k] = shift ? shift[k] : 0.0f;
  return CASMVS_OK;
}

extern "C" int casmvs_conv2d_ci_splitf16_supported(int cin, int cout, int W) { return c2_shape_ok(cin, cout) && W % 2 == 0 && W >= 2; }

extern "C" int casmvs_conv2d_ci_splitf16_forward_f32(const void *packed, const float *in, float *out, float *out_nhwc, int N, int cin, int cout, int H,
                                                     int W, float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && (out || out_nhwc), "conv2d_ci_splitf16_forward: null pointer (out may be NULL with out_nhwc given)");
  CASMVS_REQUIRE(N > 0 && H > 0 && W > 0 && casmvs_conv2d_ci_splitf16_supported(cin, cout, W), "conv2d_ci_splitf16_forward: N=%d cin=%d cout=%d H=%d W=%d", N, cin, cout, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out)) & 7) == 0 &&
                 ((reinterpret_cast<size_t>(packed) | reinterpret_cast<size_t>(out_nhwc)) & 15) == 0,
                 "conv2d_ci_splitf16_forward: 8-byte aligned tensors, 16-byte aligned image / pixel-major output");
  CASMVS_REQUIRE((size_t)cin * H * W < ((size_t)1 << 29) && (size_t)cout * H * W < ((size_t)1 << 29), "conv2d_ci_splitf16_forward: one image's tensor must hold < 2^29 floats");
  hipStream_t st = (hipStream_t)stream;
  if (cin == 16) return launch_c2<16, 16>(packed, in, out, out_nhwc, N, H, W, slope, st);
  if (cout == 32) return launch_c2<32, 32>(packed, in, out, out_nhwc, N, H, W, slope, st);
  return launch_c2<32, 16>(packed, in, out, out_nhwc, N, H, W, slope, st);
}
