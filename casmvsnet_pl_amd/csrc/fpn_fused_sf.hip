// FeatureNet's full-resolution FPN tail (fpn_fused.hip: feat0 = smooth0( lat0(conv0) + upsample2x(feat1') ) as ONE 3x3 convolution
// over 40 channels = [conv0 | up(feat1')] with host-composed weights and nine border bias classes) on the f16 matrix cores in the
// float32-grade split arithmetic of conv0_splitf16.hip.
//
// Reference semantics: models/mvsnet.py:36-38,50-51,54.
//
// Why: the float32 form costs 95 us of float32 MFMA + 100 us of interpolation per step at batch 2 and the two do not co-execute
// (the float32-input MFMA issues at the vector rate and blocks the VALU of its SIMD).  The f16 matrix instruction needs 3/16 of
// the matrix time and runs beside another wave's VALU work, so the kernel becomes its staging: interpolation (ATen's
// upsample_bilinear2d rule, horizontal tent then vertical blend, as fpn_fused.hip) + per-tile scaling + two-slice split.
//
// Formulation: PX form of conv0_splitf16.hip in 2D - rows = (8 output channels x 2 x-phases), K = 32 = 4 input x-offsets x 8
// channels of a chunk, 5 chunks (chunk 0: conv0's 8 channels, chunks 1-4: 8 upsampled channels each), 3 row taps.  Workgroup =
// 256 threads, output tile 20 x 32 pixels, wave w = rows 5 w .. 5 w + 4; halo tile 22 x 40 pixels x 2 slices x 16 B = 28 KiB
// (220 staging items on the 256 threads: one round at 86 % lane use; a 16-row tile used 70 %);
// the lane images of all 5 chunks (30 KiB) stay in LDS: persistent workgroups, tiles XCD-major.  One staging item = (row, 4 x)
// x 8 channels: 8 (direct) or 16 (two source rows) 16-byte loads, in flight during the previous chunk's matrix phase.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"
#include "split_f16.h"

#ifndef CASMVS_FS_PAIR
#define CASMVS_FS_PAIR 0
#endif
#ifndef CASMVS_FS_EARLY_PREFETCH
#define CASMVS_FS_EARLY_PREFETCH 0   // A/B builds: 1 = the next chunk's loads issued as soon as V exists, before both barriers (round 6: 448-467 us against 454: no gain, profiles/r06_costvol_v5_v7_unrolled_and_fpn_prefetch.txt)
#endif

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct FsCfg {
  static constexpr int THREADS = 256, NT = 5;
  static constexpr int TY = 4 * NT, TX = 32;
  static constexpr int IY = TY + 2, IX = TX + 8;                    // rows y0 - 1 .. y0 + 20, columns x0 - 4 .. x0 + 35
  static constexpr int ROW = IX + 1, NV = IY * ROW;                 // 16-byte slots per staged row / per slice: 902
  static __host__ __device__ constexpr int slot(int x) { return x ^ (((x >> 3) & 1) << 1); }   // as SfCfg::slot
  static constexpr int ITEMS = IY * (IX / 4);                       // 220: one round
  static_assert(ITEMS <= THREADS, "one staging round");
  static constexpr int CD = 8, CU = 32, NCH = (CD + CU) / 8;        // 5 chunks
  static constexpr int WUNITS = NCH * 3 * 2 * 64;                   // [chunk][ky][slice][lane]: 1920 16-byte units
  static constexpr int NWL = (WUNITS + THREADS - 1) / THREADS;      // 8
  static constexpr size_t ACT_BYTES = (size_t)2 * NV * 16, W_BYTES = (size_t)WUNITS * 16;
#ifndef CASMVS_FS_LDS_PAD
#define CASMVS_FS_LDS_PAD 0   // debug builds: extra dynamic LDS per workgroup (e.g. 50000: one workgroup per CU)
#endif
  static constexpr size_t LDS_BYTES = ACT_BYTES + W_BYTES + 16 + CASMVS_FS_LDS_PAD;     // 59 600
};

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void split8(const float (&x)[8], float mult, u32x4 (&o)[2]) { casmvs::split8_f16(x, mult, o); }   // split_f16.h

// c0 (N, 8, H, W), f1 (N, 32, H/2, W/2); wpk: [chunk][ky][slice][lane] 16-byte lane images, then unscale = 2^-kw (one float);
// bias9 (3, 3, 8): [row class][column class][co].  out (N, 8, H, W) or NULL; out2: NULL or (N, H, W, 8) pixel-major (at least one).
__global__ __launch_bounds__(FsCfg::THREADS, 2) void fpn_tail0_sf_kernel(const float *__restrict__ c0, const float *__restrict__ f1,
                                                                        const unsigned char *__restrict__ wpk, const float *__restrict__ bias9,
                                                                        float *__restrict__ out, float *__restrict__ out2, int N, int H, int W,
                                                                        int tiles_x, int tiles_y) {
  using Cfg = FsCfg;
  constexpr int NT = Cfg::NT, NWL = Cfg::NWL, IX = Cfg::IX, NV = Cfg::NV, NCH = Cfg::NCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                          // [2][NV]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                          // [chunk][ky][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, u = lane >> 4;
  const int total = tiles_x * tiles_y * N;
  if ((int)blockIdx.x >= total) return;
  const int hc = H >> 1, wc = W >> 1, hw = H * W, hwc = hc * wc;
  const float unscale = *reinterpret_cast<const float *>(wpk + Cfg::W_BYTES);
  // the lane images of all chunks: once per workgroup
  {
    const rsrc_t wsrc = make_rsrc(reinterpret_cast<const float *>(wpk), Cfg::W_BYTES);
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int unit = tid + i * Cfg::THREADS;
      const u32x4 v = __builtin_bit_cast(u32x4, buf_load4(wsrc, unit < Cfg::WUNITS ? unit * 16 : kOOB, 0));
      if (unit < Cfg::WUNITS) wl[unit] = v;
    }
  }
  const float sy = H > 1 ? (float)(hc - 1) / (float)(H - 1) : 0.0f;   // ATen: scale = (in - 1) / (out - 1) in float
  const float sx = W > 1 ? (float)(wc - 1) / (float)(W - 1) : 0.0f;
  const rsrc_t none = make_rsrc(c0, 0);

  // lane's B slot of staged row 0 of this wave's first output row (ky = 0): (NT wave) * ROW + slot(2 j + u + 3)
  const int vbase = NT * wave * Cfg::ROW + Cfg::slot(2 * jcol + u + 3);

  // staging plan of a tile: item e = tid -> (staged row, 4-x group).  Two parts: the integer part (what the chunk-0 prefetch of the
  // next tile needs) runs in the last chunk of the current tile; the floating-point part (the interpolation constants of chunks
  // 1..4) runs at the top of the tile, when this wave has no matrix instruction in flight.  (With the conversions and compares of
  // the tent matrix issued between the wave's own f16 MFMAs, staged values of lanes 48-63 came out wrong in ~1 of 500 tiles,
  // only with two workgroups per CU - tools/debug/fpn_sf_check.py)
  int voff_d, voff_u0, voff_u1, vox, vxor;
  float ly0, ly1;
  f32x2 T2[2][4];   // the horizontal tent of the item's pixel PAIRS (2 jp, 2 jp + 1): weight of window column m
  auto plan_int = [&](int ty0, int tx0) {
    const int e = tid;
    const int iy = e / (IX / 4), g = e - iy * (IX / 4);
    const int gy = ty0 - 1 + iy, gx = tx0 - 4 + 4 * g;
    const bool ok = e < Cfg::ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 4 == 0: a group is inside or outside
    vox = e < Cfg::ITEMS ? iy * Cfg::ROW + 4 * g : -1;
    vxor = ((g >> 1) & 1) << 1;
    voff_d = ok ? (gy * W + gx) * 4 : kOOB;
  };
  auto plan_fp = [&](int ty0, int tx0) {
    const int e = tid;
    const int iy = e / (IX / 4), g = e - iy * (IX / 4);
    const int gy = ty0 - 1 + iy, gx = tx0 - 4 + 4 * g;
    const bool ok = e < Cfg::ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
    // ATen upsample_bilinear2d, align_corners: src = dst * scale; i0 = (int)src; i1 = i0 + (i0 < in - 1); l1 = src - i0
    const float fy = sy * (float)(ok ? gy : 0);
    const int y0 = (int)fy, y1 = y0 + (y0 < hc - 1 ? 1 : 0);
    ly1 = fy - (float)y0;
    ly0 = 1.0f - ly1;
    int xb = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float fx = sx * (float)((ok ? gx : 0) + j);
      const int x0 = (int)fx, x1 = x0 + (x0 < wc - 1 ? 1 : 0);
      const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
      if (j == 0) xb = x0 < wc - 4 ? x0 : wc - 4;   // the 4-column window [xb, xb + 4) holds every column the 4 pixels use
#pragma unroll
      for (int m = 0; m < 4; ++m) T2[j >> 1][m][j & 1] = (m == x0 - xb ? lx0 : 0.0f) + (m == x1 - xb ? lx1 : 0.0f);
    }
    voff_u0 = ok ? (y0 * wc + xb) * 4 : kOOB;
    voff_u1 = ok ? (y1 * wc + xb) * 4 : kOOB;
  };
  f32x4v ra[8], rb[8];
  auto prefetch = [&](int n, int chunk, bool exists) {   // every load of (tile, chunk); nothing here waits
    const bool direct = chunk == 0;
    const rsrc_t r_a = !exists ? none : (direct ? make_rsrc(c0 + (size_t)n * Cfg::CD * hw, (size_t)Cfg::CD * hw * 4)
                                                : make_rsrc(f1 + (size_t)n * Cfg::CU * hwc, (size_t)Cfg::CU * hwc * 4));
    const rsrc_t r_b = (exists && !direct) ? r_a : none;
    const int cs = direct ? hw * 4 : hwc * 4, c00 = direct ? 0 : (chunk - 1) * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      ra[c] = buf_load4(r_a, direct ? voff_d : voff_u0, (c00 + c) * cs);   // upsampled: 4-byte aligned 16-byte loads
      rb[c] = buf_load4(r_b, voff_u1, (c00 + c) * cs);
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto decode = [&](int v, int &n, int &ty0, int &tx0) {
#if CASMVS_FS_PAIR   // A/B builds: the two workgroups of a CU on neighbouring tiles (buffer_ops.h: cu_pair_remap)
    v = cu_pair_remap(v, total);
#endif
    int item = xcd_major(v, total);   // x fastest, then y, then image
    tx0 = (item % tiles_x) * Cfg::TX;
    item /= tiles_x;
    ty0 = (item % tiles_y) * Cfg::TY;
    n = item / tiles_y;
  };
  int item = blockIdx.x, n, ty0, tx0;
  decode(item, n, ty0, tx0);
  plan_int(ty0, tx0);
  prefetch(n, 0, true);
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    int nn = n, nty0 = ty0, ntx0 = tx0;
    if (have_next) decode(next_item, nn, nty0, ntx0);
    plan_fp(ty0, tx0);
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- interpolate (chunks 1..4), find the staged tile's largest magnitude ----
      // Two x per register pair (round 6): a staged value lives in the pair it was LOADED in (x, x + 1 of one channel), the vertical blend, the tent and the
      // scaling are packed float32 instructions along x.  (The scalar form was vectorised by the compiler along CHANNEL pairs - the pairs the split's
      // conversions take - at the price of 82 register moves per chunk, a fifth of the chunk's vector instructions.)  Same operations, same order per value.
      f32x2 V2[8][2];
#ifndef CASMVS_FS_DEBUG
#define CASMVS_FS_DEBUG 0   // debug builds (WRONG results): 1 = no interpolation (every chunk staged like chunk 0), 2 = the next tile's plan from the
#endif                      // first tile's (no per-tile plan arithmetic), 4 = no DPP reduction (every tile scaled by 2^0)
      if (ch == 0 || (CASMVS_FS_DEBUG & 1)) {   // conv0's channels as they are
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          V2[c][0] = f32x2{ra[c][0], ra[c][1]};
          V2[c][1] = f32x2{ra[c][2], ra[c][3]};
        }
      } else {
        // bilinear 2x, align_corners: the vertical blend of the two source rows first (4 window columns), then the horizontal tent
        // (two non-zero weights per pixel; the exact zeros of T add nothing).  ATen blends horizontally first: the two orders differ
        // by float32 roundings of the interpolated value - below this kernel's own 2^-22 slice error.
        const f32x2 l0{ly0, ly0}, l1{ly1, ly1};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const f32x2 w01 = __builtin_elementwise_fma(l1, f32x2{rb[c][0], rb[c][1]}, l0 * f32x2{ra[c][0], ra[c][1]});
          const f32x2 w23 = __builtin_elementwise_fma(l1, f32x2{rb[c][2], rb[c][3]}, l0 * f32x2{ra[c][2], ra[c][3]});
          const f32x2 w0{w01.x, w01.x}, w1{w01.y, w01.y}, w2{w23.x, w23.x}, w3{w23.y, w23.y};
#pragma unroll
          for (int jp = 0; jp < 2; ++jp)
            V2[c][jp] = __builtin_elementwise_fma(T2[jp][3], w3, __builtin_elementwise_fma(T2[jp][2], w2, __builtin_elementwise_fma(T2[jp][1], w1, T2[jp][0] * w0)));
        }
      }
#if CASMVS_FS_EARLY_PREFETCH
      // the loads of the NEXT chunk (or of the next tile's chunk 0) go out here - their registers are free as soon as V exists - instead of behind the
      // second barrier: they then have the reduction, both barriers, the split and the matrix phase to land, not the matrix phase alone (SQ counters,
      // round 5: this kernel's waves were parked 35 % of their time; one chunk's matrix phase is ~400 cycles, a memory round trip 2 000+)
      int nvox = vox, nvxor = vxor;
      if (ch + 1 < NCH) {
        prefetch(n, ch + 1, true);
      } else {
        const int cvox = vox, cvxor = vxor;
        if (!(CASMVS_FS_DEBUG & 2)) plan_int(nty0, ntx0);   // (rewrites voff_d / vox / vxor: the current tile's LDS slots are kept for the stores below)
        nvox = vox;
        nvxor = vxor;
        vox = cvox;
        vxor = cvxor;
        prefetch(nn, 0, have_next);
      }
#endif
      float m = 0.0f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(V2[c][j >> 1][j & 1]));
      const unsigned wm = (CASMVS_FS_DEBUG & 4) ? 0x47000000u : casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
      if (lane == 0) wmax[wave] = wm;
      __syncthreads();   // every wave is done with the previous chunk's LDS; the four maxima (and, first time, the lane images) are visible
      float mult, inv;   // max |x| 2^kx in [2^14, 2^15); 2^-kx
      casmvs::tile_scale(wmax, mult, inv);
      if (vox >= 0) {
        const f32x2 mult2{mult, mult};
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) V2[c][jp] = V2[c][jp] * mult2;   // exact (a power of two); the split takes the scaled values
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] = V2[c][j >> 1][j & 1];
          u32x4 o[2];
          casmvs::split8_scaled_f16(x, o);
#pragma unroll
          for (int s = 0; s < 2; ++s) act[s * NV + vox + (j ^ vxor)] = o[s];
        }
      }
      __syncthreads();
#if CASMVS_FS_EARLY_PREFETCH
      vox = nvox;
      vxor = nvxor;
#else
      if (ch + 1 < NCH) {
        prefetch(n, ch + 1, true);
      } else {
        if (!(CASMVS_FS_DEBUG & 2)) plan_int(nty0, ntx0);
        prefetch(nn, 0, have_next);
      }
#endif
      // ---- matrix phase: the wave's six staged rows read once, 3 ky x 4 output rows x 3 partial products ----
      f32x4 part[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) part[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x4 row[NT + 2][2];
#pragma unroll
      for (int yr = 0; yr < NT + 2; ++yr)
#pragma unroll
        for (int s = 0; s < 2; ++s) row[yr][s] = act[s * NV + vbase + yr * Cfg::ROW];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        u32x4 a[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) a[s] = wl[((ch * 3 + ky) * 2 + s) * 64 + lane];
        constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int t = 0; t < NT; ++t) part[t] = mfma_f16(a[PA[p]], row[t + ky][PB[p]], part[t]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = fmaf(part[t][q], inv, acc[t][q]);
    }
    // ---- epilogue: 2^-kw, + bias class of the pixel; lane holds rows 4 u + r = (co = 2 u + (r >> 1), x phase r & 1) of column j ----
    const rsrc_t dst = out ? make_rsrc(out + (size_t)n * 8 * hw, (size_t)8 * hw * 4) : make_rsrc(out2, 0);   // out == NULL: an empty range drops the stores
    const rsrc_t dst2 = make_rsrc(out2 ? out2 + (size_t)n * 8 * hw : out, (size_t)8 * hw * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int oy = ty0 + NT * wave + t, ox = tx0 + 2 * jcol;
      const bool ok = oy < H && ox < W;   // W even: the pixel pair is inside or outside
      const int rcls = oy == 0 ? 0 : (oy == H - 1 ? 2 : 1);
      const int c0cls = ox == 0 ? 0 : 1, c1cls = ox + 1 == W - 1 ? 2 : 1;   // ox even: never the last column; ox + 1 odd: never the first
      float o[2][2];   // [x phase][channel 2 u + h]
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int co = 2 * u + h;
        o[0][h] = fmaf(acc[t][2 * h], unscale, bias9[(rcls * 3 + c0cls) * 8 + co]);
        o[1][h] = fmaf(acc[t][2 * h + 1], unscale, bias9[(rcls * 3 + c1cls) * 8 + co]);
        buf_store2(f32x2{o[0][h], o[1][h]}, dst, ok ? (co * hw + oy * W + ox) * 4 : kOOB, 0);
      }
      if (out2) {   // pixel-major copy: channels (2 u, 2 u + 1) of pixels ox, ox + 1
        const int pbase = ((oy * W + ox) * 8 + 2 * u) * 4;
        buf_store2(f32x2{o[0][0], o[0][1]}, dst2, ok ? pbase : kOOB, 0);
        buf_store2(f32x2{o[1][0], o[1][1]}, dst2, ok ? pbase + 32 : kOOB, 0);
      }
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!have_next) break;
    item = next_item;
    n = nn;
    ty0 = nty0;
    tx0 = ntx0;
  }
}

inline uint16_t f16_bits_fs(float x) {
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

}  // namespace

extern "C" size_t casmvs_fpn_tail0_splitf16_packed_bytes(void) { return FsCfg::W_BYTES + 16; }

// HOST-side packing of the composed 40-channel 3x3 layer (mvsnet.compose_fpn_tail): weight (8, 40, 3, 3) float32 ->
// w' = 2^kw w (max |w'| in [2^13, 2^14)); per chunk of 8 input channels, per ky, per slice (f16(w'), f16(w' - f16(w'))), per lane the 8
// float16 values A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[co = i >> 1][chunk * 8 + e][ky][kx = (lane >> 4) - (i & 1)]); then 2^-kw.
extern "C" int casmvs_fpn_tail0_splitf16_pack(const float *weight40, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight40 && packed, "fpn_tail0_splitf16_pack: null pointer");
  float wmax = 0.0f;
  for (int i = 0; i < 8 * 40 * 9; ++i) {
    CASMVS_REQUIRE(std::isfinite(weight40[i]), "fpn_tail0_splitf16_pack: weight %d is not finite", i);
    wmax = std::fmax(wmax, std::fabs(weight40[i]));
  }
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
  const int kw = 14 - ex;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int ch = 0; ch < FsCfg::NCH; ++ch)
    for (int ky = 0; ky < 3; ++ky) {
      uint16_t img[2][64][8];
      for (int l = 0; l < 64; ++l) {
        const int i = l & 15, co = i >> 1, s = i & 1, uu = l >> 4, kx = uu - s;
        for (int e = 0; e < 8; ++e) {
          const float w = (kx >= 0 && kx <= 2) ? std::ldexp(weight40[((co * 40 + ch * 8 + e) * 3 + ky) * 3 + kx], kw) : 0.0f;
          const float a = (float)(_Float16)w;
          img[0][l][e] = f16_bits_fs(w);
          img[1][l][e] = f16_bits_fs(w - a);
        }
      }
      std::memcpy(p, img, sizeof(img));
      p += 2 * 64 * 8;
    }
  float *tail = reinterpret_cast<float *>(p);
  tail[0] = std::ldexp(1.0f, -kw);
  tail[1] = tail[2] = tail[3] = 0.0f;
  return CASMVS_OK;
}

extern "C" int casmvs_fpn_tail0_splitf16_f32(const void *packed, const float *bias9, const float *conv0, const float *feat1_sum,
                                             float *feat0, float *feat0_nhwc, int N, int H, int W, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && bias9 && conv0 && feat1_sum && (feat0 || feat0_nhwc), "fpn_tail0_splitf16: null pointer (feat0 may be NULL with feat0_nhwc given)");
  CASMVS_REQUIRE(N > 0 && N <= 65535 && H >= 4 && W >= 8 && H % 2 == 0 && W % 4 == 0, "fpn_tail0_splitf16: N=%d H=%d W=%d (H even, W %% 4 == 0, W >= 8)", N, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(conv0) | reinterpret_cast<size_t>(feat0) | reinterpret_cast<size_t>(packed)) & 15) == 0 &&
                 (reinterpret_cast<size_t>(feat0_nhwc) & 15) == 0, "fpn_tail0_splitf16: conv0 / feat0 / packed / feat0_nhwc must be 16-byte aligned");
  CASMVS_REQUIRE((size_t)32 * (H / 2) * (W / 2) < ((size_t)1 << 29) && (size_t)8 * H * W < ((size_t)1 << 29), "fpn_tail0_splitf16: image too large");
  using Cfg = FsCfg;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY);
  const long total = (long)tiles_x * tiles_y * N;
  CASMVS_REQUIRE(total < (1L << 31), "fpn_tail0_splitf16: too many tiles");
  auto kernel = fpn_tail0_sf_kernel;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "fpn_tail0_sf_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, (hipStream_t)stream, conv0,
                     feat1_sum, reinterpret_cast<const unsigned char *>(packed), bias9, feat0, feat0_nhwc, N, H, W, tiles_x, tiles_y);
  return casmvs::check_launch("fpn_tail0_sf_kernel");
}
