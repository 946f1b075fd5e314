// CostRegNet.conv0 (Conv3d Cin -> 8, k3 s1 p1 + folded ABN + leaky-relu) on the f16 matrix cores with float32-grade
// arithmetic: every float32 operand as the sum of TWO float16 numbers (22 of its 24 significand bits), three partial
// products accumulated in float32.  The sibling of conv0_splitbf16.hip (three bf16 slices, six products): half the matrix
// instructions and 2/3 of the LDS bytes per staged voxel, which lets TWO workgroups share a CU - one splits and stages its
// next tile while the other one multiplies (the bf16 kernel's 146 KB tile leaves room for one workgroup, whose load /
// split / multiply phases the ablations showed back to back: 245 us against an 87 us matrix floor).
//
// Reference semantics: models/mvsnet.py:63,91 (`conv0`), models/modules.py:21-31 (ConvBnReLU3D).
//
// Arithmetic.  float16 has an 11-bit significand and a narrow exponent, so both operands are first scaled by exact powers
// of two into the top of its range:
//   weights (host, once):  w' = 2^kw w with max |w'| in [2^13, 2^14);  w_a = f16(w'), w_b = f16(w' - w_a)
//   staged tile (device):  x' = 2^kx x with max over the (tile, chunk of 8 channels) |x'| in [2^14, 2^15);
//                          x_a = f16(x'), x_b = f16(x' - x_a)                   (round to nearest; x' - x_a is exact)
//   x' w' = x_a w_a + x_a w_b + x_b w_a + [x_b w_b + (x' - x_a - x_b) w' + ...]
// Every f16 x f16 product is exact in float32; the bracket is <= 3 * 2^-22 |x w| - the size of a few float32 roundings,
// of which a 216..864-term float32 dot product holds hundreds.  An element 2^-18 below its tile's maximum starts to lose
// bits of x_b to the float16 subnormal range, with an ABSOLUTE error <= 2^-40 of the tile maximum - far below what float32
// accumulation of the neighbouring large products leaves.  Measured against a float64 convolution the result is as close
// as the float32-MFMA kernel's (tests/test_gpu_parity.py).  The matrix-unit result of a chunk is multiplied by 2^-kx
// (exact) and summed over the chunks in float32; 2^-kw is folded into the ABN scale.
//
// Formulation: as conv0_splitbf16.hip (PX form, K = 4 x-offsets x 8 input channels, rows = (co, x phase)); workgroup =
// 256 threads (4 waves), output tile 4 x 4 x 32 voxels, wave w = z plane w, 4 (y) column tiles; input halo tile
// 6 x 6 x 40 voxels x 2 slices x 16 B = 46 KiB + 18 KiB of lane images; for a fixed kz a wave reads the six staged rows
// once and uses each for up to three (ky, y) pairs.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"
#include "split_f16.h"

#ifndef CASMVS_SF_PRIO
#define CASMVS_SF_PRIO 0   // A/B builds: 1 = raised wave priority during the matrix phase (tools/native/conv0_ab.cpp)
#endif
#ifndef CASMVS_SF_ORDER
#define CASMVS_SF_ORDER (-1)  // A/B builds force one tile order for every shape (0 .. 3, see sf_decode); -1: the measured choice per CIN
#endif
#ifndef CASMVS_SF_ABL
#define CASMVS_SF_ABL 0   // profiling builds only (WRONG results): 1 no MFMAs, 2 no split / LDS staging writes, 4 no global loads, 8 no tap reads
#endif

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct SfCfg {
  static constexpr int THREADS = 256, WAVES = 4, NT = 4;
  static constexpr int TZ = 4, TY = 4, TX = 32;
  static constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 8;     // x0 - 4 .. x0 + 35 (16-byte aligned global groups)
  static constexpr int ROW = IX + 1;                                // 16-byte slots per staged row (odd: rows rotate through the banks)
  static constexpr int NV = IZ * IY * ROW;                          // slots per slice: 1476
  static __host__ __device__ constexpr int slot(int x) { return x ^ (((x >> 3) & 1) << 1); }   // as SbCfg::slot
  static constexpr int ITEMS = IZ * IY * (IX / 4);                  // (z, y, group of 4 x) staging items: 360
  static constexpr int NR = (ITEMS + THREADS - 1) / THREADS;        // staging rounds per thread: 2
  static constexpr int WUNITS = 9 * 2 * 64;                         // 16-byte units of a chunk's lane images: [kz * 3 + ky][slice][lane]
  static constexpr int NWL = (WUNITS + THREADS - 1) / THREADS;      // 5
  static constexpr size_t ACT_BYTES = (size_t)2 * NV * 16, W_BYTES = (size_t)WUNITS * 16;
  static constexpr size_t LDS_BYTES = ACT_BYTES + W_BYTES + 16;     // + the four waves' tile maxima: 65 680 (two workgroups per CU)
};

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// 8 channels of one voxel, scaled by the tile's power of two -> the two 16-byte float16 vectors: split_f16.h
__device__ __forceinline__ void split_voxel_f16(const float (&x)[8], float mult, u32x4 (&o)[2]) { casmvs::split8_f16(x, mult, o); }

struct SfTile {
  int tx0, ty0, tz0, b;
};
// Work order of the persistent workgroups (results do not depend on it).  Items are dealt XCD-major (buffer_ops.h), so the
// workgroups resident on an XCD at one time walk neighbouring tiles and meet each other's halo lines in that XCD's L2.
//   ORDER 0: z fastest, then x, then y (round 3's first form)      1: x fastest, then z, then y
//   ORDER 2: as 1, and the two workgroups a CU holds (dispatch slots idx and idx + 32 of an XCD) take NEIGHBOURING items
//   ORDER 3: x, then y, then z
// Measured with tools/native/conv0_ab.cpp (batch 2, dirtied caches, bit-identical outputs; profiles/r03_conv0_tile_order_ab.txt):
// against order 0, order 1 is +2.6 % at every level, order 2 +4.7 % / 0 / +6.4 % at cin = 32 / 16 / 8, order 3 +2 / -2.6 / 0.
// (round 4: a tile grid shifted by 4 voxels in x - two 128-byte lines per staged row instead of three - was measured on the MI355X and removed:
// equal at cin 8 / 16, 10 % slower at cin 32, the output rows then straddle two lines; profiles/r04_native_checks_first_run.txt)
template <int ORDER>
__device__ __forceinline__ SfTile sf_decode(int v, int total, int tiles_x, int tiles_y, int tiles_z) {
  if (ORDER == 2) {
    const int xcd = v & 7, idx = v >> 3;
    if ((idx | 63) < (total >> 3)) v = ((((idx & ~63) | ((idx & 31) << 1) | ((idx >> 5) & 1))) << 3) | xcd;   // a bijection inside full groups of 64 slots
  }
  int item = xcd_major(v, total);
  SfTile t;
  if (ORDER == 3) {
    t.tx0 = (item % tiles_x) * SfCfg::TX;
    item /= tiles_x;
    t.ty0 = (item % tiles_y) * SfCfg::TY;
    item /= tiles_y;
    t.tz0 = (item % tiles_z) * SfCfg::TZ;
    t.b = item / tiles_z;
    return t;
  }
  if (ORDER == 1 || ORDER == 2) {
    t.tx0 = (item % tiles_x) * SfCfg::TX;
    item /= tiles_x;
    t.tz0 = (item % tiles_z) * SfCfg::TZ;
    item /= tiles_z;
  } else {
    t.tz0 = (item % tiles_z) * SfCfg::TZ;
    item /= tiles_z;
    t.tx0 = (item % tiles_x) * SfCfg::TX;
    item /= tiles_x;
  }
  t.ty0 = (item % tiles_y) * SfCfg::TY;
  t.b = item / tiles_y;
  return t;
}

// in (B, CIN, D, H, W) float32, W % 4 == 0, 16-byte aligned; wpk: [chunk][kz * 3 + ky][slice][lane] 16-byte lane images, then
// scale[8] (ABN scale x 2^-kw), shift[8] (float32); out (B, 8, D, H, W).  TERMS: 3 (default) or 4 (+ x_b w_b: A/B of the accuracy).
template <int CIN, int TERMS>
__global__ __launch_bounds__(SfCfg::THREADS, 2) void conv0_sf_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                                    float *__restrict__ out, int B, int D, int H, int W, int tiles_x,
                                                                    int tiles_y, int tiles_z, float slope) {
  using Cfg = SfCfg;
  constexpr int ORDER = CASMVS_SF_ORDER >= 0 ? CASMVS_SF_ORDER : (CIN == 16 ? 1 : 2);
  constexpr int NCH = CIN / 8, NT = Cfg::NT, NR = Cfg::NR, NWL = Cfg::NWL, IX = Cfg::IX, IY = Cfg::IY, NV = Cfg::NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                      // [2][NV]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                      // [9][2][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, u = lane >> 4;
  const int total = tiles_x * tiles_y * tiles_z * B;
  if ((int)blockIdx.x >= total) return;
  const int HW = H * W, cs = D * HW;
  const size_t in_ss = (size_t)CIN * cs, out_ss = (size_t)8 * cs;
  const float *tail = reinterpret_cast<const float *>(wpk + (size_t)NCH * Cfg::W_BYTES);
  float sc[2], sh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sc[h] = tail[2 * u + h];
    sh[h] = tail[8 + 2 * u + h];
  }
  const rsrc_t wsrc = make_rsrc(reinterpret_cast<const float *>(wpk), (size_t)NCH * Cfg::W_BYTES);
  const rsrc_t none = make_rsrc(in, 0);

  // lane's B voxel of staged row 0 of this wave's z plane (kz = 0): slot (wave * IY) * ROW + slot(2 j + u + 3)
  const int vbase = wave * IY * Cfg::ROW + Cfg::slot(2 * jcol + u + 3);

  // staging plan of the current prefetch target: item e = tid + 256 r -> (iz, iy, 4-x group)
  int voff[NR], vox[NR], vxor[NR];
  auto plan = [&](const SfTile &tc) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = tid + r * Cfg::THREADS;
      const int iz = e / (IY * (IX / 4)), rem = e - iz * (IY * (IX / 4));
      const int iy = rem / (IX / 4), g = rem - iy * (IX / 4);
      const int gz = tc.tz0 - 1 + iz, gy = tc.ty0 - 1 + iy, gx = tc.tx0 - 4 + 4 * g;
      const bool ok = e < Cfg::ITEMS && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 4 == 0
      voff[r] = ok ? (gz * HW + gy * W + gx) * 4 : kOOB;
      vox[r] = e < Cfg::ITEMS ? (iz * IY + iy) * Cfg::ROW + 4 * g : -1;   // + slot-swizzled j (bit 1 flips with bit 3 of x = bit 1 of g)
      vxor[r] = ((g >> 1) & 1) << 1;
    }
  };
  f32x4v R[NR][8];
  u32x4 WR[NWL];
  auto prefetch = [&](const SfTile &tc, int chunk, bool exists, bool weights) {   // every load of (tile, chunk); nothing here waits
    const rsrc_t src = exists ? make_rsrc(in + (size_t)tc.b * in_ss, in_ss * 4) : none;
    if (weights) {
#pragma unroll
      for (int i = 0; i < NWL; ++i) {
        const int unit = tid + i * Cfg::THREADS;
        WR[i] = __builtin_bit_cast(u32x4, buf_load4(exists ? wsrc : none, unit < Cfg::WUNITS ? unit * 16 : kOOB, chunk * (int)Cfg::W_BYTES));
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) R[r][c] = (CASMVS_SF_ABL & 4) ? f32x4v{1.f, (float)c, 2.f, 3.f} : buf_load4(src, voff[r], (chunk * 8 + c) * cs * 4);
  };

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  int item = blockIdx.x;
  SfTile cur = sf_decode<ORDER>(item, total, tiles_x, tiles_y, tiles_z);
  plan(cur);
  prefetch(cur, 0, true, true);
  bool first = true;
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    const SfTile nxt = have_next ? sf_decode<ORDER>(next_item, total, tiles_x, tiles_y, tiles_z) : cur;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- the staged tile's largest magnitude (this thread's loads -> wave -> workgroup) ----
      float m = 0.0f;
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(R[r][c][j]));
      const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
      if (lane == 0) wmax[wave] = wm;
      __syncthreads();   // every wave is done with the previous chunk's LDS; the four maxima are visible
      float mult, inv;   // max |x| 2^kx in [2^14, 2^15); 2^-kx
      casmvs::tile_scale(wmax, mult, inv);
      // ---- registers -> LDS: the two float16 slices of every staged voxel, the chunk's lane images ----
      if (NCH > 1 || first) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
          const int unit = tid + i * Cfg::THREADS;
          if (unit < Cfg::WUNITS) wl[unit] = WR[i];
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (vox[r] < 0 || ((CASMVS_SF_ABL & 2) && R[r][0][0] != 12345.f)) continue;   // (second round: 104 of the 256 threads)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] = R[r][c][j];
          u32x4 o[2];
          split_voxel_f16(x, mult, o);
#pragma unroll
          for (int s = 0; s < 2; ++s) act[s * NV + vox[r] + (j ^ vxor[r])] = o[s];
        }
      }
      __syncthreads();
      first = false;
      if (ch + 1 < NCH) {
        prefetch(cur, ch + 1, true, true);
      } else {
        plan(nxt);
        prefetch(nxt, 0, have_next, NCH > 1);
      }
      // ---- matrix phase: 3 kz x (6 staged rows read once) x 3 ky x 4 column tiles x TERMS partial products ----
      f32x4 part[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) part[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#if CASMVS_SF_PRIO
      __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        u32x4 row[NT + 2][2];
#pragma unroll
        for (int yr = 0; yr < NT + 2; ++yr)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            row[yr][s] = (CASMVS_SF_ABL & 8) ? u32x4{(unsigned)kz, 1u, 2u, (unsigned)yr} : act[s * NV + vbase + (kz * IY + yr) * Cfg::ROW];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          u32x4 a[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) a[s] = wl[((kz * 3 + ky) * 2 + s) * 64 + lane];
          // partial products by decreasing magnitude class; consecutive MFMAs use different accumulators
          constexpr int PA[4] = {0, 0, 1, 1}, PB[4] = {0, 1, 0, 1};
#pragma unroll
          for (int p = 0; p < TERMS; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              if (CASMVS_SF_ABL & 1) part[t][0] += __builtin_bit_cast(float, a[PA[p]][0] ^ row[t + ky][PB[p]][1]);   // keeps the operands live
              else part[t] = mfma_f16(a[PA[p]], row[t + ky][PB[p]], part[t]);
            }
        }
      }
#if CASMVS_SF_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = NCH > 1 ? fmaf(part[t][q], inv, acc[t][q]) : part[t][q] * inv;
    }
    // ---- epilogue: y = lrelu(acc * scale + shift); lane holds rows 4 u + r = (co = 2 u + (r >> 1), x phase r & 1), column j ----
    const rsrc_t dst = make_rsrc(out + (size_t)cur.b * out_ss, out_ss * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int oz = cur.tz0 + wave, oy = cur.ty0 + t, ox = cur.tx0 + 2 * jcol;
      const bool ok = oz < D && oy < H && ox < W;   // W even: the pixel pair is inside or outside
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v0 = fmaf(acc[t][2 * h], sc[h], sh[h]), v1 = fmaf(acc[t][2 * h + 1], sc[h], sh[h]);
        v0 = v0 > 0.0f ? v0 : v0 * slope;
        v1 = v1 > 0.0f ? v1 : v1 * slope;
        buf_store2(f32x2{v0, v1}, dst, ok ? ((2 * u + h) * cs + (oz * H + oy) * W + ox) * 4 : kOOB, 0);
      }
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!have_next) break;
    item = next_item;
    cur = nxt;
  }
}

// lane-semantics probe of v_mfma_f32_16x16x32_f16: D = A B for small integer matrices (exact in float16)
__global__ void mfma_f16_probe_kernel(float *out) {
  const int lane = threadIdx.x, i = lane & 15, kb = lane >> 4;
  union { u32x4 v; _Float16 h[8]; } a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * kb + e;
    a.h[e] = (_Float16)((k == i || k == i + 16) ? (float)(1 + i) : 0.0f);
    b.h[e] = (_Float16)(float)(1 + k + 3 * i);
  }
  const f32x4 d = mfma_f16(a.v, b.v, f32x4{0.f, 0.f, 0.f, 0.f});
  for (int r = 0; r < 4; ++r) out[r * 64 + lane] = d[r];
}

inline uint16_t f16_bits(float x) {   // round to nearest even (host)
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}
inline float f16_value(float x) { return (float)(_Float16)x; }

}  // namespace

extern "C" size_t casmvs_conv0_splitf16_packed_bytes(int cin) {
  if (cin != 8 && cin != 16 && cin != 32) return 0;
  return (size_t)(cin / 8) * SfCfg::W_BYTES + 16 * sizeof(float);
}

// HOST-side packing: weight (8, cin, 3, 3, 3) float32 -> w' = 2^kw w (max |w'| in [2^13, 2^14)); per chunk of 8 input channels,
// per (kz, ky), per slice (f16(w'), f16(w' - f16(w'))), per lane the 8 float16 values
// A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[co = i >> 1][chunk * 8 + e][kz][ky][kx = (lane >> 4) - (i & 1)]);
// then scale[8] * 2^-kw, shift[8].
extern "C" int casmvs_conv0_splitf16_pack(int cin, const float *weight, const float *scale, const float *shift, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "conv0_splitf16_pack: null pointer");
  CASMVS_REQUIRE(cin == 8 || cin == 16 || cin == 32, "conv0_splitf16_pack: cin=%d (8, 16 or 32)", cin);
  float wmax = 0.0f;
  for (size_t i = 0; i < (size_t)8 * cin * 27; ++i) {
    CASMVS_REQUIRE(std::isfinite(weight[i]), "conv0_splitf16_pack: weight %zu is not finite", i);
    wmax = std::fmax(wmax, std::fabs(weight[i]));
  }
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);   // wmax in [2^(ex-1), 2^ex)
  const int kw = 14 - ex;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int ch = 0; ch < cin / 8; ++ch)
    for (int r9 = 0; r9 < 9; ++r9) {
      uint16_t img[2][64][8];
      for (int l = 0; l < 64; ++l) {
        const int i = l & 15, co = i >> 1, s = i & 1, uu = l >> 4, kx = uu - s;
        for (int e = 0; e < 8; ++e) {
          const float w = (kx >= 0 && kx <= 2) ? std::ldexp(weight[(((size_t)co * cin + ch * 8 + e) * 9 + r9) * 3 + kx], kw) : 0.0f;
          const float a = f16_value(w);
          img[0][l][e] = f16_bits(w);
          img[1][l][e] = f16_bits(w - a);
        }
      }
      std::memcpy(p, img, sizeof(img));
      p += 2 * 64 * 8;
    }
  float *tail = reinterpret_cast<float *>(p);
  for (int c = 0; c < 8; ++c) tail[c] = std::ldexp(scale ? scale[c] : 1.0f, -kw);
  for (int c = 0; c < 8; ++c) tail[8 + c] = shift ? shift[c] : 0.0f;
  return CASMVS_OK;
}

extern "C" int casmvs_conv0_splitf16_supported(int cin, int W) { return (cin == 8 || cin == 16 || cin == 32) && W % 4 == 0 && W >= 4; }

namespace {
int conv0_sf_launch(const void *packed, const float *in, float *out, int B, int cin, int D, int H, int W, float slope, int terms, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "conv0_splitf16_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && casmvs_conv0_splitf16_supported(cin, W), "conv0_splitf16_forward: B=%d cin=%d D=%d H=%d W=%d", B, cin, D, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(packed)) & 15) == 0, "conv0_splitf16_forward: 16-byte aligned pointers");
  CASMVS_REQUIRE((size_t)cin * D * H * W < ((size_t)1 << 29), "conv0_splitf16_forward: one sample's input tensor must hold < 2^29 floats");
  CASMVS_REQUIRE(terms == 0 || terms == 3 || terms == 4, "conv0_splitf16_forward: terms=%d (0 = 3, 3 or 4)", terms);
  using Cfg = SfCfg;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY), tiles_z = casmvs::ceil_div(D, Cfg::TZ);
  const long total = (long)tiles_x * tiles_y * tiles_z * B;
  CASMVS_REQUIRE(total < (1L << 31), "conv0_splitf16_forward: too many tiles");
  const unsigned char *wp = reinterpret_cast<const unsigned char *>(packed);
  hipStream_t st = (hipStream_t)stream;
#define CASMVS_SF(CIN, T)                                                                                                       \
  {                                                                                                                             \
    auto kernel = conv0_sf_kernel<CIN, T>;                                                                                      \
    if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "conv0_sf_kernel")) return rc; \
    const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);         \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, in, wp, \
                       out, B, D, H, W, tiles_x, tiles_y, tiles_z, slope);                                                      \
  }
  const bool four = terms == 4;
  if (cin == 8) { if (four) CASMVS_SF(8, 4) else CASMVS_SF(8, 3) }
  else if (cin == 16) { if (four) CASMVS_SF(16, 4) else CASMVS_SF(16, 3) }
  else { if (four) CASMVS_SF(32, 4) else CASMVS_SF(32, 3) }
#undef CASMVS_SF
  return casmvs::check_launch("conv0_sf_kernel");
}
}  // namespace

extern "C" int casmvs_conv0_splitf16_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int D, int H, int W,
                                                 float slope, int terms, void *stream) {
  return conv0_sf_launch(packed, in, out, B, cin, D, H, W, slope, terms, stream);
}

// As casmvs_selftest_mfma_bf16, for v_mfma_f32_16x16x32_f16.
extern "C" int casmvs_selftest_mfma_f16(float *dump) {
  casmvs::clear_error();
  float *d = nullptr;
  if (hipMalloc(&d, 4 * 64 * sizeof(float)) != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "selftest_mfma_f16: hipMalloc failed");
  hipLaunchKernelGGL(mfma_f16_probe_kernel, dim3(1), dim3(64), 0, 0, d);
  float h[4 * 64];
  hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "selftest_mfma_f16: %s", hipGetErrorString(e));
  if (dump)
    for (int i = 0; i < 4 * 64; ++i) dump[i] = h[i];
  for (int r = 0; r < 4; ++r)
    for (int l = 0; l < 64; ++l) {
      const int i = 4 * (l >> 4) + r, j = l & 15;
      const float want = (float)(1 + i) * (float)((1 + i + 3 * j) + (17 + i + 3 * j));
      if (h[r * 64 + l] != want)
        return casmvs::fail(CASMVS_ERR_HIP, "selftest_mfma_f16: reg=%d lane=%d: got %g want %g", r, l, h[r * 64 + l], want);
    }
  return CASMVS_OK;
}
