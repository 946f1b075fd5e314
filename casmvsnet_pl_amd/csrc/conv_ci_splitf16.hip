// CostRegNet's stride-1 U-Net layers with equal channel counts (conv2: 16 -> 16, conv4: 32 -> 32, conv6: 64 -> 64; Conv3d k3 s1 p1 +
// folded ABN + leaky-relu) on the f16 matrix cores with float32-grade arithmetic - the arithmetic of conv0_splitf16.hip
// (every float32 operand = two float16 slices behind exact power-of-two scalings, three partial products, float32
// accumulation) in the channel-inner ("CI") matrix form.
//
// Reference semantics: models/mvsnet.py:66,69,92-94 (`conv2`, `conv4`), models/modules.py:21-31 (ConvBnReLU3D).
//
// Why: these layers sit on the float32-input MFMA (which issues at the float32 VECTOR rate) at 0.5-0.7 of its peak; the f16
// matrix instruction does 8x the K in half the time, so the same product costs 3/16 of the matrix-pipe time and the layer
// becomes a question of staging bandwidth.
//
// Formulation: D[16 x 16] += A[16 x 32] B[32 x 16] with
//   rows    i = output channel 16 rb + i                 (rb < COUT / 16 row blocks share every B operand)
//   columns j = 16 consecutive output x of one (z, y) row
//   K       k = (h = k >> 4, ci = k & 15)                - two TAPS x 16 input channels: step m covers taps 2 m, 2 m + 1 of
//                                                          the 27 (tap t = (kz * 3 + ky) * 3 + kx; the 28th is zero weights)
// Lane l = (j = l & 15, kb = l >> 4) supplies the 8 channels 8 (kb & 1) .. + 8 of tap 2 m + (kb >> 1) at column j: ONE 16-byte
// LDS read from the staged tile, stored as planes [slice][channel half][z][y][x] of 16-byte units (8 float16 channels).  The two
// lane halves read at voxel offsets that differ by the step's tap distance - one of three values (next x, next row, next
// plane) - kept as three per-lane base sets, so a read is `base + immediate`.
//
// Workgroup = 256 threads (4 waves), output tile 4 x 4 x 16 voxels, wave w = z plane w, 4 (y) column tiles; per chunk of
// 16 input channels the halo tile 6 x 6 x 20 voxels (x0 - 2 .. x0 + 17: 8-byte aligned pairs) x 2 slices x 32 B = 45 KiB and the
// chunk's lane images 14 steps x 2 slices x COUT / 16 KiB.  conv2: 73 KiB, two workgroups per CU, weights loaded once.  Volumes of
// 2 planes (conv4 at cascade level 0) use a 2 x 8 x 16 tile (halo 4 x 10 x 20).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"
#include "split_f16.h"

#ifndef CASMVS_CI_LOAD_AUX
#define CASMVS_CI_LOAD_AUX 0    // cache-policy bits of the activation loads (debug builds: 17 = sc0 | sc1, system-coherent)
#endif
#ifndef CASMVS_CI_ORDER
#define CASMVS_CI_ORDER 0       // A/B builds: 1 = x-fastest tile order, 2 = x fastest + CU pairing (buffer_ops.h: cu_pair_remap)
#endif
#ifndef CASMVS_CI_STORE_AUX
#define CASMVS_CI_STORE_AUX 0
#endif

#ifndef CASMVS_CI_SWP
#define CASMVS_CI_SWP 1   // A/B builds: 0 = the two voxels of a staging item written in plain order (2-way conflicted 16-byte writes, no selects)
#endif

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <int CIN, int COUT, int TZ_ = 4>
struct CiCfg {
  static constexpr int THREADS = 256, WAVES = 4, NT = 4;
  static constexpr int TZ = TZ_, TY = 16 / TZ_, TX = 16;           // 16 (z, y) rows of 16 x: 4 x 4 (default) or 2 x 8 (volumes of 2 planes)
  static constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 4;       // x0 - 2 .. x0 + 17
  static constexpr int RS = IX;                                      // 16-byte units per staged row of one plane (no padding: see the write order)
  static constexpr int NVOX = IZ * IY * RS;                          // units per plane: 720 (x 16 B = 45 x 256 B: planes start on the same bank)
  static constexpr int RB = COUT / 16, NCH = CIN / 16, STEPS = 14;
  static constexpr int ITEMS = IZ * IY * (IX / 2);                   // (z, y, pair of x) staging items: 360
  static constexpr int NR = (ITEMS + THREADS - 1) / THREADS;         // 2
  static constexpr int WUNITS = STEPS * RB * 2 * 64;                 // 16-byte units of a chunk's lane images: [step][row block][slice][lane]
  static constexpr int NWL = (WUNITS + THREADS - 1) / THREADS;       // 7 RB
  static constexpr size_t ACT_BYTES = (size_t)4 * NVOX * 16, W_BYTES = (size_t)WUNITS * 16;
  static constexpr size_t LDS_BYTES = ACT_BYTES + W_BYTES + 16;      // (16, 16): 74 768
  static constexpr int WG_PER_CU = LDS_BYTES * 2 <= 160 * 1024 ? 2 : 1;
};

__device__ __forceinline__ f32x4 mfma_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ void split8_f16(const float (&x)[8], float mult, u32x4 (&o)[2]) { casmvs::split8_f16(x, mult, o); }   // split_f16.h

struct CiTile {
  int tx0, ty0, tz0, b;
};
template <typename Cfg>
__device__ __forceinline__ CiTile ci_decode(int v, int total, int tiles_x, int tiles_y, int tiles_z) {
#if CASMVS_CI_ORDER == 2
  if (Cfg::WG_PER_CU == 2) v = cu_pair_remap(v, total);
#endif
  int item = xcd_major(v, total);   // z fastest, then x, then y (the halos of neighbouring tiles share an XCD's L2)
  CiTile t;
#if CASMVS_CI_ORDER >= 1   // A/B builds: x fastest, then z, then y
  t.tx0 = (item % tiles_x) * Cfg::TX;
  item /= tiles_x;
  t.tz0 = (item % tiles_z) * Cfg::TZ;
  item /= tiles_z;
#else
  t.tz0 = (item % tiles_z) * Cfg::TZ;
  item /= tiles_z;
  t.tx0 = (item % tiles_x) * Cfg::TX;
  item /= tiles_x;
#endif
  t.ty0 = (item % tiles_y) * Cfg::TY;
  t.b = item / tiles_y;
  return t;
}

// unit offset of tap t inside a plane, relative to the lane's voxel of tap 0
template <typename Cfg>
__host__ __device__ constexpr int ci_tap_off(int t) { return ((t / 9) * Cfg::IY + (t / 3) % 3) * Cfg::RS + t % 3; }

// in (B, CIN, D, H, W) float32, W % 2 == 0, 8-byte aligned; wpk: [chunk][step][row block][slice][lane] 16-byte lane images, then
// scale[COUT] (ABN scale x 2^-kw), shift[COUT]; out (B, COUT, D, H, W).
template <int CIN, int COUT, int TZ>
__global__ __launch_bounds__(256, (CiCfg<CIN, COUT, TZ>::WG_PER_CU)) void conv_ci_sf_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                                                       float *__restrict__ out, int B, int D, int H, int W, int tiles_x,
                                                                                       int tiles_y, int tiles_z, float slope) {
  using Cfg = CiCfg<CIN, COUT, TZ>;
  constexpr int NCH = Cfg::NCH, RB = Cfg::RB, NT = Cfg::NT, NR = Cfg::NR, NWL = Cfg::NWL, IX = Cfg::IX, IY = Cfg::IY, NVOX = Cfg::NVOX, RS = Cfg::RS;
  static_assert(TZ == 4 || TZ == 2, "tile 4 x 4 x 16 or 2 x 8 x 16");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                          // [slice][half][NVOX]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                          // [step][rb][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, kb = lane >> 4, half = kb & 1, hi_tap = kb >> 1;
  const int total = tiles_x * tiles_y * tiles_z * B;
  if ((int)blockIdx.x >= total) return;
  const int HW = H * W, cs = D * HW;
  const size_t in_ss = (size_t)CIN * cs, out_ss = (size_t)COUT * cs;
  const float *tail = reinterpret_cast<const float *>(wpk + (size_t)NCH * Cfg::W_BYTES);
  float sc[RB][4], sh[RB][4];   // lane holds rows 4 kb + r of every row block
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[rb][r] = tail[rb * 16 + 4 * kb + r];
      sh[rb][r] = tail[COUT + rb * 16 + 4 * kb + r];
    }
  const rsrc_t wsrc = make_rsrc(reinterpret_cast<const float *>(wpk), (size_t)NCH * Cfg::W_BYTES);
  const rsrc_t none = make_rsrc(in, 0);

  // lane's B unit (slice 0) of column tile t at tap 0: plane `half`, voxel (wave, t, j + 1) [staged x index of x0 + j - 1];
  // the upper lane half (taps 2 m + 1) adds the step's tap distance: next x / next row / next plane
  int vbx[NT], vby[NT], vbz[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r16 = wave * NT + t;   // this wave's t-th (z, y) row of the tile
    const int v = half * NVOX + ((r16 / Cfg::TY) * IY + r16 % Cfg::TY) * RS + jcol + 1;
    vbx[t] = v + hi_tap * 1;
    vby[t] = v + hi_tap * (RS - 2);
    vbz[t] = v + hi_tap * ((IY - 2) * RS - 2);
  }

  // staging plan of the current prefetch target: item e = tid + 256 r -> (iz, iy, pair of x)
  int voff[NR], vox[NR];
  auto plan = [&](const CiTile &tc) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = tid + r * Cfg::THREADS;
      const int iz = e / (IY * (IX / 2)), rem = e - iz * (IY * (IX / 2));
      const int iy = rem / (IX / 2), g = rem - iy * (IX / 2);
      const int gz = tc.tz0 - 1 + iz, gy = tc.ty0 - 1 + iy, gx = tc.tx0 - 2 + 2 * g;
      const bool ok = e < Cfg::ITEMS && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 2 == 0
      voff[r] = ok ? (gz * HW + gy * W + gx) * 4 : kOOB;
      vox[r] = e < Cfg::ITEMS ? (iz * IY + iy) * RS + 2 * g : -1;
    }
  };
  f32x2 R[NR][16];
  u32x4 WR[NWL];
  auto prefetch = [&](const CiTile &tc, int chunk, bool exists, bool weights) {   // every load of (tile, chunk); nothing here waits
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
      for (int c = 0; c < 16; ++c)
        R[r][c] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(src, voff[r], (chunk * 16 + c) * cs * 4, CASMVS_CI_LOAD_AUX));
  };

  f32x4 acc[NT][RB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the two voxels of an item go out in the order that keeps 8 consecutive lanes on 8 different 16-byte bank groups:
  // lanes 0-3 of every 8 write their even voxel first, lanes 4-7 their odd one
  const int swp = CASMVS_CI_SWP ? (lane >> 2) & 1 : 0;

  int item = blockIdx.x;
  CiTile cur = ci_decode<Cfg>(item, total, tiles_x, tiles_y, tiles_z);
  plan(cur);
  prefetch(cur, 0, true, true);
  bool first = true;
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    const CiTile nxt = have_next ? ci_decode<Cfg>(next_item, total, tiles_x, tiles_y, tiles_z) : cur;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- the staged tile's largest magnitude (this thread's loads -> wave -> workgroup) ----
      float m = 0.0f;
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int c = 0; c < 16; ++c) m = casmvs::absmax3(m, R[r][c][0], R[r][c][1]);
      const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
      if (lane == 0) wmax[wave] = wm;
      __syncthreads();   // every wave is done with the previous chunk's LDS; the four maxima are visible
      float mult, inv;   // max |x| 2^kx in [2^14, 2^15); 2^-kx
      casmvs::tile_scale(wmax, mult, inv);
      // ---- registers -> LDS ----
      if (NCH > 1 || first) {
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
          const int unit = tid + i * Cfg::THREADS;
          if (unit < Cfg::WUNITS) wl[unit] = WR[i];
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (vox[r] < 0) continue;   // (second round: 104 of the 256 threads)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          u32x4 o[2][2];   // [voxel of the pair][slice]
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            float x[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) x[c] = R[r][hf * 8 + c][p];
            split8_f16(x, mult, o[p]);
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            u32x4 first_v, second_v;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              first_v[q] = swp ? o[1][s][q] : o[0][s][q];
              second_v[q] = swp ? o[0][s][q] : o[1][s][q];
            }
            u32x4 *pl = act + (s * 2 + hf) * NVOX + vox[r];
            pl[swp] = first_v;
            pl[1 - swp] = second_v;
          }
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
      // ---- matrix phase: 14 steps (tap pairs) x 4 column tiles x RB row blocks x 3 partial products ----
      f32x4 part[NT][RB];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) part[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < Cfg::STEPS; ++st) {
        const int t0 = 2 * st;                                             // even tap of the step
        const int off = ci_tap_off<Cfg>(t0);
        const int kx0 = t0 % 3, ky0 = (t0 / 3) % 3;
        // distance to tap t0 + 1: next x, next row (kx0 == 2), next plane (kx0 == 2 && ky0 == 2); the 28th tap (zero weights) reads the
        // next x: staged, finite data
        u32x4 bv[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int base = (kx0 != 2 || t0 == 26) ? vbx[t] : (ky0 != 2 ? vby[t] : vbz[t]);
#pragma unroll
          for (int s = 0; s < 2; ++s) bv[t][s] = act[s * 2 * NVOX + base + off];
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          u32x4 a[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) a[s] = wl[((st * RB + rb) * 2 + s) * 64 + lane];
          constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t) part[t][rb] = mfma_f16(a[PA[p]], bv[t][PB[p]], part[t][rb]);
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
    const rsrc_t dst = make_rsrc(out + (size_t)cur.b * out_ss, out_ss * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int r16 = wave * NT + t;
      const int oz = cur.tz0 + r16 / Cfg::TY, oy = cur.ty0 + r16 % Cfg::TY, ox = cur.tx0 + jcol;
      const bool ok = oz < D && oy < H && ox < W;
      const int o0 = ok ? (4 * kb * cs + (oz * H + oy) * W + ox) * 4 : kOOB;   // the lane's first channel row (4 kb) is part of the lane offset
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaf(acc[t][rb][r], sc[rb][r], sh[rb][r]);
          v = v > 0.0f ? v : v * slope;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), dst, o0, (rb * 16 + r) * cs * 4, CASMVS_CI_STORE_AUX);
        }
        acc[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if (!have_next) break;
    item = next_item;
    cur = nxt;
  }
}

inline uint16_t f16_bits_ci(float x) {   // round to nearest even (host)
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

template <int CIN, int COUT, int TZ>
int launch_ci(const void *packed, const float *in, float *out, int B, int D, int H, int W, float slope, hipStream_t st) {
  using Cfg = CiCfg<CIN, COUT, TZ>;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY), tiles_z = casmvs::ceil_div(D, Cfg::TZ);
  const long total = (long)tiles_x * tiles_y * tiles_z * B;
  CASMVS_REQUIRE(total < (1L << 31), "conv_ci_splitf16_forward: too many tiles");
  auto kernel = conv_ci_sf_kernel<CIN, COUT, TZ>;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "conv_ci_sf_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, in,
                     reinterpret_cast<const unsigned char *>(packed), out, B, D, H, W, tiles_x, tiles_y, tiles_z, slope);
  return casmvs::check_launch("conv_ci_sf_kernel");
}

inline bool ci_shape_ok(int cin, int cout) { return (cin == 16 && cout == 16) || (cin == 32 && cout == 32) || (cin == 64 && cout == 64); }

}  // namespace

extern "C" size_t casmvs_conv_ci_splitf16_packed_bytes(int cin, int cout) {
  if (!ci_shape_ok(cin, cout)) return 0;
  return (size_t)(cin / 16) * 14 * (cout / 16) * 2 * 64 * 16 + (size_t)2 * cout * sizeof(float);
}

// HOST-side packing: weight (cout, cin, 3, 3, 3) float32 -> w' = 2^kw w (max |w'| in [2^13, 2^14)); per chunk of 16 input channels,
// per step m (taps 2 m, 2 m + 1), per row block, per slice (f16(w'), f16(w' - f16(w'))), per lane the 8 float16 values
// A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[co = 16 rb + i][ci = 16 chunk + 8 ((lane >> 4) & 1) + e][tap 2 m + (lane >> 5)]), zero for
// tap 27; then scale[cout] * 2^-kw, shift[cout].
extern "C" int casmvs_conv_ci_splitf16_pack(int cin, int cout, const float *weight, const float *scale, const float *shift, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "conv_ci_splitf16_pack: null pointer");
  CASMVS_REQUIRE(ci_shape_ok(cin, cout), "conv_ci_splitf16_pack: cin=%d cout=%d (16 -> 16, 32 -> 32 or 64 -> 64)", cin, cout);
  float wmax = 0.0f;
  for (size_t i = 0; i < (size_t)cout * cin * 27; ++i) {
    CASMVS_REQUIRE(std::isfinite(weight[i]), "conv_ci_splitf16_pack: weight %zu is not finite", i);
    wmax = std::fmax(wmax, std::fabs(weight[i]));
  }
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);   // wmax in [2^(ex-1), 2^ex)
  const int kw = 14 - ex;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int ch = 0; ch < cin / 16; ++ch)
    for (int st = 0; st < 14; ++st)
      for (int rb = 0; rb < cout / 16; ++rb) {
        uint16_t img[2][64][8];
        for (int l = 0; l < 64; ++l) {
          const int i = l & 15, kb = l >> 4, tap = 2 * st + (kb >> 1), co = 16 * rb + i;
          for (int e = 0; e < 8; ++e) {
            const int ci = 16 * ch + 8 * (kb & 1) + e;
            const float w = tap < 27 ? std::ldexp(weight[((size_t)co * cin + ci) * 27 + tap], kw) : 0.0f;
            const float a = (float)(_Float16)w;
            img[0][l][e] = f16_bits_ci(w);
            img[1][l][e] = f16_bits_ci(w - a);
          }
        }
        std::memcpy(p, img, sizeof(img));
        p += 2 * 64 * 8;
      }
  float *tail = reinterpret_cast<float *>(p);
  for (int c = 0; c < cout; ++c) tail[c] = std::ldexp(scale ? scale[c] : 1.0f, -kw);
  for (int c = 0; c < cout; ++c) tail[cout + c] = shift ? shift[c] : 0.0f;
  return CASMVS_OK;
}

extern "C" int casmvs_conv_ci_splitf16_supported(int cin, int cout, int W) { return ci_shape_ok(cin, cout) && W % 2 == 0 && W >= 2; }

extern "C" int casmvs_conv_ci_splitf16_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int cout, int D, int H, int W,
                                                   float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "conv_ci_splitf16_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && casmvs_conv_ci_splitf16_supported(cin, cout, W),
                 "conv_ci_splitf16_forward: B=%d cin=%d cout=%d D=%d H=%d W=%d", B, cin, cout, D, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out)) & 7) == 0 && (reinterpret_cast<size_t>(packed) & 15) == 0,
                 "conv_ci_splitf16_forward: 8-byte aligned tensors, 16-byte aligned image");
  CASMVS_REQUIRE((size_t)cin * D * H * W < ((size_t)1 << 29) && (size_t)cout * D * H * W < ((size_t)1 << 29),
                 "conv_ci_splitf16_forward: one sample's tensors must hold < 2^29 floats");
  hipStream_t st = (hipStream_t)stream;
  // volumes of at most 2 planes (conv4 at cascade level 0: D / 4 = 2): the 2 x 8 x 16 tile - a 4-deep tile would be half padding
  if (cin == 16) return D <= 2 ? launch_ci<16, 16, 2>(packed, in, out, B, D, H, W, slope, st) : launch_ci<16, 16, 4>(packed, in, out, B, D, H, W, slope, st);
  if (cin == 32) return D <= 2 ? launch_ci<32, 32, 2>(packed, in, out, B, D, H, W, slope, st) : launch_ci<32, 32, 4>(packed, in, out, B, D, H, W, slope, st);
  // 64 -> 64: a chunk's lane images are 112 KiB - with the 45 KiB tile that is the whole LDS of a CU (one workgroup; the 2 x 8 x 16 tile
  // does not fit: volumes of <= 2 planes stay on the float32 kernel, casmvs_conv_ci_splitf16_supported_volume)
  CASMVS_REQUIRE(D >= 3, "conv_ci_splitf16_forward: 64 -> 64 needs D >= 3 (D=%d)", D);
  return launch_ci<64, 64, 4>(packed, in, out, B, D, H, W, slope, st);
}
