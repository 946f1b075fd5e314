// FeatureNet.conv0 = ConvBnReLU(3, 8, 3) -> ConvBnReLU(8, 8, 3) at full resolution (models/mvsnet.py:14-16, modules.py:8-18) as ONE kernel with BOTH layers on
// the f16 matrix cores, in the float32-grade split arithmetic of conv0_splitf16.hip (every float32 operand = two float16 slices behind an exact power-of-two
// scaling, three partial products, float32 accumulation).
//
// Why (round 6).  The two layers run as separate float32-MFMA launches at 94 + 143 us for 24 images of 512 x 640 (profiles/r06_bench_full.json): the 3-channel
// image is 94 MB, the 8-channel maps 252 MB each - 345 + 503 MB of traffic.  Fused, the 8-channel intermediate lives in LDS only: 345 MB.  Round 4's fused
// attempt (first layer on the vector ALU: 216 multiply-adds per pixel) was 1.15x the pair and was deleted; here the first layer is a matrix problem too:
//   layer 1: rows = (8 output channels x 2 x-phases), K = 32 = (2 taps ky) x (4 input x) x (3 channels + 1 zero), two K steps for the three ky (the second
//            half of step 1 is zero weights): 27 of 64 K slots used, 6 matrix instructions per 32 pixels - a tenth of the vector form's issue slots;
//   layer 2: the PX form of fpn_fused_sf.hip with one chunk: rows = (8 output channels x 2 x-phases), K = 32 = 4 input x x 8 channels, 3 ky steps.
// Layer 1's results leave the accumulators as float16 slice PAIRS of two channels (one dword per pixel and lane): the transposition into layer 2's
// 8-channel operand units is the LDS write itself.
//
// Workgroup = 256 threads, output tile 20 x 30 pixels (layer 1 computes 24 x 32: rows y0 - 1 .. y0 + 22 of which 22 are used, x0 - 1 .. x0 + 30 = 16 pixel
// pairs exactly), input tile 26 x 34.  Wave w: layer-1 rows 6 w .. 6 w + 5, output rows 5 w .. 5 w + 4.  Four workgroup barriers per tile (the two staged
// units' maxima, the two LDS images).  LDS: image slices 15 KiB + layer-1 slices 28 KiB + lane images 10 KiB = 53 KiB, requested as 56 KiB: two workgroups per CU (three measured slower).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "buffer_ops.h"
#include "common.h"
#include "split_f16.h"

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct M0Cfg {
  static constexpr int THREADS = 256, NT1 = 6, NT2 = 5;
  static constexpr int TY = 4 * NT2, TX = 30;
  static constexpr int R1 = 4 * NT1;                 // layer-1 rows computed: 24 (y0 - 1 .. y0 + 22; rows 22, 23 are forced to zero and not stored)
  static constexpr int R1_USED = TY + 2;             // 22
  static constexpr int RI = R1 + 2;                  // staged image rows: 26 (y0 - 2 .. y0 + 23)
  static constexpr int IPX = 36, IRU = IPX / 2;      // staged image pixels per row (34 used: x0 - 2 .. x0 + 31) / 16-byte units per row (2 pixels x 4 channels x 2 B)
  static constexpr int IMG_UNITS = RI * IRU;         // 16-byte units per image slice: 468
  static constexpr int IGROUPS = 17;                 // pixel pairs loaded per row and channel
  static constexpr int ITEMS = RI * IGROUPS;         // (row, pixel pair) staging items: 442 = two rounds
  static constexpr int NR = (ITEMS + THREADS - 1) / THREADS;
  static constexpr int IX = 40, ROW = IX + 1;        // layer-1 slots per row as fpn_fused_sf.hip stages them (slot index = x - (x0 - 4)); odd row stride
  static constexpr int NV = R1_USED * ROW;           // 16-byte slots per layer-1 slice: 902 (the two spare rows are not stored)
  static __host__ __device__ constexpr int slot(int x) { return x ^ (((x >> 3) & 1) << 1); }   // as FsCfg::slot
  static constexpr int A1_UNITS = 2 * 2 * 64, A2_UNITS = 3 * 2 * 64;   // lane images [step][slice][lane] / [ky][slice][lane]
  static constexpr size_t IMG_BYTES = (size_t)2 * IMG_UNITS * 16, ACT_BYTES = (size_t)2 * NV * 16;
  static constexpr size_t W_BYTES = (size_t)(A1_UNITS + A2_UNITS) * 16;       // 10 240
  static constexpr size_t PACKED_BYTES = W_BYTES + 32 * sizeof(float);      // + scale0 | shift0 | scale1 | shift1
  static constexpr size_t LDS_USED = IMG_BYTES + ACT_BYTES + W_BYTES + 32;    // 54 112 (+ the two sets of four wave maxima)
  // requested: the 56 KiB floor of the split-f16 kernels = two workgroups per CU.  Measured (round 6, 24 images of 512 x 640, dirtied caches): 161 us at two
  // workgroups per CU, 190 us at three (138 registers and 53 KiB would allow them) - as for every f16 kernel of this library, two waves per SIMD is the optimum
  static constexpr size_t LDS_BYTES = LDS_USED < CASMVS_SF_LDS_FLOOR ? (size_t)CASMVS_SF_LDS_FLOOR : LDS_USED;
};

__device__ __forceinline__ f32x4 m0_mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// imgs (N, 3, H, W) float32, W % 2 == 0, 8-byte aligned; packed: casmvs_fnet_conv0_mm_pack; out (N, 8, H, W)
__global__ __launch_bounds__(M0Cfg::THREADS, 2) void fnet_conv0_mm_kernel(const float *__restrict__ imgs, const unsigned char *__restrict__ packed,
                                                                         float *__restrict__ out, int N, int H, int W, int tiles_x, int tiles_y, float slope) {
  using Cfg = M0Cfg;
  constexpr int NT1 = Cfg::NT1, NT2 = Cfg::NT2, NR = Cfg::NR, ROW = Cfg::ROW, NV = Cfg::NV, IRU = Cfg::IRU, IMG_UNITS = Cfg::IMG_UNITS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *img = reinterpret_cast<u32x4 *>(smem_raw);                                                  // [slice][RI][IRU]: (2 pixels x (3 channels, 0)) float16
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw + Cfg::IMG_BYTES);                                 // [slice][R1][ROW]: 8 channels float16 per pixel
  unsigned *actw = reinterpret_cast<unsigned *>(smem_raw + Cfg::IMG_BYTES);                          // the same as dwords (channel pairs)
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::IMG_BYTES + Cfg::ACT_BYTES);                 // A1 [step][slice][64], A2 [ky][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::IMG_BYTES + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [2 sets][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, u = lane >> 4;
  const int total = tiles_x * tiles_y * N;
  if ((int)blockIdx.x >= total) return;
  const int HW = H * W;
  for (int unit = tid; unit < Cfg::A1_UNITS + Cfg::A2_UNITS; unit += Cfg::THREADS) wl[unit] = reinterpret_cast<const u32x4 *>(packed)[unit];
  const float *tail = reinterpret_cast<const float *>(packed + Cfg::W_BYTES);
  float sc0[2], sh0[2], sc1[2], sh1[2];   // the lane's result rows 4 u + r = (channel 2 u + (r >> 1), x phase r & 1) in both layers
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sc0[h] = tail[2 * u + h];
    sh0[h] = tail[8 + 2 * u + h];
    sc1[h] = tail[16 + 2 * u + h];
    sh1[h] = tail[24 + 2 * u + h];
  }
  const rsrc_t none = make_rsrc(imgs, 0);

  // layer 1, lane (column j, K block u): step 0 reads image row r + (u >> 1), step 1 row r + 2 (K blocks 2, 3 of step 1 carry zero weights: any staged data),
  // pixel pair 2 j + 2 (u & 1): 16-byte unit (row) * IRU + j + (u & 1)
  const int b1s0 = (u >> 1) * IRU + jcol + (u & 1), b1s1 = 2 * IRU + jcol + (u & 1);
  // layer 2, lane (column j, input x offset u): slot(2 j + u + 3) of layer-1 row (NT2 wave + t + ky)
  const int vbase2 = NT2 * wave * ROW + Cfg::slot(2 * jcol + u + 3);
  // where this lane's layer-1 results go: layer-1 row NT1 wave + t, pixels 2 j, 2 j + 1, channel pair u -> dword u of the pixel's unit.  (These 24 dword
  // stores per lane and tile are 4-way bank conflicted - the pixels of lanes j, j + 4, j + 8, j + 12 are 128 bytes apart - tools/lds_bank_profile.py: 1.54x
  // the stores' floor, ~3 % of a tile's time; the unit layout is the one that keeps layer 2's 16-byte operand reads conflict-free.)
  const int wslot0 = (NT1 * wave * ROW + Cfg::slot(2 * jcol + 3)) * 4 + u, wslot1 = (NT1 * wave * ROW + Cfg::slot(2 * jcol + 4)) * 4 + u;

  auto decode = [&](int v, int &n, int &ty0, int &tx0) {
    int item = xcd_major(v, total);   // x fastest, then y, then image
    tx0 = (item % tiles_x) * Cfg::TX;
    item /= tiles_x;
    ty0 = (item % tiles_y) * Cfg::TY;
    n = item / tiles_y;
  };
  // staging item e = tid + 256 r -> (image row ri, pixel pair g): the three channels of pixels x0 - 2 + 2 g, + 1
  int voff[NR], iunit[NR];
  auto plan = [&](int ty0, int tx0) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int e = tid + r * Cfg::THREADS;
      const int ri = e / Cfg::IGROUPS, g = e - ri * Cfg::IGROUPS;
      const int gy = ty0 - 2 + ri, gx = tx0 - 2 + 2 * g;
      const bool ok = e < Cfg::ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W and x0 even: a pair is inside or outside
      voff[r] = ok ? (gy * W + gx) * 4 : kOOB;
      iunit[r] = e < Cfg::ITEMS ? ri * IRU + g : -1;
    }
  };
  f32x2 J[NR][3];
  auto prefetch = [&](int n, bool exists) {
    const rsrc_t src = exists ? make_rsrc(imgs + (size_t)n * 3 * HW, (size_t)3 * HW * 4) : none;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) J[r][c] = buf_load2(src, voff[r], c * HW * 4);
  };

  int item = blockIdx.x, n, ty0, tx0;
  decode(item, n, ty0, tx0);
  plan(ty0, tx0);
  prefetch(n, true);
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    int nn = n, nty0 = ty0, ntx0 = tx0;
    if (have_next) decode(next_item, nn, nty0, ntx0);
    // ---- the image tile's largest magnitude -> scale; the two float16 slices of every staged pixel pair ----
    float m = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) m = casmvs::absmax3(m, J[r][c][0], J[r][c][1]);
    const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
    if (lane == 0) wmax[wave] = wm;
    __syncthreads();   // A: every wave is done with the previous tile's LDS images; the four maxima are visible (and, the first time, the lane images)
    float mult, inv;
    casmvs::tile_scale(wmax, mult, inv);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (iunit[r] < 0) continue;
      u32x4 o[2];   // [slice]: (pixel 0: channels (0, 1), (2, zero); pixel 1: the same)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        unsigned a01, b01, a2z, b2z;
        casmvs::split_pair_f16(J[r][0][p], J[r][1][p], mult, a01, b01);
        casmvs::split_pair_f16(J[r][2][p], 0.0f, mult, a2z, b2z);
        o[0][2 * p] = a01; o[0][2 * p + 1] = a2z;
        o[1][2 * p] = b01; o[1][2 * p + 1] = b2z;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) img[s * IMG_UNITS + iunit[r]] = o[s];
    }
    __syncthreads();   // B: the image slices are visible
    plan(nty0, ntx0);
    prefetch(nn, have_next);
    // ---- layer 1: 6 rows x 2 K steps x 3 partial products ----
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc1[NT1];
    {
      u32x4 a[2][2];   // [step][slice]
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int s = 0; s < 2; ++s) a[st][s] = wl[(st * 2 + s) * 64 + lane];
      constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
      // three rows at a time: their accumulators are independent, so no matrix instruction waits for its predecessor's result
#pragma unroll
      for (int t0 = 0; t0 < NT1; t0 += 3) {
        u32x4 b[3][2][2];   // [row][step][slice]
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int r1 = NT1 * wave + t0 + t;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            b[t][0][s] = img[s * IMG_UNITS + r1 * IRU + b1s0];
            b[t][1][s] = img[s * IMG_UNITS + r1 * IRU + b1s1];
          }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) acc1[t0 + t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 3; ++t) acc1[t0 + t] = m0_mfma(a[st][PA[p]], b[t][st][PB[p]], acc1[t0 + t]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- layer 1's epilogue: ABN + leaky-relu, ZERO outside the image (layer 2's padding, not a convolution result) and in the two spare rows ----
    float v1[NT1][4];
    float m2 = 0.0f;
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      const int r1 = NT1 * wave + t;
      const int gy = ty0 - 1 + r1, gx = tx0 - 1 + 2 * jcol;
      const bool row_in = r1 < Cfg::R1_USED && gy >= 0 && gy < H;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = r >> 1, ph = r & 1;
        float v = fmaf(acc1[t][r] * inv, sc0[h], sh0[h]);
        v = v > 0.0f ? v : v * slope;
        v = (row_in && gx + ph >= 0 && gx + ph < W) ? v : 0.0f;
        v1[t][r] = v;
      }
      m2 = casmvs::absmax3(casmvs::absmax3(m2, v1[t][0], v1[t][1]), v1[t][2], v1[t][3]);
    }
    const unsigned wm2 = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m2));
    if (lane == 0) wmax[4 + wave] = wm2;
    __syncthreads();   // C: the maxima of layer 1's tile are visible (every wave has finished reading the image slices)
    float mult2, inv2;
    casmvs::tile_scale(wmax + 4, mult2, inv2);
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      if (NT1 * wave + t >= Cfg::R1_USED) continue;   // (wave-uniform: the last wave's two spare rows)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        unsigned ab, bb;   // channels (2 u, 2 u + 1) of pixel 2 j + ph
        casmvs::split_pair_f16(v1[t][ph], v1[t][2 + ph], mult2, ab, bb);
        const int w = (ph ? wslot1 : wslot0) + t * ROW * 4;
        actw[w] = ab;
        actw[NV * 4 + w] = bb;
      }
    }
    __syncthreads();   // D: layer 1's slices are visible
    // ---- layer 2: 3 ky x 5 rows x 3 partial products ----
    __builtin_amdgcn_sched_barrier(0);
    u32x4 row[NT2 + 2][2];
#pragma unroll
    for (int yr = 0; yr < NT2 + 2; ++yr)
#pragma unroll
      for (int s = 0; s < 2; ++s) row[yr][s] = act[s * NV + vbase2 + yr * ROW];
    f32x4 acc2[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      u32x4 a[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) a[s] = wl[Cfg::A1_UNITS + (ky * 2 + s) * 64 + lane];
      constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < NT2; ++t) acc2[t] = m0_mfma(a[PA[p]], row[t + ky][PB[p]], acc2[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: y = lrelu(acc 2^-k scale1 + shift1); lane holds rows 4 u + r = (channel 2 u + (r >> 1), x phase r & 1) of column j ----
    const rsrc_t dst = make_rsrc(out + (size_t)n * 8 * HW, (size_t)8 * HW * 4);
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
      const int oy = ty0 + NT2 * wave + t, ox = tx0 + 2 * jcol;
      const bool ok = jcol < Cfg::TX / 2 && oy < H && ox < W;   // W even: the pixel pair is inside or outside
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v0 = fmaf(acc2[t][2 * h] * inv2, sc1[h], sh1[h]), v1o = fmaf(acc2[t][2 * h + 1] * inv2, sc1[h], sh1[h]);
        v0 = v0 > 0.0f ? v0 : v0 * slope;
        v1o = v1o > 0.0f ? v1o : v1o * slope;
        buf_store2(f32x2{v0, v1o}, dst, ok ? ((2 * u + h) * HW + oy * W + ox) * 4 : kOOB, 0);
      }
    }
    if (!have_next) break;
    item = next_item;
    n = nn;
    ty0 = nty0;
    tx0 = ntx0;
  }
}

inline uint16_t f16_bits_m0(float x) {
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

inline int scale_exponent(const float *w, int n) {   // kw with max |2^kw w| in [2^13, 2^14)
  float wmax = 0.0f;
  for (int i = 0; i < n; ++i) wmax = std::fmax(wmax, std::fabs(w[i]));
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
  return 14 - ex;
}

}  // namespace

extern "C" size_t casmvs_fnet_conv0_mm_packed_bytes(void) { return M0Cfg::PACKED_BYTES; }

// HOST-side packing.  w0 (8, 3, 3, 3), w1 (8, 8, 3, 3): the torch weights of conv0.0 / conv0.1; scale / shift: their folded eval-mode ABN (nullptr = 1 / 0).
// Layer 1, per K step st and slice: A[i = lane & 15][k = 8 kb + e], kb = lane >> 4: row i = (co = i >> 1, x phase i & 1); K block kb of step 0 = (ky = kb >> 1,
// pixel pair q = kb & 1), of step 1 = (ky = 2, q = kb) for kb < 2 and zero for kb >= 2; element e = (pixel e >> 2 of the pair, input channel e & 3; channel 3
// is zero): input x offset xp = 2 q + (e >> 2), tap kx = xp - phase (zero outside 0 .. 2).  Layer 2: fpn_fused_sf.hip's image of one chunk:
// A[i][k = 8 (lane >> 4) + e] = slice(w1'[co = i >> 1][ci = e][ky][kx = (lane >> 4) - (i & 1)]).  Then scale0 2^-kw0 | shift0 | scale1 2^-kw1 | shift1.
extern "C" int casmvs_fnet_conv0_mm_pack(const float *w0, const float *scale0, const float *shift0, const float *w1, const float *scale1, const float *shift1,
                                         void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(w0 && w1 && packed, "fnet_conv0_mm_pack: null pointer");
  for (int i = 0; i < 8 * 3 * 9; ++i) CASMVS_REQUIRE(std::isfinite(w0[i]), "fnet_conv0_mm_pack: conv0.0 weight %d is not finite", i);
  for (int i = 0; i < 8 * 8 * 9; ++i) CASMVS_REQUIRE(std::isfinite(w1[i]), "fnet_conv0_mm_pack: conv0.1 weight %d is not finite", i);
  const int kw0 = scale_exponent(w0, 8 * 3 * 9), kw1 = scale_exponent(w1, 8 * 8 * 9);
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int st = 0; st < 2; ++st) {
    uint16_t img[2][64][8];
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, co = i >> 1, ph = i & 1, kb = l >> 4;
      const bool live = st == 0 || kb < 2;
      const int ky = st == 0 ? (kb >> 1) : 2, q = kb & 1;
      for (int e = 0; e < 8; ++e) {
        const int ci = e & 3, kx = 2 * q + (e >> 2) - ph;
        const float w = (live && ci < 3 && kx >= 0 && kx <= 2) ? std::ldexp(w0[((co * 3 + ci) * 3 + ky) * 3 + kx], kw0) : 0.0f;
        const float a = (float)(_Float16)w;
        img[0][l][e] = f16_bits_m0(w);
        img[1][l][e] = f16_bits_m0(w - a);
      }
    }
    std::memcpy(p, img, sizeof(img));
    p += 2 * 64 * 8;
  }
  for (int ky = 0; ky < 3; ++ky) {
    uint16_t img[2][64][8];
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, co = i >> 1, s = i & 1, uu = l >> 4, kx = uu - s;
      for (int e = 0; e < 8; ++e) {
        const float w = (kx >= 0 && kx <= 2) ? std::ldexp(w1[((co * 8 + e) * 3 + ky) * 3 + kx], kw1) : 0.0f;
        const float a = (float)(_Float16)w;
        img[0][l][e] = f16_bits_m0(w);
        img[1][l][e] = f16_bits_m0(w - a);
      }
    }
    std::memcpy(p, img, sizeof(img));
    p += 2 * 64 * 8;
  }
  float *tail = reinterpret_cast<float *>(p);
  for (int c = 0; c < 8; ++c) tail[c] = std::ldexp(scale0 ? scale0[c] : 1.0f, -kw0);
  for (int c = 0; c < 8; ++c) tail[8 + c] = shift0 ? shift0[c] : 0.0f;
  for (int c = 0; c < 8; ++c) tail[16 + c] = std::ldexp(scale1 ? scale1[c] : 1.0f, -kw1);
  for (int c = 0; c < 8; ++c) tail[24 + c] = shift1 ? shift1[c] : 0.0f;
  return CASMVS_OK;
}

extern "C" int casmvs_fnet_conv0_mm_supported(int W) { return W % 2 == 0 && W >= 2; }

extern "C" int casmvs_fnet_conv0_mm_f32(const void *packed, const float *imgs, float *out, int N, int H, int W, float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && imgs && out, "fnet_conv0_mm: null pointer");
  CASMVS_REQUIRE(N > 0 && H > 0 && casmvs_fnet_conv0_mm_supported(W), "fnet_conv0_mm: N=%d H=%d W=%d (W even)", N, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(imgs) | reinterpret_cast<size_t>(out)) & 7) == 0 && (reinterpret_cast<size_t>(packed) & 15) == 0,
                 "fnet_conv0_mm: 8-byte aligned tensors, 16-byte aligned image");
  CASMVS_REQUIRE((size_t)8 * H * W < ((size_t)1 << 29), "fnet_conv0_mm: one image's output must hold < 2^29 floats");
  using Cfg = M0Cfg;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY);
  const long total = (long)tiles_x * tiles_y * N;
  CASMVS_REQUIRE(total < (1L << 31), "fnet_conv0_mm: too many tiles");
  auto kernel = fnet_conv0_mm_kernel;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "fnet_conv0_mm_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, (hipStream_t)stream, imgs,
                     reinterpret_cast<const unsigned char *>(packed), out, N, H, W, tiles_x, tiles_y, slope);
  return casmvs::check_launch("fnet_conv0_mm_kernel");
}
