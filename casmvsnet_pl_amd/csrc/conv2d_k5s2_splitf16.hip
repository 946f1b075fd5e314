// FeatureNet's two stride-2 layers conv1.0 (8 -> 16) and conv2.0 (16 -> 32): Conv2d k5 s2 p2 + folded ABN + leaky-relu (models/mvsnet.py:18,23,
// models/modules.py:8-18) on the f16 matrix cores in the float32-grade split arithmetic of conv0_splitf16.hip (two float16 slices per operand behind exact
// power-of-two scalings, three partial products, float32 accumulation).
//
// Why: on the float32-input MFMA these two launches keep the matrix pipe busy 54 % of their 2 x 149 us (SQ_VALU_MFMA_BUSY_CYCLES, profiles/r04_sq_counters_step.md)
// for 0.38 GB / 0.19 GB of traffic; the f16 instruction does the same products in 3/16 of the pipe time.
//
// Formulation: D[16 x 16] += A[16 x 32] B[32 x 16]: rows = 16 output channels, columns = 16 consecutive output x of one output row, K = 4 TAP slots x 8 input
// channels.  Output (oy, ox) reads inputs (2 oy - 2 + ky, 2 ox - 2 + kx): the staged patch keeps even and odd x in separate parts of a row (as conv_s2_splitf16.hip),
// so that the 16 lanes of a tap slot read 16 consecutive units.  The 25 taps go into 8 steps of 4 slots so that the two slots a 16-byte LDS read serves
// together (kb 0 / 1 and kb 2 / 3) always sit a multiple of 16 units apart: a slot pair = the same kx at two ky (the row stride is 80 units), the five ky = 4
// taps pair with a zero-weight slot that re-reads their own unit.  Workgroup = 256 threads, output tile 8 x 32; a unit = (tile, chunk of 8 input channels):
// 19 x 72 staged pixels (342 (row, quad) items: two rounds), its own scale; the next unit's loads are issued before the matrix phase.
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

// slot (step s, kb) -> tap: pair p = 2 s + (kb >> 1), member m = kb & 1.  p < 10: kx = p >> 1, ky = 2 (p & 1) + m.  10 <= p < 15: kx = p - 10, ky = 4 for
// m = 0, no tap (zero weights, the partner's unit) for m = 1.  p = 15: no tap (pair 14's unit).
struct K5Tap {
  int ky, kx;
  bool live;
};
__host__ __device__ constexpr K5Tap k5_tap(int s, int kb) {
  const int p = 2 * s + (kb >> 1), m = kb & 1;
  if (p < 10) return {2 * (p & 1) + m, p >> 1, true};
  if (p < 15) return {4, p - 10, m == 0};
  return {4, 4, false};
}

template <int CIN, int COUT>
struct K5Cfg {
  static constexpr int THREADS = 256, WAVES = 4, NT = 4, STEPS = 8;
  static constexpr int TY = 8, TX = 32;                              // output tile; wave w: rows 2 w, 2 w + 1 x two column tiles
  static constexpr int IY = 2 * TY + 3, IQ = 18;                     // staged rows 2 oy0 - 2 .. 2 oy0 + 16; quads of x from 2 ox0 - 4 (72 floats)
  static constexpr int ODD = 40, RS = 80;                            // a row: even x (staged index 2 i) at unit i, odd x at unit ODD + i, i < 36; RS = 0 (mod 16)
  static constexpr int NVOX = IY * RS;                               // units per slice: 1 520
  static constexpr int ITEMS = IY * IQ, NR = (ITEMS + THREADS - 1) / THREADS;   // 342 (row, quad) items: 2 rounds
  static constexpr int NCH = CIN / 8, RB = COUT / 16;
  static constexpr int WUNITS = NCH * STEPS * RB * 2 * 64;           // lane images [chunk][step][row block][slice][lane]: 16 KiB / 64 KiB
  static constexpr int NWL = WUNITS / THREADS;
  static constexpr size_t ACT_BYTES = (size_t)2 * NVOX * 16, W_BYTES = (size_t)WUNITS * 16;   // 48 640 + 16 384 / 65 536
  static constexpr size_t LDS_BYTES = ACT_BYTES + W_BYTES + 16;      // 65 040 (two workgroups per CU) / 114 192 (one)
  static constexpr int WG_PER_CU = LDS_BYTES * 2 <= (size_t)160 * 1024 ? 2 : 1;
  static_assert(CIN % 8 == 0 && COUT % 16 == 0 && WUNITS % THREADS == 0 && RS % 16 == 0 && ODD >= 36 && ODD + 36 <= RS, "shape");
};

__device__ __forceinline__ f32x4 k5_mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// in (N, CIN, H, W) float32, H, W even, W % 4 == 0, 16-byte aligned; wpk: [chunk][step][row block][slice][lane] 16-byte lane images, then scale[COUT]
// (ABN scale x 2^-kw), shift[COUT]; out (N, COUT, H / 2, W / 2).
template <int CIN, int COUT>
__global__ __launch_bounds__(256, (K5Cfg<CIN, COUT>::WG_PER_CU)) void conv2d_k5s2_sf_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                                                       float *__restrict__ out, int N, int H, int W, int tiles_x,
                                                                                       int tiles_y, float slope) {
  using Cfg = K5Cfg<CIN, COUT>;
  constexpr int NCH = Cfg::NCH, RB = Cfg::RB, NT = Cfg::NT, NR = Cfg::NR, NWL = Cfg::NWL, NVOX = Cfg::NVOX, RS = Cfg::RS, ODD = Cfg::ODD, IQ = Cfg::IQ, STEPS = Cfg::STEPS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                          // [slice][row][even x | odd x]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                          // [chunk][step][rb][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, kb = lane >> 4;
  const int total = tiles_x * tiles_y * N;
  if ((int)blockIdx.x >= total) return;
  const int Ho = H / 2, Wo = W / 2, hw = H * W, ohw = Ho * Wo;
  const size_t in_ss = (size_t)CIN * hw, out_ss = (size_t)COUT * ohw;
  const float *tail = reinterpret_cast<const float *>(wpk + Cfg::W_BYTES);
  float sc[RB][4], sh[RB][4];   // lane holds rows 4 kb + r of every row block
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[rb][r] = tail[rb * 16 + 4 * kb + r];
      sh[rb][r] = tail[COUT + rb * 16 + 4 * kb + r];
    }
  {
    const rsrc_t wsrc = make_rsrc(reinterpret_cast<const float *>(wpk), Cfg::W_BYTES);
    for (int i0 = 0; i0 < NWL; i0 += 8) {   // eight 16-byte loads in flight per round trip
      u32x4 WR[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) WR[i] = __builtin_bit_cast(u32x4, buf_load4(wsrc, i0 + i < NWL ? (tid + (i0 + i) * Cfg::THREADS) * 16 : kOOB, 0));
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i0 + i < NWL) wl[tid + (i0 + i) * Cfg::THREADS] = WR[i];
    }
  }
  const rsrc_t none = make_rsrc(in, 0);

  // lane's unit offset (slice 0) of step s relative to (tile row 0, column tile 0, j = 0): tap (ky, kx) -> staged row ky, staged x index 2 J + 2 + kx:
  // even kx: unit J + 1 + kx / 2, odd kx: unit ODD + J + 1 + (kx - 1) / 2
  int toff[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    int o = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      constexpr int dummy = 0;
      (void)dummy;
      const K5Tap t = k5_tap(s, k), u = t.live ? t : k5_tap(s, k & 2);   // a slot without a tap reads its partner's unit (pair 15: pair 14's)
      const K5Tap v = u.live ? u : k5_tap(STEPS - 1, 0);
      const int off = v.ky * RS + ((v.kx & 1) ? ODD : 0) + 1 + (v.kx >> 1);
      o = kb == k ? off : o;
    }
    toff[s] = o + jcol;
  }
  int vrow[NT];   // unit offset of this wave's column tile t: output row 2 wave + (t >> 1) -> staged row 2 (2 wave + (t >> 1)), columns 16 (t & 1) ..
#pragma unroll
  for (int t = 0; t < NT; ++t) vrow[t] = 2 * (2 * wave + (t >> 1)) * RS + 16 * (t & 1);

  // staging items of this thread: e = tid + 256 r -> (staged row, quad of x)
  int sunit[NR];
  bool staged[NR];
  int srow[NR], sq[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int e = tid + r * Cfg::THREADS;
    staged[r] = e < Cfg::ITEMS;
    srow[r] = e / IQ;
    sq[r] = e - srow[r] * IQ;
    sunit[r] = srow[r] * RS + 2 * sq[r];   // voxel v of the quad -> parity v & 1, index 2 q + (v >> 1)
  }
  struct Cursor {
    int item, n, oy0, ox0, ch;
    bool valid;
  };
  auto decode = [&](Cursor &c) {
    int item = xcd_major(c.item, total);
    c.ox0 = (item % tiles_x) * Cfg::TX;
    item /= tiles_x;
    c.oy0 = (item % tiles_y) * Cfg::TY;
    c.n = item / tiles_y;
  };
  auto advance = [&](const Cursor &c) {
    Cursor n = c;
    if (!c.valid) return n;
    n.ch = c.ch + 1;
    if (n.ch == NCH) {
      n.ch = 0;
      n.item = c.item + gridDim.x;
      n.valid = n.item < total;
      if (n.valid) decode(n);
    }
    return n;
  };
  f32x4 R[NR][8];
  auto load = [&](const Cursor &c) {
    const rsrc_t src = c.valid ? make_rsrc(in + (size_t)c.n * in_ss, in_ss * 4) : none;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int gy = 2 * c.oy0 - 2 + srow[r], gx = 2 * c.ox0 - 4 + 4 * sq[r];
      const bool ok = staged[r] && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 4 == 0: whole quads
      const int voff = ok ? (gy * W + gx) * 4 : kOOB;
#pragma unroll
      for (int k = 0; k < 8; ++k) R[r][k] = __builtin_bit_cast(f32x4, buf_load4(src, voff, (c.ch * 8 + k) * hw * 4));
    }
  };
  f32x4 acc[NT][RB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};

  Cursor cur;
  cur.item = blockIdx.x;
  cur.ch = 0;
  cur.valid = true;
  decode(cur);
  load(cur);
  for (;;) {
    const Cursor nxt = advance(cur);
    // ---- the staged unit's largest magnitude ----
    float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        m0 = casmvs::absmax3(casmvs::absmax3(m0, R[r][k][0], R[r][k][1]), R[r][k][2], R[r][k][3]);
        m1 = casmvs::absmax3(casmvs::absmax3(m1, R[r][k + 1][0], R[r][k + 1][1]), R[r][k + 1][2], R[r][k + 1][3]);
      }
    const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, fmaxf(m0, m1)));
    if (lane == 0) wmax[wave] = wm;
    __syncthreads();   // every wave is done with the previous unit's LDS; the four maxima are visible
    float mult, inv;
    casmvs::tile_scale(wmax, mult, inv);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (!staged[r]) continue;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = R[r][k][v];
        casmvs::split_u32x4 o[2];
        casmvs::split8_f16(x, mult, o);
        u32x4 *pl = act + sunit[r] + ((v & 1) ? ODD : 0) + (v >> 1);
        pl[0] = o[0];
        pl[NVOX] = o[1];
      }
    }
    __syncthreads();
    load(nxt);   // in flight behind the matrix phase
    // ---- matrix phase: 8 steps x column tiles x row blocks x 3 partial products ----
    const u32x4 *wch = wl + cur.ch * (STEPS * RB * 2 * 64) + lane;
    constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
    f32x4 part[NT][RB];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) part[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      u32x4 bv[NT][2];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) bv[t][sl] = act[sl * NVOX + vrow[t] + toff[s]];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        u32x4 a[2];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) a[sl] = wch[((s * RB + rb) * 2 + sl) * 64];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int t = 0; t < NT; ++t) part[t][rb] = k5_mfma(a[PA[q]], bv[t][PB[q]], part[t][rb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // the folds stay behind the last matrix instruction
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][rb][r] = NCH > 1 ? fmaf(part[t][rb][r], inv, acc[t][rb][r]) : part[t][rb][r] * inv;
    if (cur.ch == NCH - 1) {
      // ---- epilogue: y = lrelu(acc * scale + shift); lane holds channels 16 rb + 4 kb + r, column j ----
      const rsrc_t dst = make_rsrc(out + (size_t)cur.n * out_ss, out_ss * 4);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int oy = cur.oy0 + 2 * wave + (t >> 1), ox = cur.ox0 + 16 * (t & 1) + jcol;
        const bool ok = oy < Ho && ox < Wo;
        const int o0 = ok ? (4 * kb * ohw + oy * Wo + ox) * 4 : kOOB;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = fmaf(acc[t][rb][r], sc[rb][r], sh[rb][r]);
            v = v > 0.0f ? v : v * slope;
            buf_store(v, dst, o0, (rb * 16 + r) * ohw * 4);
          }
          acc[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if (!nxt.valid) break;
    cur = nxt;
  }
}

inline uint16_t f16_bits_k5(float x) {   // round to nearest even (host)
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

template <int CIN, int COUT>
int launch_k5(const void *packed, const float *in, float *out, int N, int H, int W, float slope, hipStream_t st) {
  using Cfg = K5Cfg<CIN, COUT>;
  const int tiles_x = casmvs::ceil_div(W / 2, Cfg::TX), tiles_y = casmvs::ceil_div(H / 2, Cfg::TY);
  const long total = (long)tiles_x * tiles_y * N;
  CASMVS_REQUIRE(total < (1L << 31), "conv2d_k5s2_splitf16_forward: too many tiles");
  auto kernel = conv2d_k5s2_sf_kernel<CIN, COUT>;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "conv2d_k5s2_sf_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, in,
                     reinterpret_cast<const unsigned char *>(packed), out, N, H, W, tiles_x, tiles_y, slope);
  return casmvs::check_launch("conv2d_k5s2_sf_kernel");
}

}  // namespace

extern "C" size_t casmvs_conv2d_k5s2_splitf16_packed_bytes(int cin, int cout) {
  if (cin == 8 && cout == 16) return K5Cfg<8, 16>::W_BYTES + 2 * 16 * sizeof(float);
  if (cin == 16 && cout == 32) return K5Cfg<16, 32>::W_BYTES + 2 * 32 * sizeof(float);
  return 0;
}

// HOST-side packing: weight (cout, cin, 5, 5) float32 -> w' = 2^kw w (max |w'| in [2^13, 2^14)); per chunk of 8 input channels, step, block of 16 output
// channels, slice (f16(w'), f16(w' - f16(w'))), per lane the 8 float16 values
//   A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[co = 16 rb + i][ci = 8 chunk + e][tap of slot (step, lane >> 4)])   (slots without a tap: zeros);
// then scale[cout] * 2^-kw, shift[cout].
extern "C" int casmvs_conv2d_k5s2_splitf16_pack(int cin, int cout, const float *weight, const float *scale, const float *shift, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "conv2d_k5s2_splitf16_pack: null pointer");
  CASMVS_REQUIRE(casmvs_conv2d_k5s2_splitf16_packed_bytes(cin, cout) != 0, "conv2d_k5s2_splitf16_pack: %d -> %d (8 -> 16 or 16 -> 32)", cin, cout);
  float wmax = 0.0f;
  for (int i = 0; i < cout * cin * 25; ++i) {
    CASMVS_REQUIRE(std::isfinite(weight[i]), "conv2d_k5s2_splitf16_pack: weight %d is not finite", i);
    wmax = std::fmax(wmax, std::fabs(weight[i]));
  }
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
  const int kw = 14 - ex;
  const int nch = cin / 8, nrb = cout / 16;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int chunk = 0; chunk < nch; ++chunk)
    for (int s = 0; s < 8; ++s)
      for (int rb = 0; rb < nrb; ++rb) {
        uint16_t img[2][64][8];
        for (int l = 0; l < 64; ++l) {
          const int co = 16 * rb + (l & 15);
          const K5Tap t = k5_tap(s, l >> 4);
          for (int e = 0; e < 8; ++e) {
            const int ci = 8 * chunk + e;
            const float w = t.live ? std::ldexp(weight[((size_t)co * cin + ci) * 25 + t.ky * 5 + t.kx], kw) : 0.0f;
            const float a = (float)(_Float16)w;
            img[0][l][e] = f16_bits_k5(w);
            img[1][l][e] = f16_bits_k5(w - a);
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

extern "C" int casmvs_conv2d_k5s2_splitf16_supported(int cin, int cout, int H, int W) {
  return casmvs_conv2d_k5s2_splitf16_packed_bytes(cin, cout) != 0 && H % 2 == 0 && H >= 2 && W % 4 == 0 && W >= 4;
}

extern "C" int casmvs_conv2d_k5s2_splitf16_forward_f32(const void *packed, const float *in, float *out, int N, int cin, int cout, int H, int W, float slope,
                                                       void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "conv2d_k5s2_splitf16_forward: null pointer");
  CASMVS_REQUIRE(N > 0 && casmvs_conv2d_k5s2_splitf16_supported(cin, cout, H, W),
                 "conv2d_k5s2_splitf16_forward: N=%d %d -> %d H=%d W=%d (8 -> 16 or 16 -> 32, H even, W a multiple of 4)", N, cin, cout, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(packed)) & 15) == 0 && (reinterpret_cast<size_t>(out) & 3) == 0,
                 "conv2d_k5s2_splitf16_forward: 16-byte aligned input and image");
  CASMVS_REQUIRE((size_t)cin * H * W < ((size_t)1 << 29), "conv2d_k5s2_splitf16_forward: one image's tensor must hold < 2^29 floats");
  if (cin == 8) return launch_k5<8, 16>(packed, in, out, N, H, W, slope, (hipStream_t)stream);
  return launch_k5<16, 32>(packed, in, out, N, H, W, slope, (hipStream_t)stream);
}
