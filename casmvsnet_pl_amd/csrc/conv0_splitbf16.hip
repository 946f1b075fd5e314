// CostRegNet.conv0 (Conv3d Cin -> 8, k3 s1 p1 + folded ABN + leaky-relu: the full-resolution layer that holds 45-70 % of
// the network's FLOPs) on the bf16 matrix cores with float32-grade arithmetic: every float32 operand is the EXACT sum of
// three bf16 numbers, and the product of two such sums is accumulated from six bf16 x bf16 partial products in float32.
//
// Reference semantics: models/mvsnet.py:63,91 (`conv0`), models/modules.py:21-31 (ConvBnReLU3D).
//
// Why: the f32-input MFMA of gfx950 runs at the float32 VECTOR rate (157 TFLOP/s, 1/16 of the bf16 matrix rate) and
// shares the SIMD's issue with the VALU; conv16db_kernel<PX> sits at the ceiling of that formulation (0.55 of the peak:
// 78-85 % pipe-busy x 75 % useful rows).  v_mfma_f32_16x16x32_bf16 does 8x the K of v_mfma_f32_16x16x4_f32 in ~0.6x the
// time.  Splitting x = x_h + x_m + x_l (the three 8-bit slices of the 24-bit significand: two mask-and-subtract steps,
// exact) and w = w_h + w_m + w_l (host) gives
//     x w = x_h w_h + (x_h w_m + x_m w_h) + (x_h w_l + x_l w_h + x_m w_m) + [x_m w_l + x_l w_m + x_l w_l]
// where every product of two bf16 numbers is exact in float32 and the bracket is <= 2^-23 |x w| - the size of ONE float32
// rounding, of which a 216..864-term float32 dot product already contains hundreds.  Six MFMAs of K = 32 replace eight
// float32 MFMAs of K = 4: 2.2x less matrix-pipe time; the split costs ~5.5 VALU operations per staged element, once per
// workgroup (each staged element then feeds 27 taps x 8 output channels), and co-executes with the bf16 matrix pipe.
//
// Formulation (PX with 8 channels per K block): D[16 x 16] += A[16 x 32] B[32 x 16] with
//   rows    i = (co = i >> 1, x phase s = i & 1)          - 8 output channels x 2 x-phases (output x = x0 + 2 j + s)
//   columns j = 16 even output x of one (z, y) row
//   K       k = (u = k >> 3, ci = k & 7)                  - 4 input x offsets u (input x = x0 + 2 j + u - 1, kx = u - s:
//                                                           3 of the 4 are taps of a given row) x 8 input channels
// Lane l supplies the 8 consecutive k of block u = l >> 4 for row / column l & 15: for B that is ONE 16-byte LDS read of
// the 8 channels of one voxel from the channel-innermost bf16 tile [z][y][x][8]; for A one 16-byte read of a host-built
// lane image.  C/D: lane l, register r = D[4 (l >> 4) + r][l & 15] (as the f32 16x16x4 form).
//
// Workgroup = 512 threads (8 waves, 2 per SIMD), output tile 4 x 8 x 32 voxels = 32 column tiles, 4 per wave; input halo
// tile 6 x 10 x 40 voxels x 3 slices x 16 B = 115 KiB + 27 KiB of lane images (one chunk of 8 input channels) in LDS;
// persistent workgroups walk the tiles XCD-major; the next chunk's (or tile's) global loads are in flight during the
// MFMA phase.
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"

#ifndef CASMVS_SB_ABL
#define CASMVS_SB_ABL 0   // profiling builds only (WRONG results): 1 no MFMAs, 2 no split / LDS staging writes, 4 no global loads, 8 no tap reads
#endif

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct SbCfg {
  static constexpr int THREADS = 512, WAVES = 8, NT = 4;
  static constexpr int TZ = 4, TY = 8, TX = 32;
  static constexpr int IZ = TZ + 2, IY = TY + 2, IX = TX + 8;     // x0 - 4 .. x0 + 35 (16-byte aligned global groups)
  static constexpr int ROW = IX + 1;                                // 16-byte slots per staged row (odd: rows rotate through the banks)
  static constexpr int NV = IZ * IY * ROW;                          // slots per slice: 2460
  // slot of x inside its row = x ^ (((x >> 3) & 1) << 1): the tap reads (lane (j, u) -> x = 2 j + u + 3) stay at their
  // conflict-free 4 LDS cycles and the staging writes (lane -> x = 4 g + j, fixed j: slots = j mod 4 without the swizzle,
  // a 4-way conflict) drop to 1.75x their minimum (enumerated over paddings / swizzles: tests/kernel_model.py)
  static __host__ __device__ constexpr int slot(int x) { return x ^ (((x >> 3) & 1) << 1); }
  static constexpr int ITEMS = IZ * IY * (IX / 4);                  // (z, y, group of 4 x) staging items: 600
  static constexpr int NR = (ITEMS + THREADS - 1) / THREADS;        // staging rounds per thread: 2
  static constexpr int WUNITS = 9 * 3 * 64;                         // 16-byte units of a chunk's lane images: [kz * 3 + ky][slice][lane]
  static constexpr int NWL = (WUNITS + THREADS - 1) / THREADS;      // 4
  static constexpr size_t ACT_BYTES = (size_t)3 * NV * 16, W_BYTES = (size_t)WUNITS * 16;
  static constexpr size_t LDS_BYTES = ACT_BYTES + W_BYTES;          // 145 728
  static constexpr size_t chunk_bytes() { return W_BYTES; }
};

// ---- host: exact three-way bf16 split (truncation: slices of the significand) ---------------------------------------------
inline void split3(float x, uint16_t out[3]) {
  uint32_t b;
  float r = x;
  for (int s = 0; s < 3; ++s) {
    std::memcpy(&b, &r, 4);
    b &= 0xFFFF0000u;
    float part;
    std::memcpy(&part, &b, 4);
    out[s] = (uint16_t)(b >> 16);
    r = r - part;   // exact: the remaining low bits
  }
}

// ---- device: split 8 channels of one voxel into the three 16-byte bf16 vectors -------------------------------------------
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {   // (bf16 pattern of a) | (bf16 pattern of b) << 16
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ void split_voxel(const float (&x)[8], u32x4 (&o)[3]) {
  float hi[8], mid[8], lo[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    hi[c] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[c]) & 0xFFFF0000u);
    const float r = x[c] - hi[c];
    mid[c] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, r) & 0xFFFF0000u);
    lo[c] = r - mid[c];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    o[0][p] = pack_hi16(hi[2 * p], hi[2 * p + 1]);
    o[1][p] = pack_hi16(mid[2 * p], mid[2 * p + 1]);
    o[2][p] = pack_hi16(lo[2 * p], lo[2 * p + 1]);
  }
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct SbTile {
  int tx0, ty0, tz0, b;
};
__device__ __forceinline__ SbTile sb_decode(int v, int total, int tiles_x, int tiles_y, int tiles_z) {
  int item = xcd_major(v, total);   // z fastest, then x, then y (the halos of neighbouring tiles share an XCD's L2)
  SbTile t;
  t.tz0 = (item % tiles_z) * SbCfg::TZ;
  item /= tiles_z;
  t.tx0 = (item % tiles_x) * SbCfg::TX;
  item /= tiles_x;
  t.ty0 = (item % tiles_y) * SbCfg::TY;
  t.b = item / tiles_y;
  return t;
}

// in (B, CIN, D, H, W) float32, W % 4 == 0, 16-byte aligned; wpk: [chunk][kz * 3 + ky][slice][lane] 16-byte lane images, then
// scale[8], shift[8] (float32); out (B, 8, D, H, W).  TERMS: 6 (default) or 9 (all partial products: A/B of the accuracy).
template <int CIN, int TERMS>
__global__ __launch_bounds__(SbCfg::THREADS, 2) void conv0_sb_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                                    float *__restrict__ out, int B, int D, int H, int W, int tiles_x,
                                                                    int tiles_y, int tiles_z, float slope) {
  using Cfg = SbCfg;
  constexpr int NCH = CIN / 8, NT = Cfg::NT, NR = Cfg::NR, NWL = Cfg::NWL, IX = Cfg::IX, IY = Cfg::IY, NV = Cfg::NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                    // [3][NV]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);    // [9][3][64]
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

  // this wave's column tiles: (cz, cy_t); lane's B voxel (kz = ky = 0): ((cz) * IY + cy_t) * IX + 2 j + u + 3
  int vb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) vb[t] = ((wave >> 1) * IY + (wave & 1) * 4 + t) * Cfg::ROW + Cfg::slot(2 * jcol + u + 3);

  // staging plan of the current prefetch target: item e = tid + 512 r -> (iz, iy, 4-x group)
  int voff[NR], vox[NR], vxor[NR];
  auto plan = [&](const SbTile &tc) {
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
  auto prefetch = [&](const SbTile &tc, int chunk, bool exists) {   // every load of (tile, chunk); nothing here waits
    const rsrc_t src = exists ? make_rsrc(in + (size_t)tc.b * in_ss, in_ss * 4) : none;
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int unit = tid + i * Cfg::THREADS;
      WR[i] = __builtin_bit_cast(u32x4, buf_load4(exists ? wsrc : none, unit < Cfg::WUNITS ? unit * 16 : kOOB, chunk * (int)Cfg::W_BYTES));
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int c = 0; c < 8; ++c) R[r][c] = (CASMVS_SB_ABL & 4) ? f32x4v{1.f, (float)c, 2.f, 3.f} : buf_load4(src, voff[r], (chunk * 8 + c) * cs * 4);
  };
  auto commit = [&]() {   // registers -> LDS: the three bf16 slices of every staged voxel, the chunk's lane images
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int unit = tid + i * Cfg::THREADS;
      if (unit < Cfg::WUNITS) wl[unit] = WR[i];
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (vox[r] < 0 || ((CASMVS_SB_ABL & 2) && R[r][0][0] != 12345.f)) continue;   // (second round: 88 of the 512 threads)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = R[r][c][j];
        u32x4 o[3];
        split_voxel(x, o);
#pragma unroll
        for (int s = 0; s < 3; ++s) act[s * NV + vox[r] + (j ^ vxor[r])] = o[s];
      }
    }
  };

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  int item = blockIdx.x;
  SbTile cur = sb_decode(item, total, tiles_x, tiles_y, tiles_z);
  plan(cur);
  prefetch(cur, 0, true);
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    const SbTile nxt = have_next ? sb_decode(next_item, total, tiles_x, tiles_y, tiles_z) : cur;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      __syncthreads();   // every wave is done with the previous chunk's LDS
      commit();
      __syncthreads();
      if (ch + 1 < NCH) {
        prefetch(cur, ch + 1, true);
      } else {
        plan(nxt);
        prefetch(nxt, 0, have_next);
      }
      // ---- matrix phase: 9 (kz, ky) x 4 column tiles x TERMS partial products ----
#pragma unroll
      for (int r9 = 0; r9 < 9; ++r9) {
        const int kz = r9 / 3, ky = r9 % 3;
        u32x4 a[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = wl[(r9 * 3 + s) * 64 + lane];
        u32x4 bv[NT][3];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int s = 0; s < 3; ++s)
            bv[t][s] = (CASMVS_SB_ABL & 8) ? u32x4{(unsigned)r9, 1u, 2u, (unsigned)t} : act[s * NV + vb[t] + (kz * IY + ky) * Cfg::ROW];
        // partial products by decreasing magnitude class; consecutive MFMAs use different accumulators
        constexpr int PA[9] = {0, 0, 1, 0, 2, 1, 1, 2, 2}, PB[9] = {0, 1, 0, 2, 0, 1, 2, 1, 2};
#pragma unroll
        for (int p = 0; p < TERMS; ++p)
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (CASMVS_SB_ABL & 1) acc[t][0] += __builtin_bit_cast(float, a[PA[p]][0] ^ bv[t][PB[p]][1]);   // keeps the operands live
            else acc[t] = mfma_bf16(a[PA[p]], bv[t][PB[p]], acc[t]);
          }
      }
    }
    // ---- epilogue: y = lrelu(acc * scale + shift); lane holds rows 4 u + r = (co = 2 u + (r >> 1), x phase r & 1), column j ----
    const rsrc_t dst = make_rsrc(out + (size_t)cur.b * out_ss, out_ss * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int oz = cur.tz0 + (wave >> 1), oy = cur.ty0 + (wave & 1) * 4 + t, ox = cur.tx0 + 2 * jcol;
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

// lane-semantics probe of v_mfma_f32_16x16x32_bf16: D = A B for small integer matrices (exact in bf16)
__global__ void mfma_bf16_probe_kernel(float *out) {
  const int lane = threadIdx.x, i = lane & 15, kb = lane >> 4;
  union { u32x4 v; uint16_t h[8]; } a, b;
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * kb + e;
    // A[i][k] = (i + 1) if k == i (+ 16: a second diagonal), B[k][j] = 1 + k + 3 j (asymmetric): small integers, exact in bf16
    const float av = (k == i || k == i + 16) ? (float)(1 + i) : 0.0f, bvv = (float)(1 + k + 3 * i);
    a.h[e] = (uint16_t)(__builtin_bit_cast(unsigned, av) >> 16);
    b.h[e] = (uint16_t)(__builtin_bit_cast(unsigned, bvv) >> 16);
  }
  const f32x4 d = mfma_bf16(a.v, b.v, f32x4{0.f, 0.f, 0.f, 0.f});
  for (int r = 0; r < 4; ++r) out[r * 64 + lane] = d[r];
}

}  // namespace

extern "C" size_t casmvs_conv0_splitbf16_packed_bytes(int cin) {
  if (cin != 8 && cin != 16 && cin != 32) return 0;
  return (size_t)(cin / 8) * SbCfg::W_BYTES + 16 * sizeof(float);
}

// HOST-side packing: weight (8, cin, 3, 3, 3) float32 -> per chunk of 8 input channels, per (kz, ky), per slice, per lane the
// 8 bf16 values A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w[co = i >> 1][chunk * 8 + e][kz][ky][kx = (lane >> 4) - (i & 1)]);
// then scale[8], shift[8].
extern "C" int casmvs_conv0_splitbf16_pack(int cin, const float *weight, const float *scale, const float *shift, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "conv0_splitbf16_pack: null pointer");
  CASMVS_REQUIRE(cin == 8 || cin == 16 || cin == 32, "conv0_splitbf16_pack: cin=%d (8, 16 or 32)", cin);
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int ch = 0; ch < cin / 8; ++ch)
    for (int r9 = 0; r9 < 9; ++r9) {
      uint16_t img[3][64][8];
      for (int l = 0; l < 64; ++l) {
        const int i = l & 15, co = i >> 1, s = i & 1, uu = l >> 4, kx = uu - s;
        for (int e = 0; e < 8; ++e) {
          const float w = (kx >= 0 && kx <= 2) ? weight[(((size_t)co * cin + ch * 8 + e) * 9 + r9) * 3 + kx] : 0.0f;
          uint16_t sp[3];
          split3(w, sp);
          for (int q = 0; q < 3; ++q) img[q][l][e] = sp[q];
        }
      }
      std::memcpy(p, img, sizeof(img));
      p += 3 * 64 * 8;
    }
  float *tail = reinterpret_cast<float *>(p);
  for (int c = 0; c < 8; ++c) tail[c] = scale ? scale[c] : 1.0f;
  for (int c = 0; c < 8; ++c) tail[8 + c] = shift ? shift[c] : 0.0f;
  return CASMVS_OK;
}

extern "C" int casmvs_conv0_splitbf16_supported(int cin, int W) { return (cin == 8 || cin == 16 || cin == 32) && W % 4 == 0 && W >= 4; }

extern "C" int casmvs_conv0_splitbf16_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int D, int H, int W,
                                                  float slope, int terms, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "conv0_splitbf16_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && casmvs_conv0_splitbf16_supported(cin, W), "conv0_splitbf16_forward: B=%d cin=%d D=%d H=%d W=%d", B, cin, D, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(packed)) & 15) == 0, "conv0_splitbf16_forward: 16-byte aligned pointers");
  CASMVS_REQUIRE((size_t)cin * D * H * W < ((size_t)1 << 29), "conv0_splitbf16_forward: one sample's input tensor must hold < 2^29 floats");
  CASMVS_REQUIRE(terms == 0 || terms == 6 || terms == 9, "conv0_splitbf16_forward: terms=%d (0 = 6, 6 or 9)", terms);
  using Cfg = SbCfg;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY), tiles_z = casmvs::ceil_div(D, Cfg::TZ);
  const long total = (long)tiles_x * tiles_y * tiles_z * B;
  CASMVS_REQUIRE(total < (1L << 31), "conv0_splitbf16_forward: too many tiles");
  const unsigned char *wp = reinterpret_cast<const unsigned char *>(packed);
  hipStream_t st = (hipStream_t)stream;
#define CASMVS_SB(CIN, T)                                                                                                       \
  {                                                                                                                             \
    auto kernel = conv0_sb_kernel<CIN, T>;                                                                                      \
    if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "conv0_sb_kernel")) return rc; \
    const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);         \
    hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, in, wp, \
                       out, B, D, H, W, tiles_x, tiles_y, tiles_z, slope);                                                      \
  }
  const bool nine = terms == 9;
  if (cin == 8) { if (nine) CASMVS_SB(8, 9) else CASMVS_SB(8, 6) }
  else if (cin == 16) { if (nine) CASMVS_SB(16, 9) else CASMVS_SB(16, 6) }
  else { if (nine) CASMVS_SB(32, 9) else CASMVS_SB(32, 6) }
#undef CASMVS_SB
  return casmvs::check_launch("conv0_sb_kernel");
}

// D = A B with A[i][k] = (1 + i) [k == i or k == i + 16], B[k][j] = 1 + k + 3 j: D[i][j] = (1 + i) ((1 + i + 3 j) + (17 + i + 3 j)).
// Checks (a) the C/D map lane l, register r = D[4 (l >> 4) + r][l & 15] and (b) that element e of k-block kb of A meets element e of
// k-block kb of B - all the kernel relies on.
extern "C" int casmvs_selftest_mfma_bf16(float *dump) {
  casmvs::clear_error();
  float *d = nullptr;
  if (hipMalloc(&d, 4 * 64 * sizeof(float)) != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "selftest_mfma_bf16: hipMalloc failed");
  hipLaunchKernelGGL(mfma_bf16_probe_kernel, dim3(1), dim3(64), 0, 0, d);
  float h[4 * 64];
  hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "selftest_mfma_bf16: %s", hipGetErrorString(e));
  if (dump)
    for (int i = 0; i < 4 * 64; ++i) dump[i] = h[i];
  for (int r = 0; r < 4; ++r)
    for (int l = 0; l < 64; ++l) {
      const int i = 4 * (l >> 4) + r, j = l & 15;
      const float want = (float)(1 + i) * (float)((1 + i + 3 * j) + (17 + i + 3 * j));
      if (h[r * 64 + l] != want)
        return casmvs::fail(CASMVS_ERR_HIP, "selftest_mfma_bf16: reg=%d lane=%d: got %g want %g", r, l, h[r * 64 + l], want);
    }
  return CASMVS_OK;
}
