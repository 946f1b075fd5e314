// CostRegNet.conv11 = ConvTranspose3d(16 -> 8, k3 s2 p1, output_padding 1, no bias) + ABN + leaky-relu, then `conv0 + ...` (models/mvsnet.py:84-86,
// 101) on the f16 matrix cores in the float32-grade split arithmetic of conv0_splitf16.hip.
//
// Written at the end of round 3 without a GPU run (CPU emulation only: tests/hipemu); correct on its first launch in round 4
// (profiles/r04_native_checks_first_run.txt, tools/native/deconv11_check.cpp) and since then the engine's default for this layer.
//
// Why.  deconv16_kernel runs this layer on the float32 MFMA at 370 us for the 8 x 32 x 256 x 320 volume (level 1, batch 8) - 0.9 ms of
// the 8.5 ms step over the three levels - against 190 us of HBM time for what it moves (the 16-channel input at 1/8 of the voxels = 2 n
// floats, the skip tensor 8 n, the output 8 n): it is bound by the float32 matrix rate, not by memory.  The input is SMALL; only the
// matrix work is expensive - the case for the f16 cores: three f16 products cost 3/16 of the float32 matrix time.
//
// Form.  out[o] += in[i] w[k] with o = 2 i - 1 + k per axis: an even output coordinate takes tap k = 1 from i = o / 2, an odd one taps
// k = 0 from (o + 1) / 2 and k = 2 from (o - 1) / 2.  Along x the two parities are the two halves of the MFMA rows, as in conv0's PX form:
//   rows    i = (output channel co = i >> 1, x parity px = i & 1)
//   columns j = 16 consecutive input x positions ix (output x = 2 ix + px)
//   K       k = (dx = k >> 4, ci = k & 15): the input voxels ix + dx, dx in {0, 1}, all 16 input channels;
//             weights: (px 0, dx 0) -> kx 1, (px 1, dx 0) -> kx 2, (px 1, dx 1) -> kx 0, (px 0, dx 1) -> zero
// so one MFMA covers 32 output x of one (z, y) row for one (kz, ky) pair.  A row needs 1 / 2 / 2 / 4 such pairs by the parities of
// (z, y): 2.25 on average.  Workgroup = 256 threads, output tile 4 x 8 x 32; wave w owns output rows y0 + 2 w, y0 + 2 w + 1 of all four z:
// 18 (kz, ky, row) sets x 3 partial products = 54 MFMAs from 12 + 18 LDS reads.  The staged input box is 3 x 5 x 18 voxels x 16 channels
// (17 KiB as two float16 slices), all 9 lane images 18 KiB: the kernel is a stream of skip loads and output stores (64 KiB per tile).
// Arithmetic: x' = 2^kx x per tile, two float16 slices, aa + ab + ba, float32 accumulation; y = lrelu(acc 2^-kx scale + shift) + skip.
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

struct DcCfg {
  static constexpr int THREADS = 256;
  static constexpr int TZ = 4, TY = 8, TX = 32;                       // output tile
  static constexpr int JZ = TZ / 2 + 1, JY = TY / 2 + 1, JX = TX / 2 + 2;   // input box 3 x 5 x 18 (17 needed; pairs)
  static constexpr int NBOX = JZ * JY * JX;                           // 270 voxels
  static constexpr int NVOX = 272;                                    // units per (slice, channel half) plane: a multiple of 16 (bank-aligned planes)
  static constexpr int ITEMS = JZ * JY * (JX / 2);                    // (z, y, pair of x): 135 of the 256 threads
  static constexpr int WUNITS = 9 * 2 * 64;                           // lane images [kz * 3 + ky][slice][lane]
  static constexpr size_t ACT_BYTES = (size_t)4 * NVOX * 16, W_BYTES = (size_t)WUNITS * 16;   // 17 408 + 18 432
  static constexpr size_t LDS_USED = ACT_BYTES + W_BYTES + 16;
  static constexpr size_t LDS_BYTES = LDS_USED < CASMVS_SF_LDS_FLOOR ? (size_t)CASMVS_SF_LDS_FLOOR : LDS_USED;    // at most two workgroups per CU
};

__device__ __forceinline__ f32x4 dc_mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// in (B, 16, Di, Hi, Wi) float32 (Wi even, 8-byte aligned); skip (B, 8, 2 Di, 2 Hi, 2 Wi) or nullptr; out like skip.
// wpk: [kz * 3 + ky][slice][lane] 16-byte lane images, then scale[8] (ABN scale x 2^-kw), shift[8].
__global__ __launch_bounds__(DcCfg::THREADS, 2) void deconv11_sf_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                                       const float *__restrict__ skip, float *__restrict__ out, int B, int Di, int Hi,
                                                                       int Wi, int tiles_x, int tiles_y, int tiles_z, float slope) {
  using Cfg = DcCfg;
  constexpr int NVOX = Cfg::NVOX, JY = Cfg::JY, JX = Cfg::JX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                          // [slice][half][NVOX]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                          // [9][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, kb = lane >> 4, half = kb & 1, dx = kb >> 1, u = kb;
  const int total = tiles_x * tiles_y * tiles_z * B;
  if ((int)blockIdx.x >= total) return;
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const int iHW = Hi * Wi, ics = Di * iHW, oHW = Ho * Wo, ocs = Do * oHW;
  const size_t in_ss = (size_t)16 * ics, out_ss = (size_t)8 * ocs;
  const float *tail = reinterpret_cast<const float *>(wpk + Cfg::W_BYTES);
  float sc[2], sh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sc[h] = tail[2 * u + h];
    sh[h] = tail[8 + 2 * u + h];
  }
  for (int unit = tid; unit < Cfg::WUNITS; unit += Cfg::THREADS) wl[unit] = reinterpret_cast<const u32x4 *>(wpk)[unit];
  const rsrc_t none = make_rsrc(in, 0);

  // lane's B unit (slice 0) of input row (izl, iyr) with iyr = 0 / 1 <-> box row wave / wave + 1:  plane `half`, voxel (izl, wave + iyr, jcol + dx)
  int vb[Cfg::JZ][2];
#pragma unroll
  for (int izl = 0; izl < Cfg::JZ; ++izl)
#pragma unroll
    for (int iyr = 0; iyr < 2; ++iyr) vb[izl][iyr] = half * NVOX + (izl * JY + wave + iyr) * JX + jcol + dx;

  struct Tile {
    int tx0, ty0, tz0, b;
  };
  auto decode = [&](int v) {
    int item = xcd_major(v, total);   // x fastest, then z, then y
    Tile t;
    t.tx0 = (item % tiles_x) * Cfg::TX;
    item /= tiles_x;
    t.tz0 = (item % tiles_z) * Cfg::TZ;
    item /= tiles_z;
    t.ty0 = (item % tiles_y) * Cfg::TY;
    t.b = item / tiles_y;
    return t;
  };
  // staging item e = tid -> (box plane, box row, pair of x); output-tile origins are even: the box starts at (tz0 / 2, ty0 / 2, tx0 / 2)
  const int e_iz = tid / (JY * (JX / 2)), e_rem = tid - e_iz * (JY * (JX / 2)), e_iy = e_rem / (JX / 2), e_g = e_rem - e_iy * (JX / 2);
  const int vox = tid < Cfg::ITEMS ? (e_iz * JY + e_iy) * JX + 2 * e_g : -1;
  int voff;
  auto plan = [&](const Tile &t) {
    const int gz = t.tz0 / 2 + e_iz, gy = t.ty0 / 2 + e_iy, gx = t.tx0 / 2 + 2 * e_g;
    const bool ok = tid < Cfg::ITEMS && gz < Di && gy < Hi && gx < Wi;   // Wi even: the pair is inside or outside; beyond the end = the zero padding
    voff = ok ? (gz * iHW + gy * Wi + gx) * 4 : kOOB;
  };
  f32x2 R[16];
  auto prefetch = [&](const Tile &t, bool exists) {
    const rsrc_t src = exists ? make_rsrc(in + (size_t)t.b * in_ss, in_ss * 4) : none;
#pragma unroll
    for (int c = 0; c < 16; ++c) R[c] = buf_load2(src, voff, c * ics * 4);
  };

  int item = blockIdx.x;
  Tile cur = decode(item);
  plan(cur);
  prefetch(cur, true);
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    const Tile nxt = have_next ? decode(next_item) : cur;
    // ---- the skip values of this tile's outputs, in flight under everything below: rows (zl, yo) x channel pair h ----
    const rsrc_t ssrc = skip ? make_rsrc(skip + (size_t)cur.b * out_ss, out_ss * 4) : none;
    const rsrc_t dst = make_rsrc(out + (size_t)cur.b * out_ss, out_ss * 4);
    int ooff[4][2];
    f32x2 SK[4][2][2];
#pragma unroll
    for (int zl = 0; zl < 4; ++zl)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) {
        const int oz = cur.tz0 + zl, oy = cur.ty0 + 2 * wave + yo, ox = cur.tx0 + 2 * jcol;
        const bool ok = oz < Do && oy < Ho && ox < Wo;   // Wo even
        ooff[zl][yo] = ok ? ((2 * u) * ocs + (oz * Ho + oy) * Wo + ox) * 4 : kOOB;
#pragma unroll
        for (int h = 0; h < 2; ++h) SK[zl][yo][h] = buf_load2(ssrc, ooff[zl][yo], h * ocs * 4);
      }
    // ---- the staged box's largest magnitude ----
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) m = casmvs::absmax3(m, R[c][0], R[c][1]);
    const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
    if (lane == 0) wmax[wave] = wm;
    __syncthreads();   // every wave is done with the previous tile's LDS; the four maxima are visible
    float mult, inv;
    casmvs::tile_scale(wmax, mult, inv);
    if (vox >= 0) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float x[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] = R[hf * 8 + c][p];
          u32x4 o[2];
          casmvs::split8_f16(x, mult, o);
#pragma unroll
          for (int s = 0; s < 2; ++s) act[(s * 2 + hf) * NVOX + vox + p] = o[s];
        }
    }
    __syncthreads();
    plan(nxt);
    prefetch(nxt, have_next);
    // ---- matrix phase: the wave's 3 x 2 input rows read once, then per (kz, ky) the lane image and its two (z, y) rows ----
    __builtin_amdgcn_sched_barrier(0);
    u32x4 rowv[Cfg::JZ][2][2];
#pragma unroll
    for (int izl = 0; izl < Cfg::JZ; ++izl)
#pragma unroll
      for (int iyr = 0; iyr < 2; ++iyr)
#pragma unroll
        for (int s = 0; s < 2; ++s) rowv[izl][iyr][s] = act[s * 2 * NVOX + vb[izl][iyr]];
    f32x4 acc[4][2];   // [zl][yo]: output rows (tz0 + zl, ty0 + 2 wave + yo)
#pragma unroll
    for (int zl = 0; zl < 4; ++zl)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) acc[zl][yo] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        u32x4 a[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) a[s] = wl[((kz * 3 + ky) * 2 + s) * 64 + lane];
        // tap k = 1: even outputs, input o / 2; k = 0: odd outputs, input (o + 1) / 2; k = 2: odd outputs, input (o - 1) / 2
        const int yo = ky == 1 ? 0 : 1, iyr = ky == 0 ? 1 : 0;
        constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int zl = kz == 1 ? 2 * q : 2 * q + 1;                        // the two output planes of the tile that take tap kz
          const int izl = kz == 1 ? q : (kz == 0 ? q + 1 : q);
#pragma unroll
          for (int p = 0; p < 3; ++p) acc[zl][yo] = dc_mfma(a[PA[p]], rowv[izl][iyr][PB[p]], acc[zl][yo]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: y = lrelu(acc 2^-kx scale + shift) + skip; lane holds rows 4 u + r = (co = 2 u + (r >> 1), x parity r & 1), column j ----
#pragma unroll
    for (int zl = 0; zl < 4; ++zl)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v0 = fmaf(acc[zl][yo][2 * h] * inv, sc[h], sh[h]), v1 = fmaf(acc[zl][yo][2 * h + 1] * inv, sc[h], sh[h]);
          v0 = v0 > 0.0f ? v0 : v0 * slope;
          v1 = v1 > 0.0f ? v1 : v1 * slope;
          buf_store2(f32x2{v0 + SK[zl][yo][h][0], v1 + SK[zl][yo][h][1]}, dst, ooff[zl][yo], h * ocs * 4);
        }
    if (!have_next) break;
    item = next_item;
    cur = nxt;
  }
}

inline uint16_t f16_bits_dc(float x) {
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

}  // namespace

extern "C" size_t casmvs_deconv11_splitf16_packed_bytes(void) { return DcCfg::W_BYTES + 16 * sizeof(float); }

// HOST-side packing: weight (16, 8, 3, 3, 3) float32 (ConvTranspose3d layout: cin, cout, kz, ky, kx) -> w' = 2^kw w (max |w'| in [2^13, 2^14)); per
// (kz, ky), per slice (f16(w'), f16(w' - f16(w'))), per lane the 8 float16 values
//   A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[ci = 8 ((lane >> 4) & 1) + e][co = i >> 1][kz][ky][kx]),  kx from (px = i & 1, dx = lane >> 5):
//   (0, 0) -> 1, (1, 0) -> 2, (1, 1) -> 0, (0, 1) -> no tap (zero);  then scale[8] * 2^-kw, shift[8].
extern "C" int casmvs_deconv11_splitf16_pack(const float *weight, const float *scale, const float *shift, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "deconv11_splitf16_pack: null pointer");
  float wmax = 0.0f;
  for (int i = 0; i < 16 * 8 * 27; ++i) {
    CASMVS_REQUIRE(std::isfinite(weight[i]), "deconv11_splitf16_pack: weight %d is not finite", i);
    wmax = std::fmax(wmax, std::fabs(weight[i]));
  }
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
  const int kw = 14 - ex;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int r9 = 0; r9 < 9; ++r9) {
    uint16_t img[2][64][8];
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, co = i >> 1, px = i & 1, kb = l >> 4, dxx = kb >> 1;
      const int kx = (px == 0) ? (dxx == 0 ? 1 : -1) : (dxx == 0 ? 2 : 0);
      for (int e = 0; e < 8; ++e) {
        const int ci = 8 * (kb & 1) + e;
        const float w = kx >= 0 ? std::ldexp(weight[(((size_t)ci * 8 + co) * 9 + r9) * 3 + kx], kw) : 0.0f;
        const float a = (float)(_Float16)w;
        img[0][l][e] = f16_bits_dc(w);
        img[1][l][e] = f16_bits_dc(w - a);
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

extern "C" int casmvs_deconv11_splitf16_supported(int Wi) { return Wi % 2 == 0 && Wi >= 2; }

extern "C" int casmvs_deconv11_splitf16_forward_f32(const void *packed, const float *in, const float *skip, float *out, int B, int Di, int Hi, int Wi,
                                                    float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "deconv11_splitf16_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && Di > 0 && Hi > 0 && casmvs_deconv11_splitf16_supported(Wi), "deconv11_splitf16_forward: B=%d Di=%d Hi=%d Wi=%d (Wi even)", B, Di, Hi, Wi);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(skip)) & 7) == 0 && (reinterpret_cast<size_t>(packed) & 15) == 0,
                 "deconv11_splitf16_forward: 8-byte aligned tensors, 16-byte aligned image");
  CASMVS_REQUIRE((size_t)64 * Di * Hi * Wi < ((size_t)1 << 29), "deconv11_splitf16_forward: one sample's output tensor must hold < 2^29 floats");
  using Cfg = DcCfg;
  const int tiles_x = casmvs::ceil_div(2 * Wi, Cfg::TX), tiles_y = casmvs::ceil_div(2 * Hi, Cfg::TY), tiles_z = casmvs::ceil_div(2 * Di, Cfg::TZ);
  const long total = (long)tiles_x * tiles_y * tiles_z * B;
  CASMVS_REQUIRE(total < (1L << 31), "deconv11_splitf16_forward: too many tiles");
  auto kernel = deconv11_sf_kernel;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "deconv11_sf_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, (hipStream_t)stream, in,
                     reinterpret_cast<const unsigned char *>(packed), skip, out, B, Di, Hi, Wi, tiles_x, tiles_y, tiles_z, slope);
  return casmvs::check_launch("deconv11_sf_kernel");
}
