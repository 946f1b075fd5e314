// CostRegNet.conv9 = ConvTranspose3d(32 -> 16, k3 s2 p1, output_padding 1, no bias) + ABN + leaky-relu, then `conv2 + ...` (models/mvsnet.py:80-82,
// 99) on the f16 matrix cores in the float32-grade split arithmetic of conv0_splitf16.hip.  The sibling of deconv11_splitf16.hip for 16 output channels.
//
// Written at the end of round 3 without a GPU run (CPU emulation only: tests/hipemu); correct on its first launch in round 4
// (profiles/r04_native_checks_first_run.txt, tools/native/deconv9_check.cpp) and since then the engine's default for this layer.
//
// Form.  out[o] += in[i] w[k], o = 2 i - 1 + k per axis (even o: k = 1 from o / 2; odd o: k = 0 from (o + 1) / 2 and k = 2 from (o - 1) / 2).
// With 16 output channels the MFMA rows are the channels; K = the 32 input channels of ONE input voxel; columns j = 16 consecutive input x
// positions ix.  Per (kz, ky) pair and output row three MFMA sets: kx = 1 (input ix -> even outputs 2 ix), kx = 2 (input ix -> odd outputs
// 2 ix + 1) and kx = 0 (input ix + 1 -> odd outputs 2 ix + 1): two accumulators per row, one per x parity, stored as (even, odd) pairs.
// Workgroup = 256 threads, output tile 2 x 8 x 32; wave w owns output rows y0 + 2 w, y0 + 2 w + 1 of both z: 9 (kz, ky) x 3 (kx) x 3 partial
// products = 81 MFMAs.  Input box 2 x 5 x 18 voxels x 32 channels as planes [slice][channel quarter][192] (24 KiB), lane images
// [kz * 3 + ky][kx][slice] 54 KiB: 78 KiB, two workgroups per CU.
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

struct D9Cfg {
  static constexpr int THREADS = 256;
  static constexpr int TZ = 2, TY = 8, TX = 32;                       // output tile
  static constexpr int JZ = TZ / 2 + 1, JY = TY / 2 + 1, JX = TX / 2 + 2;   // input box 2 x 5 x 18
  static constexpr int NVOX = 192;                                    // units per (slice, channel quarter) plane: 180 used, a multiple of 16
  static constexpr int ITEMS = 2 * JZ * JY * (JX / 2);                // (16-channel half, z, y, pair of x): 180 of the 256 threads
  static constexpr int WUNITS = 9 * 3 * 2 * 64;                       // lane images [kz * 3 + ky][kx][slice][lane]
  static constexpr size_t ACT_BYTES = (size_t)8 * NVOX * 16, W_BYTES = (size_t)WUNITS * 16;   // 24 576 + 55 296
  static constexpr size_t LDS_BYTES = ACT_BYTES + W_BYTES + 16;       // 79 888: two workgroups per CU, never three
};

__device__ __forceinline__ f32x4 d9_mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// in (B, 32, Di, Hi, Wi) float32 (Wi even, 8-byte aligned); skip (B, 16, 2 Di, 2 Hi, 2 Wi) or nullptr; out like skip.
// wpk: [kz * 3 + ky][kx][slice][lane] 16-byte lane images, then scale[16] (ABN scale x 2^-kw), shift[16].
__global__ __launch_bounds__(D9Cfg::THREADS, 2) void deconv9_sf_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                                      const float *__restrict__ skip, float *__restrict__ out, int B, int Di, int Hi,
                                                                      int Wi, int tiles_x, int tiles_y, int tiles_z, float slope) {
  using Cfg = D9Cfg;
  constexpr int NVOX = Cfg::NVOX, JY = Cfg::JY, JX = Cfg::JX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                          // [slice][quarter][NVOX]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                          // [9][3][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, kb = lane >> 4, u = kb;
  const int total = tiles_x * tiles_y * tiles_z * B;
  if ((int)blockIdx.x >= total) return;
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const int iHW = Hi * Wi, ics = Di * iHW, oHW = Ho * Wo, ocs = Do * oHW;
  const size_t in_ss = (size_t)32 * ics, out_ss = (size_t)16 * ocs;
  const float *tail = reinterpret_cast<const float *>(wpk + Cfg::W_BYTES);
  float sc[4], sh[4];   // the lane's result rows: output channels 4 u + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sc[r] = tail[4 * u + r];
    sh[r] = tail[16 + 4 * u + r];
  }
  for (int unit = tid; unit < Cfg::WUNITS; unit += Cfg::THREADS) wl[unit] = reinterpret_cast<const u32x4 *>(wpk)[unit];
  const rsrc_t none = make_rsrc(in, 0);

  // lane's B unit (slice 0) of input row (izl, iyr) [box row wave + iyr]: channel quarter kb, voxel (izl, wave + iyr, jcol); the kx = 0 tap reads one further
  int vb[Cfg::JZ][2];
#pragma unroll
  for (int izl = 0; izl < Cfg::JZ; ++izl)
#pragma unroll
    for (int iyr = 0; iyr < 2; ++iyr) vb[izl][iyr] = kb * NVOX + (izl * JY + wave + iyr) * JX + jcol;

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
  // staging item e = tid -> (16-channel half, box plane, box row, pair of x)
  constexpr int PAIRS = Cfg::JZ * JY * (JX / 2);   // 90
  const int e_hf = tid / PAIRS, e_pr = tid - e_hf * PAIRS;
  const int e_iz = e_pr / (JY * (JX / 2)), e_rem = e_pr - e_iz * (JY * (JX / 2)), e_iy = e_rem / (JX / 2), e_g = e_rem - e_iy * (JX / 2);
  const int vox = tid < Cfg::ITEMS ? (e_iz * JY + e_iy) * JX + 2 * e_g : -1;
  int voff;
  auto plan = [&](const Tile &t) {
    const int gz = t.tz0 / 2 + e_iz, gy = t.ty0 / 2 + e_iy, gx = t.tx0 / 2 + 2 * e_g;
    const bool ok = tid < Cfg::ITEMS && gz < Di && gy < Hi && gx < Wi;   // Wi even; beyond the end = the zero padding
    voff = ok ? (gz * iHW + gy * Wi + gx) * 4 : kOOB;
  };
  f32x2 R[16];
  auto prefetch = [&](const Tile &t, bool exists) {
    const rsrc_t src = exists ? make_rsrc(in + (size_t)t.b * in_ss, in_ss * 4) : none;
#pragma unroll
    for (int c = 0; c < 16; ++c) R[c] = buf_load2(src, voff, (e_hf * 16 + c) * ics * 4);
  };

  int item = blockIdx.x;
  Tile cur = decode(item);
  plan(cur);
  prefetch(cur, true);
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    const Tile nxt = have_next ? decode(next_item) : cur;
    // ---- the skip values of this tile's outputs, in flight under everything below: rows (zl, yo) x the lane's 4 channels ----
    const rsrc_t ssrc = skip ? make_rsrc(skip + (size_t)cur.b * out_ss, out_ss * 4) : none;
    const rsrc_t dst = make_rsrc(out + (size_t)cur.b * out_ss, out_ss * 4);
    int ooff[2][2];
    f32x2 SK[2][2][4];
#pragma unroll
    for (int zl = 0; zl < 2; ++zl)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo) {
        const int oz = cur.tz0 + zl, oy = cur.ty0 + 2 * wave + yo, ox = cur.tx0 + 2 * jcol;
        const bool ok = oz < Do && oy < Ho && ox < Wo;   // Wo even
        ooff[zl][yo] = ok ? ((4 * u) * ocs + (oz * Ho + oy) * Wo + ox) * 4 : kOOB;
#pragma unroll
        for (int r = 0; r < 4; ++r) SK[zl][yo][r] = buf_load2(ssrc, ooff[zl][yo], r * ocs * 4);
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
      for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float x[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) x[c] = R[hh * 8 + c][p];
          u32x4 o[2];
          casmvs::split8_f16(x, mult, o);
#pragma unroll
          for (int s = 0; s < 2; ++s) act[(s * 4 + e_hf * 2 + hh) * NVOX + vox + p] = o[s];
        }
    }
    __syncthreads();
    plan(nxt);
    prefetch(nxt, have_next);
    // ---- matrix phase ----
    __builtin_amdgcn_sched_barrier(0);
    u32x4 rowv[Cfg::JZ][2][2][2];   // [izl][iyr][input ix / ix + 1][slice]
#pragma unroll
    for (int izl = 0; izl < Cfg::JZ; ++izl)
#pragma unroll
      for (int iyr = 0; iyr < 2; ++iyr)
#pragma unroll
        for (int nx = 0; nx < 2; ++nx)
#pragma unroll
          for (int s = 0; s < 2; ++s) rowv[izl][iyr][nx][s] = act[s * 4 * NVOX + vb[izl][iyr] + nx];
    f32x4 acc[2][2][2];   // [zl][yo][x parity]: output rows (tz0 + zl, ty0 + 2 wave + yo)
#pragma unroll
    for (int zl = 0; zl < 2; ++zl)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo)
#pragma unroll
        for (int px = 0; px < 2; ++px) acc[zl][yo][px] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        // tap k = 1: even outputs, input o / 2; k = 0: odd outputs, input (o + 1) / 2; k = 2: odd outputs, input (o - 1) / 2
        const int zl = kz == 1 ? 0 : 1, izl = kz == 0 ? 1 : 0;
        const int yo = ky == 1 ? 0 : 1, iyr = ky == 0 ? 1 : 0;
        constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          u32x4 a[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) a[s] = wl[(((kz * 3 + ky) * 3 + kx) * 2 + s) * 64 + lane];
          const int px = kx == 1 ? 0 : 1, nx = kx == 0 ? 1 : 0;
#pragma unroll
          for (int p = 0; p < 3; ++p) acc[zl][yo][px] = d9_mfma(a[PA[p]], rowv[izl][iyr][nx][PB[p]], acc[zl][yo][px]);
        }
      }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: y = lrelu(acc 2^-kx scale + shift) + skip; lane holds rows 4 u + r = output channel, column j = input x -> outputs (2 j, 2 j + 1) ----
#pragma unroll
    for (int zl = 0; zl < 2; ++zl)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v0 = fmaf(acc[zl][yo][0][r] * inv, sc[r], sh[r]), v1 = fmaf(acc[zl][yo][1][r] * inv, sc[r], sh[r]);
          v0 = v0 > 0.0f ? v0 : v0 * slope;
          v1 = v1 > 0.0f ? v1 : v1 * slope;
          buf_store2(f32x2{v0 + SK[zl][yo][r][0], v1 + SK[zl][yo][r][1]}, dst, ooff[zl][yo], r * ocs * 4);
        }
    if (!have_next) break;
    item = next_item;
    cur = nxt;
  }
}

inline uint16_t f16_bits_d9(float x) {
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

}  // namespace

extern "C" size_t casmvs_deconv9_splitf16_packed_bytes(void) { return D9Cfg::W_BYTES + 32 * sizeof(float); }

// HOST-side packing: weight (32, 16, 3, 3, 3) float32 (ConvTranspose3d layout: cin, cout, kz, ky, kx) -> w' = 2^kw w (max |w'| in [2^13, 2^14)); per
// (kz, ky), per kx, per slice (f16(w'), f16(w' - f16(w'))), per lane the 8 float16 values
//   A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[ci = 8 (lane >> 4) + e][co = i][kz][ky][kx]);  then scale[16] * 2^-kw, shift[16].
extern "C" int casmvs_deconv9_splitf16_pack(const float *weight, const float *scale, const float *shift, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "deconv9_splitf16_pack: null pointer");
  float wmax = 0.0f;
  for (int i = 0; i < 32 * 16 * 27; ++i) {
    CASMVS_REQUIRE(std::isfinite(weight[i]), "deconv9_splitf16_pack: weight %d is not finite", i);
    wmax = std::fmax(wmax, std::fabs(weight[i]));
  }
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
  const int kw = 14 - ex;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int r9 = 0; r9 < 9; ++r9)
    for (int kx = 0; kx < 3; ++kx) {
      uint16_t img[2][64][8];
      for (int l = 0; l < 64; ++l) {
        const int co = l & 15, kbb = l >> 4;
        for (int e = 0; e < 8; ++e) {
          const int ci = 8 * kbb + e;
          const float w = std::ldexp(weight[(((size_t)ci * 16 + co) * 9 + r9) * 3 + kx], kw);
          const float a = (float)(_Float16)w;
          img[0][l][e] = f16_bits_d9(w);
          img[1][l][e] = f16_bits_d9(w - a);
        }
      }
      std::memcpy(p, img, sizeof(img));
      p += 2 * 64 * 8;
    }
  float *tail = reinterpret_cast<float *>(p);
  for (int c = 0; c < 16; ++c) tail[c] = std::ldexp(scale ? scale[c] : 1.0f, -kw);
  for (int c = 0; c < 16; ++c) tail[16 + c] = shift ? shift[c] : 0.0f;
  return CASMVS_OK;
}

extern "C" int casmvs_deconv9_splitf16_supported(int Wi) { return Wi % 2 == 0 && Wi >= 2; }

extern "C" int casmvs_deconv9_splitf16_forward_f32(const void *packed, const float *in, const float *skip, float *out, int B, int Di, int Hi, int Wi,
                                                   float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "deconv9_splitf16_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && Di > 0 && Hi > 0 && casmvs_deconv9_splitf16_supported(Wi), "deconv9_splitf16_forward: B=%d Di=%d Hi=%d Wi=%d (Wi even)", B, Di, Hi, Wi);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(skip)) & 7) == 0 && (reinterpret_cast<size_t>(packed) & 15) == 0,
                 "deconv9_splitf16_forward: 8-byte aligned tensors, 16-byte aligned image");
  CASMVS_REQUIRE((size_t)128 * Di * Hi * Wi < ((size_t)1 << 29), "deconv9_splitf16_forward: one sample's output tensor must hold < 2^29 floats");
  using Cfg = D9Cfg;
  const int tiles_x = casmvs::ceil_div(2 * Wi, Cfg::TX), tiles_y = casmvs::ceil_div(2 * Hi, Cfg::TY), tiles_z = casmvs::ceil_div(2 * Di, Cfg::TZ);
  const long total = (long)tiles_x * tiles_y * tiles_z * B;
  CASMVS_REQUIRE(total < (1L << 31), "deconv9_splitf16_forward: too many tiles");
  auto kernel = deconv9_sf_kernel;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "deconv9_sf_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, (hipStream_t)stream, in,
                     reinterpret_cast<const unsigned char *>(packed), skip, out, B, Di, Hi, Wi, tiles_x, tiles_y, tiles_z, slope);
  return casmvs::check_launch("deconv9_sf_kernel");
}
