// FeatureNet.conv0 = ConvBnReLU(3, 8, 3) -> ConvBnReLU(8, 8, 3) at full resolution (models/mvsnet.py:14-16, modules.py:8-18) as ONE kernel.
//
// *** Written at the end of round 3 WITHOUT a GPU run (the round's GPU minutes were spent): an opt-in entry point, not what the engine
// *** calls.  tools/native/fnet_conv0_check.cpp is its first test (against the two layer launches it replaces and a float64 loop).
//
// Why.  The two layers run as separate float32 MFMA launches at 156 + 139 us for 24 images of 512 x 640 (profiles/r03_step_runner_timeline.txt):
// the 3-channel image is 94 MB, the 8-channel maps 251 MB each - 345 + 502 MB of traffic for 43 + 63 us of HBM time - and the
// first layer's 3 input channels make a poor matrix problem.  Fused: read the image, write conv0's output (345 MB), the 8-channel
// intermediate lives in LDS only.
//   * the image tile (3 channels, 20 x 42 with the two halos) is staged in LDS as float32;
//   * conv0.0 is vector-ALU work: a thread owns 4 consecutive x of one staged row and all 8 channels - 27 taps x 32 outputs, the six
//     inputs of a (channel, ky) row as one 16-byte + one 8-byte LDS read, the weights as broadcast reads - then ABN + leaky-relu, and ZERO
//     where the position lies outside the image (it is conv0.1's padding, not a convolution result);
//   * those 32 values are what conv0_splitf16.hip's staging gets from memory: per-tile power-of-two scale, two float16 slices, LDS;
//   * conv0.1 is conv0_splitf16.hip's matrix form with one z tap: rows = (8 output channels x 2 x phases), K = 4 x offsets x 8 channels,
//     3 (ky) x 4 (rows of the wave) x 3 partial products on v_mfma_f32_16x16x32_f16, float32 accumulation; ABN + leaky-relu; store.
// Output tile 16 x 32, persistent 256-thread workgroups, two per CU (LDS request padded to 56 KiB: DESIGN.md 2.0).  Between a wave's
// matrix instructions only LDS reads are issued.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "buffer_ops.h"
#include "common.h"
#include "split_f16.h"

#ifndef CASMVS_F0_SCALAR_FMA
#define CASMVS_F0_SCALAR_FMA 0   // A/B builds: 1 = conv0.0 with scalar v_fma_f32 instead of packed v_pk_fma_f32
#endif

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct F0Cfg {
  static constexpr int THREADS = 256, NT = 4;
  static constexpr int TY = 16, TX = 32;
  static constexpr int IY = TY + 2, IX = TX + 8, ROW = IX + 1;     // conv0.0's outputs as conv0.1 stages them: rows y0 - 1 .. y0 + 16, x0 - 4 .. x0 + 35
  static constexpr int NV = IY * ROW;                                // 16-byte slots per slice: 738
  static __host__ __device__ constexpr int slot(int x) { return x ^ (((x >> 3) & 1) << 1); }   // as SfCfg::slot
  static constexpr int ITEMS = IY * (IX / 4);                        // (row, group of 4 x): 180 of the 256 threads
  static constexpr int JY = TY + 4, JX = 48;                         // image tile: rows y0 - 2 .. y0 + 17; float index 0 = x0 - 5 (row stride 48)
  static constexpr int JG = 12;                                      // aligned 4-float groups loaded per row: x0 - 8 .. x0 + 39
  static constexpr int JITEMS = 3 * JY * JG;                         // 720
  static constexpr int NJ = (JITEMS + THREADS - 1) / THREADS;        // 3 loads per thread
  // packed image: [conv0.1 lane images 3 x 2 x 64 x 16 B][scale1 8 | shift1 8][conv0.0 weights [ci][ky][kx][co] 216][scale0 8 | shift0 8] floats
  static constexpr int W1_UNITS = 3 * 2 * 64;
  static constexpr size_t W1_BYTES = (size_t)W1_UNITS * 16;          // 6144
  static constexpr int TAIL_FLOATS = 16 + 216 + 16;                  // 248
  static constexpr size_t PACKED_BYTES = W1_BYTES + (size_t)TAIL_FLOATS * 4;   // 7136
  static constexpr size_t ACT_BYTES = (size_t)2 * NV * 16;           // 23 616
  static constexpr size_t IMG_BYTES = (size_t)3 * JY * JX * 4;       // 11 520
  static constexpr size_t LDS_USED = ACT_BYTES + IMG_BYTES + PACKED_BYTES + 16;
  static constexpr size_t LDS_BYTES = LDS_USED < 56 * 1024 ? (size_t)56 * 1024 : LDS_USED;   // at most two workgroups per CU
};

__device__ __forceinline__ f32x4 f0_mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// imgs (N, 3, H, W) float32, W % 4 == 0, 16-byte aligned; packed: casmvs_fnet_conv0_fused_pack; out (N, 8, H, W)
__global__ __launch_bounds__(F0Cfg::THREADS, 2) void fnet_conv0_fused_kernel(const float *__restrict__ imgs, const unsigned char *__restrict__ packed,
                                                                            float *__restrict__ out, int N, int H, int W, int tiles_x, int tiles_y, float slope) {
  using Cfg = F0Cfg;
  constexpr int NT = Cfg::NT, IX = Cfg::IX, ROW = Cfg::ROW, NV = Cfg::NV, JY = Cfg::JY, JX = Cfg::JX, JG = Cfg::JG, NJ = Cfg::NJ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                                   // [2][NV]
  float *img = reinterpret_cast<float *>(smem_raw + Cfg::ACT_BYTES);                                  // [3][JY][JX]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES + Cfg::IMG_BYTES);                  // conv0.1 lane images [ky][slice][64]
  const float *tail = reinterpret_cast<const float *>(smem_raw + Cfg::ACT_BYTES + Cfg::IMG_BYTES + Cfg::W1_BYTES);   // scale1 | shift1 | w0 | scale0 | shift0
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::IMG_BYTES + Cfg::PACKED_BYTES);      // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, u = lane >> 4;
  const int total = tiles_x * tiles_y * N;
  if ((int)blockIdx.x >= total) return;
  const int HW = H * W;
  // the whole packed image -> LDS (446 16-byte units)
  for (int unit = tid; unit < (int)(Cfg::PACKED_BYTES / 16); unit += Cfg::THREADS)
    reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES + Cfg::IMG_BYTES)[unit] = reinterpret_cast<const u32x4 *>(packed)[unit];
  const rsrc_t none = make_rsrc(imgs, 0);
  const int vbase = (4 * wave) * ROW + Cfg::slot(2 * jcol + u + 3);

  // this thread's conv0.0 item: staged row iy (image row y0 - 1 + iy), x group g (x0 - 4 + 4 g ..)
  const int it_iy = tid / (IX / 4), it_g = tid - it_iy * (IX / 4);
  const bool has_item = tid < Cfg::ITEMS;
  const int vox = it_iy * ROW + 4 * it_g, vxor = ((it_g >> 1) & 1) << 1;

  auto decode = [&](int v, int &n, int &ty0, int &tx0) {
    int item = xcd_major(v, total);   // x fastest, then y, then image
    tx0 = (item % tiles_x) * Cfg::TX;
    item /= tiles_x;
    ty0 = (item % tiles_y) * Cfg::TY;
    n = item / tiles_y;
  };
  // image-tile loads: item e = tid + 256 r -> (channel, row, aligned group of 4 x starting at x0 - 8 + 4 g)
  int joff[NJ];
  auto plan = [&](int ty0, int tx0) {
#pragma unroll
    for (int r = 0; r < NJ; ++r) {
      const int e = tid + r * Cfg::THREADS;
      const int c = e / (JY * JG), rem = e - c * (JY * JG), iy = rem / JG, g = rem - iy * JG;
      const int gy = ty0 - 2 + iy, gx = tx0 - 8 + 4 * g;
      const bool ok = e < Cfg::JITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 4 == 0
      joff[r] = ok ? ((c * H + gy) * W + gx) * 4 : kOOB;
    }
  };
  f32x4v J[NJ];
  auto prefetch = [&](int n, bool exists) {
    const rsrc_t src = exists ? make_rsrc(imgs + (size_t)n * 3 * HW, (size_t)3 * HW * 4) : none;
#pragma unroll
    for (int r = 0; r < NJ; ++r) J[r] = buf_load4(src, joff[r], 0);
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
    // ---- image tile: registers -> LDS (float index 0 = x0 - 5: the loaded group g starts at index 4 g - 3) ----
#pragma unroll
    for (int r = 0; r < NJ; ++r) {
      const int e = tid + r * Cfg::THREADS;
      if (e < Cfg::JITEMS) {
        const int c = e / (JY * JG), rem = e - c * (JY * JG), iy = rem / JG, g = rem - iy * JG;
        float *dst = img + (c * JY + iy) * JX + 4 * g - 3;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * g - 3 + j >= 0) dst[j] = J[r][j];
      }
    }
    __syncthreads();   // B1: the image tile (and, the first time, the packed image) is visible
    // ---- conv0.0 on the vector ALU: 4 x positions x 8 channels per thread ----
    float R[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) R[c][j] = 0.0f;
    if (has_item) {
      const float *w0 = tail + 16;
      // (ci, ky) as a rolled loop of 9: fully unrolled, the compiler hoisted all 54 weight reads and spilled 440 bytes per lane
#pragma unroll 1
      for (int cik = 0; cik < 9; ++cik) {
          const int ci = cik / 3, ky = cik - 3 * ci;
          const float *rowp = img + (ci * JY + it_iy + ky) * JX + 4 * it_g;   // inputs x0 - 5 + 4 g .. + 5 = output x - 1 .. x + 4
          const f32x4v lo = *reinterpret_cast<const f32x4v *>(rowp);
          const f32x2 hi = *reinterpret_cast<const f32x2 *>(rowp + 4);
          const float in6[6] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1]};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float *wp = w0 + ((ci * 3 + ky) * 3 + kx) * 8;
            const f32x4v wa = *reinterpret_cast<const f32x4v *>(wp), wb = *reinterpret_cast<const f32x4v *>(wp + 4);
            const float w8[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#if CASMVS_F0_SCALAR_FMA
#pragma unroll
              for (int c = 0; c < 8; ++c) R[c][j] = fmaf(in6[j + kx], w8[c], R[c][j]);
#else
#pragma unroll
              for (int c = 0; c < 8; c += 2) {
                const f32x2 r = __builtin_elementwise_fma(f32x2{in6[j + kx], in6[j + kx]}, f32x2{w8[c], w8[c + 1]}, f32x2{R[c][j], R[c + 1][j]});
                R[c][j] = r[0];
                R[c + 1][j] = r[1];
              }
#endif
            }
          }
        }
      // ABN + leaky-relu of conv0.0; zero outside the image (conv0.1's zero padding)
      const float *s0 = tail + 16 + 216;
      const int gy = ty0 - 1 + it_iy, gx = tx0 - 4 + 4 * it_g;
      const bool row_in = gy >= 0 && gy < H;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float scl = s0[c], sft = s0[8 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = fmaf(R[c][j], scl, sft);
          v = v > 0.0f ? v : v * slope;
          R[c][j] = (row_in && gx + j >= 0 && gx + j < W) ? v : 0.0f;
        }
      }
    }
    // ---- the staged tile's largest magnitude ----
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(R[c][j]));
    const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
    if (lane == 0) wmax[wave] = wm;
    __syncthreads();   // B2: every wave is done with the image tile and with the previous tile's staged slices; the maxima are visible
    float mult, inv;
    casmvs::tile_scale(wmax, mult, inv);
    if (has_item) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = R[c][j];
        u32x4 o[2];
        casmvs::split8_f16(x, mult, o);
#pragma unroll
        for (int s = 0; s < 2; ++s) act[s * NV + vox + (j ^ vxor)] = o[s];
      }
    }
    __syncthreads();   // B3: the slices are visible
    plan(nty0, ntx0);
    prefetch(nn, have_next);
    // ---- conv0.1: 3 ky x 4 rows x 3 partial products; only LDS reads between the matrix instructions ----
    __builtin_amdgcn_sched_barrier(0);
    u32x4 row[NT + 2][2];
#pragma unroll
    for (int yr = 0; yr < NT + 2; ++yr)
#pragma unroll
      for (int s = 0; s < 2; ++s) row[yr][s] = act[s * NV + vbase + yr * ROW];
    f32x4 part[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) part[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      u32x4 a[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) a[s] = wl[(ky * 2 + s) * 64 + lane];
      constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < NT; ++t) part[t] = f0_mfma(a[PA[p]], row[t + ky][PB[p]], part[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: y = lrelu(part * 2^-kx * scale1 + shift1); lane holds rows 4 u + r = (co = 2 u + (r >> 1), x phase r & 1), column j ----
    const rsrc_t dst = make_rsrc(out + (size_t)n * 8 * HW, (size_t)8 * HW * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int oy = ty0 + 4 * wave + t, ox = tx0 + 2 * jcol;
      const bool ok = oy < H && ox < W;   // W even: the pixel pair is inside or outside
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float scl = tail[2 * u + h], sft = tail[8 + 2 * u + h];
        float v0 = fmaf(part[t][2 * h] * inv, scl, sft), v1 = fmaf(part[t][2 * h + 1] * inv, scl, sft);
        v0 = v0 > 0.0f ? v0 : v0 * slope;
        v1 = v1 > 0.0f ? v1 : v1 * slope;
        buf_store2(f32x2{v0, v1}, dst, ok ? ((2 * u + h) * HW + oy * W + ox) * 4 : kOOB, 0);
      }
    }
    if (!have_next) break;
    item = next_item;
    n = nn;
    ty0 = nty0;
    tx0 = ntx0;
  }
}

inline uint16_t f16_bits_f0(float x) {
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

}  // namespace

extern "C" size_t casmvs_fnet_conv0_fused_packed_bytes(void) { return F0Cfg::PACKED_BYTES; }

// HOST-side packing.  w0 (8, 3, 3, 3), w1 (8, 8, 3, 3): the torch weights of conv0.0 / conv0.1; scale / shift: their folded eval-mode ABN
// (nullptr = 1 / 0).  conv0.1 goes in as conv0_splitf16.hip packs a (kz, ky) pair: w' = 2^kw w with max |w'| in [2^13, 2^14), per ky and
// slice the lane image A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[co = i >> 1][ci = e][ky][kx = (lane >> 4) - (i & 1)]); scale1 carries 2^-kw.
extern "C" int casmvs_fnet_conv0_fused_pack(const float *w0, const float *scale0, const float *shift0, const float *w1, const float *scale1,
                                            const float *shift1, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(w0 && w1 && packed, "fnet_conv0_fused_pack: null pointer");
  float wmax = 0.0f;
  for (int i = 0; i < 8 * 8 * 9; ++i) {
    CASMVS_REQUIRE(std::isfinite(w1[i]), "fnet_conv0_fused_pack: conv0.1 weight %d is not finite", i);
    wmax = std::fmax(wmax, std::fabs(w1[i]));
  }
  for (int i = 0; i < 8 * 3 * 9; ++i) CASMVS_REQUIRE(std::isfinite(w0[i]), "fnet_conv0_fused_pack: conv0.0 weight %d is not finite", i);
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
  const int kw = 14 - ex;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int ky = 0; ky < 3; ++ky) {
    uint16_t img[2][64][8];
    for (int l = 0; l < 64; ++l) {
      const int i = l & 15, co = i >> 1, s = i & 1, uu = l >> 4, kx = uu - s;
      for (int e = 0; e < 8; ++e) {
        const float w = (kx >= 0 && kx <= 2) ? std::ldexp(w1[((co * 8 + e) * 3 + ky) * 3 + kx], kw) : 0.0f;
        const float a = (float)(_Float16)w;
        img[0][l][e] = f16_bits_f0(w);
        img[1][l][e] = f16_bits_f0(w - a);
      }
    }
    std::memcpy(p, img, sizeof(img));
    p += 2 * 64 * 8;
  }
  float *tail = reinterpret_cast<float *>(p);
  for (int c = 0; c < 8; ++c) tail[c] = std::ldexp(scale1 ? scale1[c] : 1.0f, -kw);
  for (int c = 0; c < 8; ++c) tail[8 + c] = shift1 ? shift1[c] : 0.0f;
  for (int ci = 0; ci < 3; ++ci)
    for (int k = 0; k < 9; ++k)
      for (int co = 0; co < 8; ++co) tail[16 + (ci * 9 + k) * 8 + co] = w0[(co * 3 + ci) * 9 + k];
  for (int c = 0; c < 8; ++c) tail[16 + 216 + c] = scale0 ? scale0[c] : 1.0f;
  for (int c = 0; c < 8; ++c) tail[16 + 216 + 8 + c] = shift0 ? shift0[c] : 0.0f;
  return CASMVS_OK;
}

extern "C" int casmvs_fnet_conv0_fused_supported(int W) { return W % 4 == 0 && W >= 4; }

extern "C" int casmvs_fnet_conv0_fused_f32(const void *packed, const float *imgs, float *out, int N, int H, int W, float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && imgs && out, "fnet_conv0_fused: null pointer");
  CASMVS_REQUIRE(N > 0 && H > 0 && casmvs_fnet_conv0_fused_supported(W), "fnet_conv0_fused: N=%d H=%d W=%d (W %% 4 == 0)", N, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(imgs) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(packed)) & 15) == 0, "fnet_conv0_fused: 16-byte aligned pointers");
  CASMVS_REQUIRE((size_t)8 * H * W < ((size_t)1 << 29), "fnet_conv0_fused: one image's output must hold < 2^29 floats");
  using Cfg = F0Cfg;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY);
  const long total = (long)tiles_x * tiles_y * N;
  CASMVS_REQUIRE(total < (1L << 31), "fnet_conv0_fused: too many tiles");
  auto kernel = fnet_conv0_fused_kernel;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "fnet_conv0_fused_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, (hipStream_t)stream, imgs,
                     reinterpret_cast<const unsigned char *>(packed), out, N, H, W, tiles_x, tiles_y, slope);
  return casmvs::check_launch("fnet_conv0_fused_kernel");
}
