// CostRegNet's tail as ONE kernel: conv11 (ConvTranspose3d 16 -> 8 + ABN + leaky-relu) + the `conv0 + ...` skip (models/mvsnet.py:84-86, 101), the
// `prob` head (Conv3d 8 -> 1, :89, 104) and, when the depth range is one chunk, softmax / depth regression / confidence (:174-193).
//
// *** Written at the end of round 3 WITHOUT a GPU run: an opt-in entry point, nothing in the package calls it.  Its source runs correctly on the CPU
// *** under tests/hipemu (tests/hipemu/run_kernels3.cpp: against the layers in float64); its time on the MI355X is unknown.
//
// Why.  conv11 writes an 8-channel full-resolution tensor (8 n floats) that `prob` reads back at once (8 n): 16 n of the 28 n floats the pair moves,
// 0.9 + 0.6 ms of the 8.5 ms step (profiles/r03_step_runner_timeline.txt).  Here the depth-walking `prob` kernel (prob_regress.hip) gets its input
// planes from a PRODUCER inside the workgroup instead of from memory: the transposed convolution of deconv11_splitf16.hip (x parities on the MFMA
// rows, K = 2 input x positions x 16 channels, float32-grade split-f16 arithmetic) evaluated for the plane patch the walk needs next, the skip
// tensor added, the result written straight into the LDS slot in the layout the walk reads - which is the layout the MFMA leaves it in: a result
// lane holds (channel pair 2 u, 2 u + 1) x (positions x, x + 1) = one 16-byte staging item of prob_zwalk_kernel.
//
// Geometry.  A workgroup owns 62 x 8 output pixels (x0 = 62 k - 1, so that the halo patch x0 - 1 .. x0 + 62 = 64 positions starts on an even x: two
// MFMA column groups) and walks z; per input plane of `prob` the patch is 10 rows x 64 positions x 8 channels = 20 (row, column group) units, five per
// wave.  conv11's own input (16 channels at half resolution) is staged per half-resolution plane as a 6 x 34 box (two float16 slices, own power-of-two
// scale per plane; a ring of two planes: full-resolution plane z needs z / 2 and (z + 1) / 2).  Positions outside the volume are ZERO in the slot
// (`prob`'s padding), not convolution results.  LDS: slot 21 KiB + ring 26 KiB + lane images 18 KiB = 65 KiB: two workgroups per CU.
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"
#include "softmax_regress.h"
#include "split_f16.h"

namespace {

using namespace casmvs::buf;
typedef float cp_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 cp_f16x8 __attribute__((ext_vector_type(8)));

struct CpCfg {
  static constexpr int THREADS = 256;
  static constexpr int TXO = 62, TY = 8;           // output pixels per workgroup; a consumer thread = 2 consecutive x of one row
  static constexpr int IY = TY + 2;                // slot rows y0 - 1 .. y0 + TY
  static constexpr int NPP = 33;                   // position pairs per slot row (32 used: 64 positions; the 33rd pads the row stride as in prob_regress.hip)
  static constexpr int RS = 4 * NPP, SP = IY * RS, NPAIR = 4, SLOT = NPAIR * SP;   // floats: 132, 1320, 5280
  static constexpr int JY = 6, JX = 34;            // conv11's input box per half-resolution plane: rows y0 / 2 - 1 .. + 4, x xs / 2 .. + 33
  static constexpr int NVOX = 208;                 // units per (slice, channel half) plane: 204 used, a multiple of 16
  static constexpr int WUNITS = 9 * 2 * 64;
  static constexpr size_t SLOT_BYTES = (size_t)SLOT * 4 + 32;                    // 21 152
  static constexpr size_t ACT_BYTES = (size_t)2 * 4 * NVOX * 16;                 // ring of two planes: 26 624
  static constexpr size_t W_BYTES = (size_t)WUNITS * 16;                         // 18 432
  static constexpr size_t LDS_BYTES = SLOT_BYTES + ACT_BYTES + W_BYTES + 32;     // 66 240: two workgroups per CU
};

__device__ __forceinline__ cp_f32x4 cp_mfma(u32x4 a, u32x4 b, cp_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cp_f16x8, a), __builtin_bit_cast(cp_f16x8, b), c, 0, 0, 0);
}

// prob_regress.hip's zwalk_plane for this slot geometry: A[2 - kz] += sum_{pair, ky, kx} in * w[kz][ky][kx] for the kz in KZM
template <int KZM>
__device__ __forceinline__ void cp_zwalk_plane(const float *rows, const float *__restrict__ wpk, f32x2 (&A)[3][2]) {
  using Cfg = CpCfg;
  constexpr int NSTEP = Cfg::NPAIR * 3;
  f32x4v lo[2], hi[2];
  f32x2 Wt[2][3][3];
  auto fetch = [&](auto buf_, int i) {
    constexpr int BUF = decltype(buf_)::value;
    const float *row = rows + (i / 3) * Cfg::SP + (i % 3) * Cfg::RS;
    lo[BUF] = *reinterpret_cast<const f32x4v *>(row);
    hi[BUF] = *reinterpret_cast<const f32x4v *>(row + 4);
    const float *wq = wpk + (i / 3) * 64 + (i % 3) * 6;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      if (!((KZM >> kz) & 1)) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) Wt[BUF][kz][kx] = f32x2{wq[kz * 18 + 2 * kx], wq[kz * 18 + 2 * kx + 1]};
    }
  };
  auto fmas = [&](auto buf_) {
    constexpr int BUF = decltype(buf_)::value;
    const f32x2 P[4] = {f32x2{lo[BUF][0], lo[BUF][1]}, f32x2{lo[BUF][2], lo[BUF][3]}, f32x2{hi[BUF][0], hi[BUF][1]}, f32x2{hi[BUF][2], hi[BUF][3]}};
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        if (!((KZM >> kz) & 1)) continue;
        A[2 - kz][0] = __builtin_elementwise_fma(P[kx], Wt[BUF][kz][kx], A[2 - kz][0]);
        A[2 - kz][1] = __builtin_elementwise_fma(P[kx + 1], Wt[BUF][kz][kx], A[2 - kz][1]);
      }
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  fetch(B0{}, 0);
#pragma unroll
  for (int i = 0; i < NSTEP; i += 2) {
    fetch(B1{}, i + 1);
    __builtin_amdgcn_sched_barrier(0);
    fmas(B0{});
    __builtin_amdgcn_sched_barrier(0);
    if (i + 2 < NSTEP) fetch(B0{}, i + 2);
    __builtin_amdgcn_sched_barrier(0);
    fmas(B1{});
    __builtin_amdgcn_sched_barrier(0);
  }
}

// u9 (B, 16, Di/2, Hi/2, Wi/2): conv11's input; skip (B, 8, Di, Hi, Wi): conv0's output; cost (B, Di, Hi, Wi); Di, Hi, Wi even, Wi % 4 == 0.
// grid: x = tiles_x * tiles_y * chunks (XCD-major, chunk fastest, then x, then y), y = batch.
template <int DT, bool FUSE>
__global__ __launch_bounds__(CpCfg::THREADS, 2) void conv11_prob_kernel(
    const float *__restrict__ u9, const unsigned char *__restrict__ dpk, const float *__restrict__ skip, const float *__restrict__ wpk,
    const float *__restrict__ dvals, float *cost, float *__restrict__ depth, float *__restrict__ conf, int32_t *__restrict__ index, int Di, int Hi,
    int Wi, int tiles_x, int tiles_y, int zc, float slope) {
  using Cfg = CpCfg;
  constexpr int RS = Cfg::RS, NVOX = Cfg::NVOX, JX = Cfg::JX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *slot = reinterpret_cast<float *>(smem_raw);                                                              // [pair][row][RS]
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw + Cfg::SLOT_BYTES);                                             // [ring][slice][half][NVOX]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::SLOT_BYTES + Cfg::ACT_BYTES);                             // [9][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::SLOT_BYTES + Cfg::ACT_BYTES + Cfg::W_BYTES);      // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, kb = lane >> 4, half = kb & 1, dx = kb >> 1, u = kb;
  const int ntile = tiles_x * tiles_y;
  const int nchunk = gridDim.x / ntile;
  const int bid = xcd_major(blockIdx.x, gridDim.x);
  const int chunk = bid % nchunk, tl = bid / nchunk;
  const int x0 = (tl % tiles_x) * Cfg::TXO - 1, y0 = (tl / tiles_x) * Cfg::TY;   // first output pixel; the slot starts one position / row earlier
  const int xs = x0 - 1, ys = y0 - 1;                                            // xs even (62 k - 2), ys odd
  const int b = blockIdx.y;
  const int z_lo = chunk * zc, z_hi = min(z_lo + zc, Di);
  const int HiWi = Hi * Wi, ocs = Di * HiWi;
  const int Dh = Di / 2, Hh = Hi / 2, Wh = Wi / 2, hHW = Hh * Wh, ics = Dh * hHW;
  const rsrc_t usrc = make_rsrc(u9 + (size_t)b * 16 * ics, (size_t)16 * ics * 4);
  const rsrc_t ssrc = make_rsrc(skip + (size_t)b * 8 * ocs, (size_t)8 * ocs * 4);
  const rsrc_t dst = make_rsrc(cost + (size_t)b * ocs, (size_t)ocs * 4);
  const float *dtail = reinterpret_cast<const float *>(dpk + Cfg::W_BYTES);   // conv11: scale[8] (x 2^-kw) | shift[8]
  const float *ptail = wpk + 8 * 32;                                           // prob: scale[4] | shift[4] after the [pair][64] weight rows
  const float psc = ptail[0], psh = ptail[4];
  float dsc[2], dsh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    dsc[h] = dtail[2 * u + h];
    dsh[h] = dtail[8 + 2 * u + h];
  }
  for (int unit = tid; unit < Cfg::WUNITS; unit += Cfg::THREADS) wl[unit] = reinterpret_cast<const u32x4 *>(dpk)[unit];

  // ---- producer: staging of one half-resolution plane of u9 (box rows y0 / 2 - 1 .. + 4, x xs / 2 .. + 33): thread = one voxel, 16 channels ----
  const int s_iy = tid / JX, s_bx = tid - s_iy * JX;
  const bool s_item = tid < Cfg::JY * JX;
  const int s_gy = y0 / 2 - 1 + s_iy, s_gx = xs / 2 + s_bx;
  const int s_voff = (s_item && s_gy >= 0 && s_gy < Hh && s_gx >= 0 && s_gx < Wh) ? (s_gy * Wh + s_gx) * 4 : kOOB;
  float inv_ring[2] = {0.0f, 0.0f};   // 2^-kx of the plane in ring slot 0 / 1
  auto stage_plane = [&](int iz) {   // every thread of the workgroup; two barriers
    const bool exists = iz >= 0 && iz < Dh;
    float R[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) R[c] = buf_load(usrc, exists ? s_voff : kOOB, (c * ics + (exists ? iz : 0) * hHW) * 4);
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) m = fmaxf(m, fabsf(R[c]));
    const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
    if (lane == 0) wmax[wave] = wm;
    __syncthreads();
    float mult, inv;
    casmvs::tile_scale(wmax, mult, inv);
    const int rg = iz & 1;
    if (rg) inv_ring[1] = inv; else inv_ring[0] = inv;
    if (s_item) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = R[hf * 8 + c];
        u32x4 o[2];
        casmvs::split8_f16(x, mult, o);
#pragma unroll
        for (int s = 0; s < 2; ++s) act[((rg * 2 + s) * 2 + hf) * NVOX + tid] = o[s];
      }
    }
    __syncthreads();
  };

  // ---- producer: one (slot row r, column group gp) unit of full-resolution plane z into the slot ----
  // row r: oy = ys + r, odd for even r (ys odd): taps ky = 0 from box row r / 2 + 1 and ky = 2 from r / 2; even oy (odd r): ky = 1 from (r + 1) / 2.
  // plane z: even: kz = 1 from z / 2; odd: kz = 0 from (z + 1) / 2 and kz = 2 from (z - 1) / 2.  Column j of group gp: input xs / 2 + 16 gp + j (+ dx).
  auto produce_unit = [&](int z, int r, int gp, f32x2 sk0, f32x2 sk1) {
    const int bxl = 16 * gp + jcol + dx;
    cp_f32x4 part[2] = {cp_f32x4{0.f, 0.f, 0.f, 0.f}, cp_f32x4{0.f, 0.f, 0.f, 0.f}};   // by ring slot of the input plane (own scale each)
    constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
    const bool zodd = z & 1, yodd = !(r & 1);
#pragma unroll
    for (int tz = 0; tz < 2; ++tz) {
      if (!zodd && tz) continue;
      const int kz = zodd ? (tz ? 2 : 0) : 1;
      const int iz = zodd ? (tz ? (z - 1) / 2 : (z + 1) / 2) : z / 2;
      const int rg = iz & 1;
#pragma unroll
      for (int ty = 0; ty < 2; ++ty) {
        if (!yodd && ty) continue;
        const int ky = yodd ? (ty ? 2 : 0) : 1;
        const int iyl = yodd ? (ty ? r / 2 : r / 2 + 1) : (r + 1) / 2;
        u32x4 a[2], bq[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          a[s] = wl[((kz * 3 + ky) * 2 + s) * 64 + lane];
          bq[s] = act[((rg * 2 + s) * 2 + half) * NVOX + iyl * JX + bxl];
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (rg) part[1] = cp_mfma(a[PA[p]], bq[PB[p]], part[1]);
          else part[0] = cp_mfma(a[PA[p]], bq[PB[p]], part[0]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // result lane: rows 4 u + q = (channel 2 u + (q >> 1), x parity q & 1), column j -> positions x = xs + 32 gp + 2 j, + 1 of the channel pair u
    const int oy = ys + r, ox = xs + 32 * gp + 2 * jcol;
    const bool inside = oy >= 0 && oy < Hi && ox >= 0 && ox < Wi;   // ox, Wi even: the position pair is inside or outside
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int h = q >> 1;
      float t = fmaf(part[0][q], inv_ring[0], part[1][q] * inv_ring[1]);
      t = fmaf(t, dsc[h], dsh[h]);
      t = t > 0.0f ? t : t * slope;
      v[q] = t;
    }
    // slot item of (pair u, row r, position pair 16 gp + j): (c0[x], c1[x], c0[x + 1], c1[x + 1])
    const f32x4v item = inside ? f32x4v{v[0] + sk0[0], v[2] + sk1[0], v[1] + sk0[1], v[3] + sk1[1]} : f32x4v{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4v *>(slot + u * Cfg::SP + r * RS + 4 * (16 * gp + jcol)) = item;
    __builtin_amdgcn_sched_barrier(0);   // the next unit's matrix instructions stay behind this unit's floating-point epilogue
  };
  // this wave's five units: u_idx = 5 wave + i -> (r, gp) = (u_idx >> 1, u_idx & 1)
  auto skip_offset = [&](int i) {
    const int u_idx = 5 * wave + i, r = u_idx >> 1, gp = u_idx & 1;
    const int oy = ys + r, ox = xs + 32 * gp + 2 * jcol;
    return (oy >= 0 && oy < Hi && ox >= 0 && ox < Wi) ? ((2 * u) * ocs + oy * Wi + ox) * 4 : kOOB;
  };
  int soff[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) soff[i] = skip_offset(i);

  // ---- consumer (prob_zwalk_kernel): thread = output pixels x0 + 2 xi, + 1 of row y0 + yi ----
  const int xi = tid & 31, yi = tid >> 5;
  const int oyc = y0 + yi, oxc = x0 + 2 * xi;
  // outputs of this tile: x0 <= x < x0 + 62 (xi < 31) and inside the image; x0 = -1 for the first tile: its pixel -1 does not exist
  const bool px0 = xi < 31 && oyc < Hi && oxc >= 0 && oxc < Wi, px1 = xi < 31 && oyc < Hi && oxc + 1 >= 0 && oxc + 1 < Wi;
  f32x2 A[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i) A[i][0] = A[i][1] = f32x2{0.f, 0.f};
  const int nplanes = z_hi - z_lo + 2;
  const int z0 = z_lo - 1;
  int staged_hi = ((z0 >= 0 ? z0 : 0) >> 1) - 1;   // highest half-resolution plane in the ring
  for (int it = 0; it < nplanes; ++it) {
    const int zin = z0 + it;
    const bool exists = zin >= 0 && zin < Di;
    if (exists) {
      // the skip values of this plane's units: in flight under the staging and the matrix work
      f32x2 SK[5][2];
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) SK[i][h] = buf_load2(ssrc, soff[i], (h * ocs + zin * HiWi) * 4);
      while (staged_hi < ((zin + 1) >> 1)) stage_plane(++staged_hi);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int u_idx = 5 * wave + i;
        produce_unit(zin, u_idx >> 1, u_idx & 1, SK[i][0], SK[i][1]);
      }
    }
    __syncthreads();   // the slot holds plane zin
    if (exists) {
      const float *rows = slot + yi * RS + 4 * xi;
      if (it == 0) cp_zwalk_plane<1>(rows, wpk, A);
      else if (it == nplanes - 1) cp_zwalk_plane<4>(rows, wpk, A);
      else cp_zwalk_plane<7>(rows, wpk, A);
    }
    {  // output plane zin - 1 is complete; it lies in [z_lo, z_hi) from step 2 on
      float o0 = fmaf(A[0][0][0] + A[0][0][1], psc, psh), o1 = fmaf(A[0][1][0] + A[0][1][1], psc, psh);
      const int co = (zin - 1) * HiWi * 4;
      // x0 is odd: the two pixels are stored one by one
      buf_store(o0, dst, (it >= 2 && px0) ? (oyc * Wi + oxc) * 4 : kOOB, it >= 2 ? co : 0);
      buf_store(o1, dst, (it >= 2 && px1) ? (oyc * Wi + oxc + 1) * 4 : kOOB, it >= 2 ? co : 0);
    }
    A[0][0] = A[1][0];
    A[0][1] = A[1][1];
    A[1][0] = A[2][0];
    A[1][1] = A[2][1];
    A[2][0] = A[2][1] = f32x2{0.f, 0.f};
    __syncthreads();   // every wave is done with the slot
  }

  if constexpr (FUSE) {
    // every cost value of this thread's two pixels was stored by this thread: wait for the stores, then read them back
#ifndef HIPEMU_LDS_BYTES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma nounroll
    for (int j = 0; j < 2; ++j) {
      if (!(j ? px1 : px0)) continue;
      const size_t pix = (size_t)oyc * Wi + oxc + j;
      const float *cp = cost + (size_t)b * ocs + pix, *dp = dvals + (size_t)b * ocs + pix;
      float d, c;
      int ix;
      casmvs::softmax_regress_pixel<DT>(cp, dp, (size_t)HiWi, Di, d, c, ix);
      const size_t o = (size_t)b * HiWi + pix;
      depth[o] = d;
      conf[o] = c;
      if (index) index[o] = ix;
    }
  }
}

int cp_auto_zchunk(int tiles, int D) {   // as prob_regress.hip: the whole range when the pixel tiles alone fill the chip
  const int want = 600;
  if (tiles >= want) return D;
  const int cand[] = {32, 24, 16, 12, 8, 4};
  int best = D;
  for (int zc : cand) {
    if (zc >= D || D % zc) continue;
    best = zc;
    if ((long)tiles * (D / zc) >= want) break;
  }
  return best;
}

}  // namespace

extern "C" int casmvs_conv11_prob_regress_supported(int D, int h, int w) { return D > 0 && h > 0 && w >= 4 && D % 2 == 0 && h % 2 == 0 && w % 4 == 0; }

extern "C" int casmvs_conv11_prob_regress_f32(const void *deconv11_image, const float *prob_packed, const float *u9, const float *skip,
                                              const float *depth_values, float *cost, float *depth, float *confidence, int32_t *index, int B, int D,
                                              int h, int w, float slope, int zchunk, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(deconv11_image && prob_packed && u9 && skip && cost, "conv11_prob_regress: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && casmvs_conv11_prob_regress_supported(D, h, w), "conv11_prob_regress: B=%d D=%d h=%d w=%d (D, h even, w %% 4 == 0)", B, D, h, w);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(u9) | reinterpret_cast<size_t>(skip) | reinterpret_cast<size_t>(cost) | reinterpret_cast<size_t>(deconv11_image)) & 15) == 0,
                 "conv11_prob_regress: 16-byte aligned tensors");
  CASMVS_REQUIRE((size_t)8 * D * h * w < ((size_t)1 << 29), "conv11_prob_regress: one sample's skip tensor must hold < 2^29 floats");
  const bool regress = depth != nullptr;
  if (regress) CASMVS_REQUIRE(depth_values && confidence, "conv11_prob_regress: depth_values / confidence are required with depth");
  CASMVS_REQUIRE(zchunk >= 0, "conv11_prob_regress: zchunk=%d", zchunk);
  using Cfg = CpCfg;
  const int tiles_x = casmvs::ceil_div(w + 1, Cfg::TXO), tiles_y = casmvs::ceil_div(h, Cfg::TY);   // tile k covers x 62 k - 1 .. 62 k + 60
  const int zc = zchunk > 0 ? (zchunk < D ? zchunk : D) : cp_auto_zchunk(tiles_x * tiles_y * B, D);
  const int nchunk = casmvs::ceil_div(D, zc);
  const bool fuse = regress && nchunk == 1;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(tiles_x * tiles_y * nchunk), (unsigned)B), blk(Cfg::THREADS);
  const unsigned char *dpk = reinterpret_cast<const unsigned char *>(deconv11_image);
#define CASMVS_CP(DT, FUSE)                                                                                                                   \
  {                                                                                                                                           \
    auto kernel = conv11_prob_kernel<DT, FUSE>;                                                                                               \
    if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "conv11_prob_kernel")) return rc;          \
    hipLaunchKernelGGL(kernel, grid, blk, Cfg::LDS_BYTES, st, u9, dpk, skip, prob_packed, depth_values, cost, depth, confidence, index, D, h, \
                       w, tiles_x, tiles_y, zc, slope);                                                                                       \
  }
  if (!fuse) {
    CASMVS_CP(0, false)
  } else {
    switch (D) {
      case 8: CASMVS_CP(8, true) break;
      case 16: CASMVS_CP(16, true) break;
      case 32: CASMVS_CP(32, true) break;
      case 48: CASMVS_CP(48, true) break;
      default: CASMVS_CP(0, true) break;
    }
  }
#undef CASMVS_CP
  if (int rc = casmvs::check_launch("conv11_prob_kernel")) return rc;
  if (regress && !fuse) return casmvs_softmax_regress_f32(cost, depth_values, depth, confidence, index, B, D, h, w, stream);
  return CASMVS_OK;
}
