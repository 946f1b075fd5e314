// CostRegNet's tail as ONE kernel that walks the depth axis: conv11 = ConvTranspose3d(16 -> 8, k3 s2 p1 op1) + ABN + leaky-relu, `conv0 + ...`
// (models/mvsnet.py:84-86,101), then `prob` = Conv3d(8 -> 1, k3 p1, bias) (:89,104) and the softmax / depth regression / confidence over depth
// (:174-193, models/modules.py:95-104).
//
// Why.  As two kernels the 8-channel full-resolution tensor between conv11 and `prob` is written and read back: 1.34 GB per launch pair at cascade levels
// 0 / 1 (batch 8), 3.2 GB of the step's HBM traffic, and conv11 alone already runs at the practical streaming rate.  The tiled fusion of
// round 3 recomputed a one-voxel halo of conv11's output around every 4 x 8 x 32 tile (2x the matrix work and the skip reads) and lost.  Walking z removes the
// z halo: a workgroup owns 16 x 60 output pixels, produces conv11's output plane z for the 18 x 62 halo tile (1.16x) straight into the LDS slot `prob`'s
// multiply phase reads (prob_zwalk.h: the phase is shared with prob_regress.hip), and never stores it.
//
// Per output plane z (one workgroup barrier per plane):
//   (a) transposed convolution, the arithmetic of deconv11_splitf16.hip (same lane images): rows = (output channel, x parity), columns = 16 input x,
//       K = (dx, 16 input channels); 36 (row, column tile) units over 8 waves; an even plane takes tap kz = 1 of input plane z / 2, an odd plane taps
//       kz = 0 / 2 of planes (z + 1) / 2 and (z - 1) / 2 (two accumulator chains: the two staged planes carry different power-of-two scales).  Epilogue:
//       2^-kx, ABN, leaky-relu, + the skip tensor (prefetched a plane ahead), zeros outside the volume (`prob`'s padding) -> slot[z & 1], in the
//       [pair][row][position][channel of the pair] layout: the accumulator's (co, x parity) rows ARE a lane's two positions x two channels.
//   (b) every second plane the next input plane (10 x 36 voxels x 16 channels) is split into two float16 slices into the box ring (two planes); its
//       loads were issued two planes earlier, its maximum published one plane earlier (no extra barrier).
//   (c) after the barrier: `prob`'s 108 packed FMAs per voxel on slot[z & 1] into three rotating accumulators; output plane z - 1 is complete: cost store.
// After the walk every thread runs the softmax regression on the cost values of its two pixels (its own stores), as prob_zwalk_kernel does.
//
// Measured (MI355X, batch 8; profiles/r04_conv11_prob_zfused_*.txt): 1.26x / 1.27x / 1.12x the two kernels at levels 1 / 0 / 2 (640 -> 509 us), 0.28 ms of the
// step.  Shader-clock trace of a plane step (-DCASMVS_ZF_TRACE, tools/native/zf_trace.cpp): ~9 350 cycles = matrix phase 750-1 700, epilogue 1 400-2 300,
// next loads + box ring 800-3 300, barrier 600-2 700, `prob`'s multiply phase 2 000-3 700, cost store 430-540: a chain of vector-issue-bound phases
// (~1 000 vector instructions per SIMD and step) with the 8 waves in step.  A variant with producer waves (conv11 of plane s) beside consumer waves (`prob` of
// plane s - 1, two row groups per thread; commit 5dbb662) ran at the same speed (480 / 517 us: the producers' own chain - 9 units' epilogues, all loads - is as
// long as the whole step) and was removed; what would shorten the step is fewer vector instructions: `prob` on the matrix cores, packed epilogue arithmetic.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"
#include "prob_zwalk.h"
#include "softmax_regress.h"
#include "split_f16.h"

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct FzCfg {
  static constexpr int THREADS = 512, WAVES = 8;
  static constexpr int TY = 16, TX = 60;                               // output pixels per workgroup
  // `prob`'s plane slot (ProbZCfg's layout): rows y0 - 1 .. y0 + TY, positions x0 - 1 .. x0 + TX
  static constexpr int IY = TY + 2, NPOS = TX + 2;
  static constexpr int RS = 132;                                       // floats per row of a channel pair (124 used; prob_regress.hip's conflict-free stride)
  static constexpr int SP = IY * RS, SLOT = 4 * SP;                    // floats per channel pair / per plane slot (38 016 B)
  // conv11's input box of one input plane: rows iy0 - 1 .. iy0 + 8, columns ix0 - 2 .. ix0 + 33 (whole pairs from an even column), 16 channels
  static constexpr int JY = TY / 2 + 2, JX = TX / 2 + 6;               // 10 x 36
  static constexpr int NVB = ((JY * JX + 15) / 16) * 16;               // 16-byte units per (slice, channel half) plane: 368 (planes start on the same bank)
  static constexpr int BOX = 4 * NVB;                                  // units per staged input plane (23 552 B)
  static constexpr int ITEMS = 2 * JY * (JX / 2);                      // (channel half, row, pair of columns): 360 of the 512 threads
  static constexpr int UNITS = IY * 2;                                 // (slot row, column tile of 16 input x) matrix units: 36
  static constexpr int NU = (UNITS + WAVES - 1) / WAVES;               // per wave: 5 (waves 0-3) / 4
  static constexpr int WUNITS = 9 * 2 * 64;                            // lane images [kz * 3 + ky][slice][lane] (deconv11_splitf16.hip's image)
  static constexpr size_t SLOT_BYTES = (size_t)SLOT * 4, BOX_BYTES = (size_t)BOX * 16, W_BYTES = (size_t)WUNITS * 16;
  static constexpr size_t FRONT = 16;                                  // bytes in front of slot 0: the epilogue's unmasked store of position -1 of its first row lands here
#ifndef CASMVS_ZF_WLDS
#define CASMVS_ZF_WLDS 0   // A/B builds: 1 = `prob`'s weights from an LDS image through wave-uniform 16-byte reads (round 6: scalar spills 112 -> 39, counted waits - and 3 % SLOWER, 219 / 510 / 520 us against 213 / 490 / 500: the LDS pipe is the scarcer resource)
#endif
  static constexpr size_t PW_BYTES = CASMVS_ZF_WLDS ? 12 * 20 * 4 : 0;   // `prob`'s weights, [step][20]
  static constexpr size_t LDS_BYTES = FRONT + 2 * SLOT_BYTES + 2 * BOX_BYTES + W_BYTES + 64 + PW_BYTES;   // 142 608: one workgroup of 8 waves per CU
  static_assert(TX % 4 == 0 && TY % 2 == 0 && ITEMS <= THREADS && TX / 2 + 2 <= 32, "shape");
};

__device__ __forceinline__ f32x4 fz_mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// mult = 2^kx puts the largest of the eight waves' maxima into [2^14, 2^15) (split_f16.h: tile_scale for four)
__device__ __forceinline__ void fz_scale8(const unsigned *wm, float &mult, float &inv) {
  const casmvs::split_u32x4 a = *reinterpret_cast<const casmvs::split_u32x4 *>(wm), b = *reinterpret_cast<const casmvs::split_u32x4 *>(wm + 4);
  int e = (int)(max(max(max(a[0], a[1]), max(a[2], a[3])), max(max(b[0], b[1]), max(b[2], b[3]))) >> 23);
  const bool infinite = e >= 255;
  e = e < 15 ? 15 : (e > 254 ? 254 : e);
  mult = __builtin_bit_cast(float, infinite ? 0x7fc00000u : (unsigned)(268 - e) << 23);
  inv = __builtin_bit_cast(float, (unsigned)(e - 14) << 23);
}

#ifdef CASMVS_ZF_TRACE
// Profiling builds (tools/native/zf_trace.cpp): waves 0 and 7 of workgroup (0, 0) stamp the shader clock at the phase boundaries of every plane step.
__device__ unsigned long long g_zf_trace[2][512];
#define ZF_STAMP()                                                                                                            \
  do {                                                                                                                        \
    if (blockIdx.x == 0 && blockIdx.y == 0 && (wave == 0 || wave == 7) && zf_tn < 512) g_zf_trace[wave == 7][zf_tn++] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define ZF_STAMP() do {} while (0)
#endif

// in (B, 16, Di, Hi, Wi) float32 (conv9's output; Wi even, 8-byte aligned); skip (B, 8, 2 Di, 2 Hi, 2 Wi) (conv0's output); wdc: deconv11_splitf16.hip's
// packed image (lane images, scale[8] x 2^-kw, shift[8]); wpk: `prob`'s P1 image (conv3d_mfma.hip) with its scale / shift tail; dvals (B, Do, Ho, Wo);
// cost (B, Do, Ho, Wo) is written; depth / conf (B, Ho, Wo) [, index].  grid: x = tiles_x * tiles_y (XCD-major), y = batch.  DT: compile-time Do (0: generic).
template <int DT>
__global__ __launch_bounds__(FzCfg::THREADS, 1) void conv11_prob_zfused_kernel(
    const float *__restrict__ in, const unsigned char *__restrict__ wdc, const float *__restrict__ skip, const float *__restrict__ wpk,
    const float *__restrict__ dvals, float *cost, float *__restrict__ depth, float *__restrict__ conf, int32_t *__restrict__ index, int Di, int Hi, int Wi,
    int tiles_x, int tiles_y, float slope, float pslope) {
  using Cfg = FzCfg;
  constexpr int RS = Cfg::RS, SP = Cfg::SP, SLOT = Cfg::SLOT, JX = Cfg::JX, JY = Cfg::JY, NVB = Cfg::NVB, BOX = Cfg::BOX, NU = Cfg::NU;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *slots = reinterpret_cast<float *>(smem_raw + Cfg::FRONT);                                                      // [2][SLOT]
  u32x4 *box = reinterpret_cast<u32x4 *>(smem_raw + Cfg::FRONT + 2 * Cfg::SLOT_BYTES);                                   // [2 planes][slice][half][NVB]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::FRONT + 2 * Cfg::SLOT_BYTES + 2 * Cfg::BOX_BYTES);               // [9][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::FRONT + 2 * Cfg::SLOT_BYTES + 2 * Cfg::BOX_BYTES + Cfg::W_BYTES);   // [2 sets][8]
  // `wave` as a SCALAR: the compiler cannot see that threadIdx.x >> 6 is uniform, and every branch on a value derived from it (row parity of the wave's units,
  // whether its last unit exists, whether it stages) became exec-mask control flow with the masks carried - and spilled - across the whole walk
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jcol = lane & 15, kb = lane >> 4, half = kb & 1, dx = kb >> 1, u = kb;
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const int iHW = Hi * Wi, ics = Di * iHW, oHW = Ho * Wo, ocs = Do * oHW;
  const int bid = xcd_major(blockIdx.x, gridDim.x);
  const int tx0 = (bid % tiles_x) * Cfg::TX, ty0 = (bid / tiles_x) * Cfg::TY;
  const int ix0 = tx0 / 2, iy0 = ty0 / 2;
  const int b = blockIdx.y;
  const rsrc_t isrc = make_rsrc(in + (size_t)b * 16 * ics, (size_t)16 * ics * 4);
  const rsrc_t ssrc = make_rsrc(skip + (size_t)b * 8 * ocs, (size_t)8 * ocs * 4);
  const rsrc_t cdst = make_rsrc(cost + (size_t)b * ocs, (size_t)ocs * 4);
  const float *dtail = reinterpret_cast<const float *>(wdc + Cfg::W_BYTES);
  float sc[2], sh[2];   // conv11's folded ABN of the lane's channel pair (2 u, 2 u + 1)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sc[h] = dtail[2 * u + h];
    sh[h] = dtail[8 + 2 * u + h];
  }
  const float *ptail = wpk + 8 * 32;  // `prob`: scale[4] | shift[4] after the [pair][64] weight rows
  const float psc = ptail[0], psh = ptail[4];
  for (int unit = tid; unit < Cfg::WUNITS; unit += Cfg::THREADS) wl[unit] = reinterpret_cast<const u32x4 *>(wdc)[unit];
#if CASMVS_ZF_WLDS
  float *pwl = reinterpret_cast<float *>(smem_raw + Cfg::FRONT + 2 * Cfg::SLOT_BYTES + 2 * Cfg::BOX_BYTES + Cfg::W_BYTES + 64);   // [12 steps][20]
  if (tid < 12 * 20) {   // step i = (pair i / 3, ky i % 3); entry e = 2 (3 kz + kx) + c  <-  the P1 image's wpk[pair * 64 + kz * 18 + ky * 6 + 2 kx + c]
    const int i = tid / 20, e = tid - i * 20, kz = e / 6, r = e - kz * 6;
    pwl[tid] = e < 18 ? wpk[(i / 3) * 64 + kz * 18 + (i % 3) * 6 + r] : 0.0f;
  }
#endif

  // ---- (a) this wave's matrix units: q -> slot row (wave >> 1) + 4 q (one row parity per wave), column tile wave & 1 ----
  const int tile = wave & 1, row0 = wave >> 1;
  const bool row_odd = row0 & 1;                  // slot row r <-> output y = ty0 - 1 + r: odd rows are EVEN output rows (tap ky = 1)
  const int J = 16 * tile + jcol;                 // the lane's input column ix0 - 1 + J -> output x = tx0 - 2 + 2 J (+ 1)
  // B unit (slice 0) of box row rb: plane `half`, column index (ix0 - 1 + J + dx) - (ix0 - 2)
  const int bcol = half * NVB + J + dx + 1;
  // skip / validity of the lane's two positions per unit (tile constants)
  const int ox = tx0 - 2 + 2 * J;                 // even; the pair (ox, ox + 1) is inside or outside the row (Wo even)
  const bool x_in = ox >= 0 && ox < Wo;
  int sk_off[NU];                                 // byte offset of (channel 2 u, plane 0, row, ox) or kOOB
  unsigned vmask[NU];                             // ~0 inside the volume, 0 outside: ANDed onto the epilogue's values (a select on a loop-invariant per-lane
                                                  // predicate is hoisted by the compiler as a 64-bit lane mask in two scalar registers - five of them, and
                                                  // the exec juggling of every masked store, were most of the 112 scalar spills of round 5's build)
  const bool last_unit = row0 + 4 * (NU - 1) < Cfg::IY;   // unit NU - 1 exists for waves 0-3 (slot rows 16, 17); wave-uniform
#pragma unroll
  for (int q = 0; q < NU; ++q) {
    const int row = row0 + 4 * q, oy = ty0 - 1 + row;
    const bool in_vol = row < Cfg::IY && oy >= 0 && oy < Ho && x_in;
    sk_off[q] = in_vol ? ((2 * u) * ocs + oy * Wo + ox) * 4 : kOOB;
    vmask[q] = in_vol ? 0xffffffffu : 0u;
#ifndef HIPEMU_LDS_BYTES
    asm volatile("" : "+v"(vmask[q]));   // opaque: the optimiser otherwise folds `t & (in_vol ? ~0 : 0)` back into a select on the hoisted lane mask
#endif
  }
  f32x2 SK[NU][2];
  auto load_skip = [&](int z) {                   // the skip values of plane min(z, Do - 1): unconditional issue (plane Do is never computed: its values are
    const int soff = min(z, Do - 1) * oHW * 4;    // never used - re-reading the last plane keeps ONE descriptor and the same loads on every path)
#pragma unroll
    for (int q = 0; q < NU; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h) SK[q][h] = buf_load2(ssrc, sk_off[q], soff + h * ocs * 4);
  };

  // ---- (b) box staging item of this thread: (channel half, box row, pair of columns) ----
  // waves 2-7 stage (384 threads for 360 items: the last 24 repeat item 359 - the same loads, the same values to the same addresses): a wave-uniform
  // predicate is a scalar branch, a per-lane one a lane mask carried in scalar registers across the whole walk
  const bool stager = wave >= 2;
  const int s_e = min(tid - 128, Cfg::ITEMS - 1);
  const int s_h = s_e / (JY * (JX / 2)), s_rem = s_e - s_h * (JY * (JX / 2)), s_r = s_rem / (JX / 2), s_g = s_rem - s_r * (JX / 2);
  const int s_unit = s_h * NVB + s_r * JX + 2 * s_g;
  int s_voff;
  {
    const int gy = iy0 - 1 + s_r, gx = ix0 - 2 + 2 * s_g;   // gx even, Wi even: the pair is inside or outside
    const bool ok = stager && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
    s_voff = ok ? ((8 * s_h) * ics + gy * Wi + gx) * 4 : kOOB;
  }
  f32x2 R[8];
  auto load_box = [&](int iz) {                   // input plane iz; a plane behind the volume (taps kz = 0 of the last odd output plane) reads zeros: every
    const int voff = iz < Di ? s_voff : kOOB;     // lane's offset out of range - one descriptor, the same loads on every path
    const int soff = min(iz, Di - 1) * iHW * 4;
#pragma unroll
    for (int c = 0; c < 8; ++c) R[c] = buf_load2(isrc, voff, soff + c * ics * 4);
  };
  auto box_max = [&](int set) {
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) m = casmvs::absmax3(m, R[c][0], R[c][1]);
    const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
    if (lane == 0) wmax[set * 8 + wave] = wm;
  };
  auto box_write = [&](int plane_slot, float mult) {
#ifdef HIPEMU_LDS_BYTES
    // (CPU emulation only) the 24 threads that repeat item 359 store the same values to the same LDS words as thread 487: nothing on the GPU, a reported
    // write-write race under ThreadSanitizer, whose report list the suite requires to be empty
    if (tid - 128 >= Cfg::ITEMS) return;
#endif
    if (stager) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = R[c][p];
        casmvs::split_u32x4 o[2];
        casmvs::split8_f16(x, mult, o);
        u32x4 *dst = box + plane_slot * BOX + s_unit + p;
        dst[0] = o[0];
        dst[2 * NVB] = o[1];
      }
    }
  };

  // ---- (c) `prob`: thread = pixels (tx0 + 2 xi, + 1) of row ty0 + yi ----
  const int xi = tid & 31, yi = tid >> 5;
  const int oy = ty0 + yi, oxp = tx0 + 2 * xi;
  const bool pix_ok = 2 * xi < Cfg::TX && oy < Ho && oxp < Wo;   // Wo even: the pixel pair is inside or outside
  const int out_voff = pix_ok ? (oy * Wo + oxp) * 4 : kOOB;
  f32x2 A[3][2];  // [0]: output plane z - 1, [1]: z, [2]: z + 1; [pixel]; (even, odd channels' partial sums)
#pragma unroll
  for (int i = 0; i < 3; ++i) A[i][0] = A[i][1] = f32x2{0.f, 0.f};

  float inv_b[2];   // 2^-kx of the plane in box ring slot 0 / 1
#ifdef CASMVS_ZF_TRACE
  int zf_tn = 0;
#endif
  // one plane step; ODD: z = 2 k + 1 (taps kz = 0 of input plane k + 1, kz = 2 of plane k), else z = 2 k (tap kz = 1 of plane k)
  auto step = [&](int k, auto odd_) {
    constexpr bool ODD = decltype(odd_)::value;
    const int z = 2 * k + (ODD ? 1 : 0);
    float *slot = slots + (ODD ? SLOT : 0);
    ZF_STAMP();   // t0: top of the step
    // -- (a) matrix phase: every unit's chains, nothing but matrix instructions and their operand reads --
    const int pa = ODD ? ((k + 1) & 1) : (k & 1), pb = k & 1;   // box ring slots of chain 0 / chain 1 (chain 1: odd planes only)
    const float inv0 = pa ? inv_b[1] : inv_b[0], inv1 = pb ? inv_b[1] : inv_b[0];
    f32x4 acc0[NU], acc1[NU];
#pragma unroll
    for (int q = 0; q < NU; ++q) acc0[q] = acc1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
    __builtin_amdgcn_sched_barrier(0);
    // the lane images of the step: kz0 = (ODD ? 0 : 1) for chain 0, kz = 2 for chain 1; ky = 1 (even output rows) or ky = 0 and 2 (odd output rows)
    auto chains = [&](auto rowodd_) {
      constexpr bool RODD = decltype(rowodd_)::value;
      constexpr int NKY = RODD ? 1 : 2;
      constexpr int KY[2] = {RODD ? 1 : 0, 2};
      // granule g = (tap i = g / NU, unit q = g % NU): tap-major, so that one tap's lane images are live at a time; a granule's B operands are read while
      // the previous granule multiplies (two operand buffers)
      constexpr int NG = NU * NKY;
      u32x4 a0[2], a1[2], bvb[2][2][2];   // lane images [slice] of chain 0 / 1; B operands [buffer][chain][slice]
      // odd slot row r = row0 + 4 q (even output y): box row (r + 1) / 2 = (row0 + 1) / 2 + 2 q; even slot row: tap ky = 0 from box row r / 2 + 1, ky = 2 from
      // r / 2 = row0 / 2 + 2 q: one per-lane base, the rest immediates
      const u32x4 *b0 = box + pa * BOX + bcol + (RODD ? (row0 + 1) / 2 : row0 / 2) * JX, *b1 = box + pb * BOX + bcol + (RODD ? (row0 + 1) / 2 : row0 / 2) * JX;
      auto fetch = [&](int buf, int g) {
        const int i = g / NU, q = g % NU;
        const int off = (2 * q + ((!RODD && i == 0) ? 1 : 0)) * JX;   // (unit 4 of waves 4-7 does not exist: box row <= 10 stays inside the LDS allocation)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bvb[buf][0][s] = b0[s * 2 * NVB + off];
          if (ODD) bvb[buf][1][s] = b1[s * 2 * NVB + off];
        }
      };
      fetch(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int i = g / NU, q = g % NU;
        if (q == 0) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            a0[s] = wl[(((ODD ? 0 : 1) * 3 + KY[i]) * 2 + s) * 64 + lane];
            if (ODD) a1[s] = wl[((2 * 3 + KY[i]) * 2 + s) * 64 + lane];
          }
        }
        if (g + 1 < NG) fetch((g + 1) & 1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (q < NU - 1 || last_unit) {
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            acc0[q] = fz_mfma(a0[PA[p]], bvb[g & 1][0][PB[p]], acc0[q]);
            if (ODD) acc1[q] = fz_mfma(a1[PA[p]], bvb[g & 1][1][PB[p]], acc1[q]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (row_odd) chains(std::true_type{});
    else chains(std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    ZF_STAMP();   // t1: matrix instructions issued
    float *slot_lane = slot + u * SP + row0 * RS + 4 * J;
    // -- epilogue: lane holds rows 4 u + r = (co = 2 u + (r >> 1), x parity r & 1) of column J: positions 2 J - 1 (parity 0) and 2 J (parity 1) --
#pragma unroll
    for (int q = 0; q < NU; ++q) {
      if (q == NU - 1 && !last_unit) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = acc0[q][r] * inv0;
        if (ODD) t = t + acc1[q][r] * inv1;
        t = fmaf(t, sc[r >> 1], sh[r >> 1]);
        t = fmaxf(t, t * slope);    // leaky-relu for 0 <= slope <= 1 (checked on the host): t > 0 ? t : t * slope, the same product, no compare + select
        t = t + SK[q][r >> 1][r & 1];
        v[r] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, t) & vmask[q]);   // outside the volume: `prob`'s zero padding
      }
      float *prow = slot_lane + q * 4 * RS;                                                    // slot + u SP + row RS + 4 J
      // both stores unmasked: position -1 of column J = 0 is the previous row's unused tail (floats 130, 131 of its 132; in front of slot 0: Cfg::FRONT),
      // position 62 of J = 31 this row's (floats 124, 125) - nothing reads either
      *reinterpret_cast<f32x2 *>(prow - 2) = f32x2{v[0], v[2]};                               // position 2 J - 1: (channel 2 u, 2 u + 1)
      *reinterpret_cast<f32x2 *>(prow) = f32x2{v[1], v[3]};                                   // position 2 J
    }
    ZF_STAMP();   // t2: epilogue (skip values landed, slot written)
    // the next plane's skip values: in flight across the `prob` phase
    load_skip(z + 1);
    // -- (b) the box ring: an even step writes input plane k + 1 (first read by step 2 k + 1) and puts plane k + 2 in flight; an odd step publishes its maximum --
    if (!ODD) {
      if (k >= 1) {   // (planes 0 and 1: the prologue)
        float mult, inv;
        fz_scale8(wmax + ((k + 1) & 1) * 8, mult, inv);
        box_write((k + 1) & 1, mult);
        if ((k + 1) & 1) inv_b[1] = inv;
        else inv_b[0] = inv;
      }
      load_box(k + 2);
    } else {
      box_max(k & 1);   // plane k + 2 -> set (k + 2) & 1
    }
    ZF_STAMP();   // t3: loads issued, box ring work done
    __syncthreads();   // slot[z & 1] and the box are published; the other slot is free
    ZF_STAMP();   // t4: past the barrier
    // -- (c) `prob`: plane z feeds output planes z - 1, z, z + 1 --
#if CASMVS_ZF_WLDS
    casmvs::pz::zwalk_plane<7, SP, RS, true>(slot + yi * RS + 4 * xi, pwl, A);
#else
    casmvs::pz::zwalk_plane<7, SP, RS>(slot + yi * RS + 4 * xi, wpk, A);
#endif
#ifndef HIPEMU_LDS_BYTES
    // pin the accumulators here: two thirds of the plane's FMAs feed A[1] / A[2], whose next use is behind the NEXT plane's matrix phase - the optimiser sank
    // them (and their 100 operand registers) across it, which spilled
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(A[i][j]));
#endif
    ZF_STAMP();   // t5: `prob`'s multiply phase
    {  // output plane z - 1 is complete
      float o0 = fmaf(A[0][0][0] + A[0][0][1], psc, psh), o1 = fmaf(A[0][1][0] + A[0][1][1], psc, psh);
      o0 = o0 > 0.0f ? o0 : o0 * pslope;
      o1 = o1 > 0.0f ? o1 : o1 * pslope;
      // (step z = 0 has no finished plane: it stores its value - the bias through the activation - into plane Do - 1, which the store behind the walk
      // overwrites: vmcnt retires in order, and every step in between waits for loads issued after this store)
      buf_store2(f32x2{o0, o1}, cdst, out_voff, (z >= 1 ? z - 1 : Do - 1) * oHW * 4);
    }
    A[0][0] = A[1][0];
    A[0][1] = A[1][1];
    A[1][0] = A[2][0];
    A[1][1] = A[2][1];
    A[2][0] = A[2][1] = f32x2{0.f, 0.f};
  };

  // ---- prologue: input planes 0 and 1 into the box ring, the skip values of plane 0 ----
  load_box(0);
  load_skip(0);
  box_max(0);
  __syncthreads();   // (also publishes the lane images)
  {
    float mult;
    fz_scale8(wmax, mult, inv_b[0]);
    box_write(0, mult);
  }
  load_box(1);
  box_max(1);
  __syncthreads();
  {
    float mult;
    fz_scale8(wmax + 8, mult, inv_b[1]);
    box_write(1, mult);
  }
  __syncthreads();
  for (int k = 0; k < Di; ++k) {
    step(k, std::false_type{});
    step(k, std::true_type{});
  }
  {  // output plane Do - 1 (plane Do does not exist)
    float o0 = fmaf(A[0][0][0] + A[0][0][1], psc, psh), o1 = fmaf(A[0][1][0] + A[0][1][1], psc, psh);
    o0 = o0 > 0.0f ? o0 : o0 * pslope;
    o1 = o1 > 0.0f ? o1 : o1 * pslope;
    buf_store2(f32x2{o0, o1}, cdst, out_voff, (Do - 1) * oHW * 4);
  }
  // every cost value of this thread's two pixels was stored by this thread: wait for the stores, then read them back
#ifndef HIPEMU_LDS_BYTES
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  if (pix_ok) {
    const size_t pix = (size_t)oy * Wo + oxp;
    const float *cp = cost + (size_t)b * ocs + pix, *dp = dvals + (size_t)b * ocs + pix;
    const size_t o = (size_t)b * oHW + pix;
#pragma nounroll
    for (int j = 0; j < 2; ++j) {
      float d, c;
      int ix;
      casmvs::softmax_regress_pixel<DT>(cp + j, dp + j, (size_t)oHW, Do, d, c, ix);
      depth[o + j] = d;
      conf[o + j] = c;
      if (index) index[o + j] = ix;
    }
  }
}

}  // namespace

// Do = 2 Di planes of Ho x Wo = 2 Hi x 2 Wi pixels; the whole depth range is walked by one workgroup per 16 x 60 pixel tile.
extern "C" int casmvs_conv11_prob_zfused_supported(int Di, int Hi, int Wi) { return Di >= 1 && Hi >= 1 && Wi >= 2 && Wi % 2 == 0; }

extern "C" int casmvs_conv11_prob_zfused_f32(const void *deconv11_packed, const float *prob_packed, const float *in, const float *skip,
                                             const float *depth_values, float *cost, float *depth, float *confidence, int32_t *index, int B, int Di, int Hi,
                                             int Wi, float slope, float prob_slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(deconv11_packed && prob_packed && in && skip && depth_values && cost && depth && confidence, "conv11_prob_zfused: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && casmvs_conv11_prob_zfused_supported(Di, Hi, Wi), "conv11_prob_zfused: B=%d Di=%d Hi=%d Wi=%d (Wi even)", B, Di, Hi, Wi);
  CASMVS_REQUIRE(slope >= 0.0f && slope <= 1.0f && prob_slope >= 0.0f && prob_slope <= 1.0f, "conv11_prob_zfused: leaky-relu slopes %g / %g outside [0, 1] (the kernel forms max(t, slope t))", slope, prob_slope);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(skip) | reinterpret_cast<size_t>(cost) | reinterpret_cast<size_t>(depth) |
                   reinterpret_cast<size_t>(confidence)) & 7) == 0 && (reinterpret_cast<size_t>(deconv11_packed) & 15) == 0,
                 "conv11_prob_zfused: 8-byte aligned tensors, 16-byte aligned image");
  CASMVS_REQUIRE((size_t)64 * Di * Hi * Wi < ((size_t)1 << 29), "conv11_prob_zfused: one sample's skip tensor must hold < 2^29 floats");
  using Cfg = FzCfg;
  const int Do = 2 * Di, tiles_x = casmvs::ceil_div(2 * Wi, Cfg::TX), tiles_y = casmvs::ceil_div(2 * Hi, Cfg::TY);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)B), blk(Cfg::THREADS);
#define CASMVS_FZ(DT)                                                                                                                        \
  do {                                                                                                                                       \
    auto kernel = conv11_prob_zfused_kernel<DT>;                                                                                             \
    if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "conv11_prob_zfused_kernel")) return rc; \
    hipLaunchKernelGGL(kernel, grid, blk, Cfg::LDS_BYTES, st, in, reinterpret_cast<const unsigned char *>(deconv11_packed), skip, prob_packed, \
                       depth_values, cost, depth, confidence, index, Di, Hi, Wi, tiles_x, tiles_y, slope, prob_slope);                        \
  } while (0)
  switch (Do) {
    case 8: CASMVS_FZ(8); break;
    case 32: CASMVS_FZ(32); break;
    case 48: CASMVS_FZ(48); break;
    default: CASMVS_FZ(0); break;
  }
#undef CASMVS_FZ
  return casmvs::check_launch("conv11_prob_zfused_kernel");
}

#ifdef CASMVS_ZF_TRACE
extern "C" int casmvs_zf_trace_read(unsigned long long *host) {
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_zf_trace), sizeof(unsigned long long) * 2 * 512) == hipSuccess ? 0 : -3;
}
#endif
