// CostRegNet.conv0 (Conv3d Cin -> 8, k3 s1 p1 + folded ABN + leaky-relu) in split-f16 arithmetic, INPUT-stationary along z.
//
// Status: conv0_zw_kernel below (the warp-specialised z-march) is what the engine runs at all three cascade levels - the step's dominant kernel
// (`roofline` of the bench line); conv0_zm_kernel (two phases per unit, all waves in step) is the -DCASMVS_ZM_WS=0 A/B build.  History:
// written against the CPU emulation at the end of round 3, first run on the MI355X in round 4 (tools/native/conv0_zm_check.cpp: against conv0_sf_kernel
// and a float64 convolution; profiles/r04_native_checks_first_run.txt).
//
// Why.  conv0_sf_kernel (conv0_splitf16.hip) is bound by its L1 -> L2 request stream: a 4 x 4 x 32 output tile stages a 6 x 6 x 40 halo
// box per chunk of 8 input channels - 2.8 staged voxels per output voxel, each row of 40 floats touching three cache lines - and without
// its global loads the kernel runs in half the time (profiles/r03_conv0_splitf16_ablations.txt).  Here a workgroup owns a 16 (y) x 32 (x)
// patch and MARCHES along z: every input plane of the patch (18 x 40 with the halo) is staged ONCE per chunk and feeds the three
// output planes it touches, whose accumulators rotate through registers.  1.41 staged voxels per output voxel (the y / x halo only),
// half the split work, a third of the row reads from LDS; the matrix instructions per output are the same.
//
// Arithmetic: that of conv0_splitf16.hip (w' = 2^kw w on the host; x' = 2^kx x with kx chosen per STAGED UNIT - here one plane patch of one
// chunk - so that max |x'| is in [2^14, 2^15); a = f16(x'), b = f16(x' - a); products aa, ab, ba on v_mfma_f32_16x16x32_f16 with float32
// accumulation; the unit's matrix result times 2^-kx added to the output plane's float32 accumulator).  The packed weight image is the one
// casmvs_conv0_splitf16_pack writes.  Results differ from conv0_sf_kernel's in the last bits (other staging units, other summation
// grouping); both sit ~3e-7 of the range from a float64 convolution (host model: tests/kernel_model.py: emulate_conv0_zmarch).
//
// Shape of the work: item = (sample, y tile, x tile, z segment); segments only where the (y, x) patches alone would not fill the chip
// (a segment re-stages one halo plane at each end).  All chunks' lane images stay in LDS (18 KiB per chunk) next to ONE plane patch
// (23 KiB): cin = 8 / 16 -> 41 / 59 KiB, two workgroups per CU (the request is padded to 56 KiB so that never three share one: the
// rule for f16 matrix kernels).  cin = 32 is 95 KiB: one workgroup per CU (instantiated so that the first test can time it
// against conv0_sf_kernel; expected to stay on the tiled kernel).
// Matrix phase of a unit: the wave's six staged rows are read once (12 x 16 B per lane), then 3 (kz) x 3 (ky) x 4 (rows) x 3 partial
// products = 108 MFMAs between which only LDS reads of the lane images are issued (no floating-point vector work: DESIGN.md 2.0).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "buffer_ops.h"
#include "common.h"
#include "split_f16.h"

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// WIDE: patches of 8 (y) x 64 (x) instead of 16 x 32.  The kernel is bound by its HBM-side read traffic (PMC, batch 8: 2.3x the input at cin = 16 with
// 16 x 32 patches - profiles/r04_pmc_traffic.md): a staged row of 40 floats from x0 - 4 touches its own 128-byte line plus a sector of each x neighbour's
// for one halo float each; a row of 72 floats pays the same two sectors for 256 bytes of its own.  Measured (tools/native/conv0_zm_check.cpp, batch 8,
// dirtied caches, profiles/r04_conv0_zm_wide_ab.txt): cin 16 830 -> 738 us, cin 8 454 -> 391 us (the tiled kernel: 1116 / 495); cin 32 (one workgroup per
// CU, W = 160 = 2.5 patches) 709 -> 841: the wide form is the default where cin <= 16.
template <bool WIDE_>
struct ZmCfgT {
  static constexpr int THREADS = 256, NT = 4;
  static constexpr bool WIDE = WIDE_;
  static constexpr int TY = WIDE ? 8 : 16, TX = WIDE ? 64 : 32;
  static constexpr int IY = TY + 2, IX = TX + 8, ROW = IX + 1;    // x0 - 4 .. x0 + 35 (16-byte aligned global groups); odd row stride
  static constexpr int NV = IY * ROW;                               // 16-byte slots per slice: 738
  static __host__ __device__ constexpr int slot(int x) { return x ^ (((x >> 3) & 1) << 1); }   // as SfCfg::slot
  static constexpr int ITEMS = IY * (IX / 4);                       // (y, group of 4 x) staging items: 180 of the 256 threads
  static constexpr int WUNITS = 9 * 2 * 64;                         // 16-byte units of a chunk's lane images
  static constexpr size_t ACT_BYTES = (size_t)2 * NV * 16;          // 23 616
  static __host__ __device__ constexpr size_t w_bytes(int nch) { return (size_t)nch * WUNITS * 16; }
  static __host__ __device__ constexpr size_t lds_bytes(int nch) {
    return ACT_BYTES + w_bytes(nch) + 16 < CASMVS_SF_LDS_FLOOR ? (size_t)CASMVS_SF_LDS_FLOOR : ACT_BYTES + w_bytes(nch) + 16;
  }
};

template <int N, class F>
__device__ __forceinline__ void zm_static_for(F &&f) {
  if constexpr (N > 0) {
    zm_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__device__ __forceinline__ f32x4 zm_mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct ZmItem {
  int tx0, ty0, zs, ze, b;
};

// in (B, CIN, D, H, W) float32, W % 4 == 0, 16-byte aligned; wpk: the image of casmvs_conv0_splitf16_pack; out (B, 8, D, H, W).
template <int CIN, bool WIDE>
__global__ __launch_bounds__(ZmCfgT<WIDE>::THREADS, 2) void conv0_zm_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                                    float *__restrict__ out, int B, int D, int H, int W, int tiles_x,
                                                                    int tiles_y, int nseg, int zlen, float slope) {
  using Cfg = ZmCfgT<WIDE>;
  constexpr int NCH = CIN / 8, NT = Cfg::NT, IX = Cfg::IX, ROW = Cfg::ROW, NV = Cfg::NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                              // [2][NV]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                              // [chunk][9][2][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::w_bytes(NCH));   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, u = lane >> 4;
  const int total = tiles_x * tiles_y * nseg * B;
  if ((int)blockIdx.x >= total) return;
  const int HW = H * W, cs = D * HW;
  const size_t in_ss = (size_t)CIN * cs, out_ss = (size_t)8 * cs;
  const float *tail = reinterpret_cast<const float *>(wpk + Cfg::w_bytes(NCH));
  float sc[2], sh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sc[h] = tail[2 * u + h];
    sh[h] = tail[8 + 2 * u + h];
  }
  const rsrc_t none = make_rsrc(in, 0);
  // every chunk's lane images, once per workgroup (visible after the first unit's first barrier)
  for (int unit = tid; unit < NCH * Cfg::WUNITS; unit += Cfg::THREADS) wl[unit] = reinterpret_cast<const u32x4 *>(wpk)[unit];

  // lane's B voxel of the wave's first staged row: the wave owns four output rows (staged rows wy .. wy + 5) of a 32-voxel x range starting at wx:
  // 16 x 32 patches: rows 4 wave, the whole width; 8 x 64 patches: rows 4 (wave >> 1), x half (wave & 1)
  const int wy = Cfg::WIDE ? 4 * (wave >> 1) : 4 * wave, wx = Cfg::WIDE ? 32 * (wave & 1) : 0;
  const int vbase = wy * ROW + Cfg::slot(wx + 2 * jcol + u + 3);

  auto decode = [&](int v) {
    int item = xcd_major(v, total);   // x fastest, then the z segment, then y
    ZmItem t;
    t.tx0 = (item % tiles_x) * Cfg::TX;
    item /= tiles_x;
    const int seg = item % nseg;
    item /= nseg;
    t.ty0 = (item % tiles_y) * Cfg::TY;
    t.b = item / tiles_y;
    t.zs = seg * zlen;
    t.ze = min(t.zs + zlen, D);
    return t;
  };
  // staging plan of the current prefetch target: item e = tid -> (iy, 4-x group); the plane and the channel come as a scalar offset
  int voff, vox, vxor;
  auto plan = [&](const ZmItem &tc) {
    const int e = tid, iy = e / (IX / 4), g = e - iy * (IX / 4);
    const int gy = tc.ty0 - 1 + iy, gx = tc.tx0 - 4 + 4 * g;
    const bool ok = e < Cfg::ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 4 == 0
    voff = ok ? (gy * W + gx) * 4 : kOOB;
    vox = e < Cfg::ITEMS ? iy * ROW + 4 * g : -1;
    vxor = ((g >> 1) & 1) << 1;
  };
  f32x4v R[8];
  auto prefetch = [&](const ZmItem &tc, int zi, int chunk, bool exists) {   // every load of one unit; nothing here waits
    const rsrc_t src = exists ? make_rsrc(in + (size_t)tc.b * in_ss, in_ss * 4) : none;
#pragma unroll
    for (int c = 0; c < 8; ++c) R[c] = buf_load4(src, voff, ((chunk * 8 + c) * cs + zi * HW) * 4);
  };

  // acc[0] / [1] / [2]: output planes zi - 1 / zi / zi + 1 while input plane zi is being processed
  f32x4 acc[3][NT];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  int item = blockIdx.x;
  ZmItem cur = decode(item);
  plan(cur);
  prefetch(cur, max(cur.zs - 1, 0), 0, true);
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool have_next = next_item < total;
    const ZmItem nxt = have_next ? decode(next_item) : cur;
    const int zhi = min(cur.ze, D - 1);   // the last plane of this item that exists
    const rsrc_t dst = make_rsrc(out + (size_t)cur.b * out_ss, out_ss * 4);
#pragma unroll 1
    for (int zi = cur.zs - 1; zi <= cur.ze; ++zi) {
      if (zi >= 0 && zi < D) {
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch) {
          // ---- the staged unit's largest magnitude (this thread's loads -> wave -> workgroup) ----
          float m = 0.0f;
#pragma unroll
          for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) m = fmaxf(m, fabsf(R[c][j]));
          const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
          if (lane == 0) wmax[wave] = wm;
          __syncthreads();   // every wave is done with the previous unit's LDS; the four maxima are visible
          float mult, inv;
          casmvs::tile_scale(wmax, mult, inv);
          // ---- registers -> LDS: the two float16 slices of every staged voxel ----
          if (vox >= 0) {
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
          __syncthreads();
          // ---- the next unit's loads: next chunk of this plane, next plane, or the first plane of the next item ----
          if (ch + 1 < NCH) {
            prefetch(cur, zi, ch + 1, true);
          } else if (zi < zhi) {
            prefetch(cur, zi + 1, 0, true);
          } else {
            plan(nxt);
            prefetch(nxt, max(nxt.zs - 1, 0), 0, have_next);
          }
          // ---- matrix phase: six staged rows read once; 3 kz x 3 ky x 4 rows x 3 partial products ----
          __builtin_amdgcn_sched_barrier(0);   // the prefetch's address arithmetic is integer work, but keep the phase boundary explicit
          u32x4 row[NT + 2][2];
#pragma unroll
          for (int yr = 0; yr < NT + 2; ++yr)
#pragma unroll
            for (int s = 0; s < 2; ++s) row[yr][s] = act[s * NV + vbase + yr * ROW];
          f32x4 part[3][NT];
#pragma unroll
          for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int t = 0; t < NT; ++t) part[kz][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              u32x4 a[2];
#pragma unroll
              for (int s = 0; s < 2; ++s) a[s] = wl[((ch * 9 + kz * 3 + ky) * 2 + s) * 64 + lane];
              constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
              for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < NT; ++t) part[kz][t] = zm_mfma(a[PA[p]], row[t + ky][PB[p]], part[kz][t]);
            }
          // input plane zi is tap kz of output plane zi + 1 - kz.  The fold is floating-point vector work: it must not be scheduled
          // between the matrix instructions above (the compiler did move the kz = 0 folds up) - pinned.
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[2 - kz][t][q] = fmaf(part[kz][t][q], inv, acc[2 - kz][t][q]);
        }
      }
      // ---- output plane zi - 1 has seen its three input planes: y = lrelu(acc * scale + shift); then the accumulators move up ----
      const int zo = zi - 1;
      const bool plane_ok = zo >= cur.zs && zo < cur.ze;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int oy = cur.ty0 + wy + t, ox = cur.tx0 + wx + 2 * jcol;
        const bool ok = plane_ok && oy < H && ox < W;   // W even: the pixel pair is inside or outside
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v0 = fmaf(acc[0][t][2 * h], sc[h], sh[h]), v1 = fmaf(acc[0][t][2 * h + 1], sc[h], sh[h]);
          v0 = v0 > 0.0f ? v0 : v0 * slope;
          v1 = v1 > 0.0f ? v1 : v1 * slope;
          buf_store2(f32x2{v0, v1}, dst, ok ? ((2 * u + h) * cs + (zo * H + oy) * W + ox) * 4 : kOOB, 0);
        }
        acc[0][t] = acc[1][t];
        acc[1][t] = acc[2][t];
        acc[2][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    // what the segment's last halo plane left behind belongs to planes of the next segment: drop it
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!have_next) break;
    item = next_item;
    cur = nxt;
  }
}

// ---- the same data flow with staging and multiplication overlapped INSIDE one workgroup (round 4) -------------------------------------------------
// conv0_zm_kernel overlaps the two phases of a unit across the TWO workgroups a CU holds; at cin = 32 the lane images (72 KiB) leave room for one
// workgroup only and the phases run back to back (0.74-1.04x the tiled kernel).  Here a workgroup is 8 waves: waves 4-7 (producers) load, scale, split
// and write unit n + 1 into one of TWO plane-patch buffers while waves 0-3 (consumers) multiply unit n out of the other; ONE workgroup barrier per
// unit (buffer n + 1 is complete, buffer n is free; the producers' wave maxima travel one iteration ahead through four LDS slots).  The producers keep the loads of CASMVS_ZW_NSET units in flight (one register set each).  Floating-point vector work is done by the producer waves and, outside
// their matrix phases, by the consumers' folds / epilogues - never between a wave's own matrix instructions.  One workgroup per CU
// (2 x 23 KiB + 18 KiB x cin / 8 of LDS), two waves per SIMD: a producer beside a consumer.
// Units, scaling, summation order and therefore the RESULT BITS are those of conv0_zm_kernel with the same patch shape.
#ifndef CASMVS_ZW_NSET
#define CASMVS_ZW_NSET 2
#endif
#ifndef CASMVS_ZW_PRIO
#define CASMVS_ZW_PRIO 0
#endif
#ifdef CASMVS_ZW_TRACE
// Profiling builds (tools/native/zw_trace.cpp): wave 0 of each role of workgroup 0 stamps the shader clock at the phase boundaries of its first units.
__device__ unsigned long long g_zw_trace[2][512];
#define ZW_STAMP(role)                                                                          \
  do {                                                                                          \
    if (blockIdx.x == 0 && w4 == 0 && zw_tn < 512) g_zw_trace[role][zw_tn++] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define ZW_STAMP(role) do {} while (0)
#endif
template <int CIN, bool WIDE>
__global__ __launch_bounds__(512, 1) void conv0_zw_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk, float *__restrict__ out,
                                                         int B, int D, int H, int W, int tiles_x, int tiles_y, int nseg, int zlen, float slope) {
  using Cfg = ZmCfgT<WIDE>;
  constexpr int NCH = CIN / 8, NT = Cfg::NT, IX = Cfg::IX, ROW = Cfg::ROW, NV = Cfg::NV;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                                  // [2 buffers][2 slices][NV]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + 2 * Cfg::ACT_BYTES);                              // [chunk][9][2][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + 2 * Cfg::ACT_BYTES + Cfg::w_bytes(NCH));   // [4 slots][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;          // wave-uniform
  const int w4 = wave & 3, t4 = tid & 255;  // wave / thread inside the role's group of four waves
  const int jcol = lane & 15, u = lane >> 4;
  const int total = tiles_x * tiles_y * nseg * B;
  if ((int)blockIdx.x >= total) return;
  const int HW = H * W, cs = D * HW;
  const size_t in_ss = (size_t)CIN * cs, out_ss = (size_t)8 * cs;
  for (int unit = tid; unit < NCH * Cfg::WUNITS; unit += 512) wl[unit] = reinterpret_cast<const u32x4 *>(wpk)[unit];

  auto decode = [&](int v) {
    int item = xcd_major(v, total);   // x fastest, then the z segment, then y
    ZmItem t;
    t.tx0 = (item % tiles_x) * Cfg::TX;
    item /= tiles_x;
    const int seg = item % nseg;
    item /= nseg;
    t.ty0 = (item % tiles_y) * Cfg::TY;
    t.b = item / tiles_y;
    t.zs = seg * zlen;
    t.ze = min(t.zs + zlen, D);
    return t;
  };
  // units of this workgroup: every (chunk, existing input plane zs - 1 .. ze) of its items item, item + gridDim.x, ...
  int nunits = 0;
  for (int it = blockIdx.x; it < total; it += gridDim.x) {
    const ZmItem t = decode(it);
    nunits += (min(t.ze, D - 1) - max(t.zs - 1, 0) + 1) * NCH;
  }

#ifdef CASMVS_ZW_TRACE
  int zw_tn = 0;
#endif
  if (producer) {
    // ---- producers: unit n -> buffer n & 1 in iteration n; the loads of units n + 1 and n + 2 are in flight meanwhile ----
#if CASMVS_ZW_PRIO
    __builtin_amdgcn_s_setprio(CASMVS_ZW_PRIO);   // the producer's vector instructions go first: the consumer beside it needs one issue slot per 16 cycles
#endif
    const rsrc_t none = make_rsrc(in, 0);
    struct Cursor {   // the unit whose loads are issued next
      int item, zi, ch, zhi;
      ZmItem t;
      bool valid;
    };
    Cursor pc;
    pc.item = blockIdx.x;
    pc.t = decode(pc.item);
    pc.zi = max(pc.t.zs - 1, 0);
    pc.zhi = min(pc.t.ze, D - 1);
    pc.ch = 0;
    pc.valid = true;
    int voff;
    const int e = t4, iy = e / (IX / 4), g = e - iy * (IX / 4);
    const int vox = e < Cfg::ITEMS ? iy * ROW + 4 * g : -1;
    const int vxor = ((g >> 1) & 1) << 1;
    const int scratch_unit = (int)((2 * Cfg::ACT_BYTES + Cfg::w_bytes(NCH) + 64) / 16) + 2 * t4;   // two 16-byte units per producer thread, never read
    auto plan = [&](const ZmItem &tc) {
      const int gy = tc.ty0 - 1 + iy, gx = tc.tx0 - 4 + 4 * g;
      const bool ok = e < Cfg::ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 4 == 0
      voff = ok ? (gy * W + gx) * 4 : kOOB;
    };
    plan(pc.t);
    auto advance = [&]() {   // -> the next unit (chunk, then plane, then the first plane of the next item)
      if (++pc.ch < NCH) return;
      pc.ch = 0;
      if (pc.zi < pc.zhi) { ++pc.zi; return; }
      pc.item += gridDim.x;
      pc.valid = pc.item < total;
      if (pc.valid) {
        pc.t = decode(pc.item);
        pc.zi = max(pc.t.zs - 1, 0);
        pc.zhi = min(pc.t.ze, D - 1);
        plan(pc.t);
      }
    };
    constexpr int NSET = CASMVS_ZW_NSET;   // register sets = units whose loads are in flight
    f32x4v R[NSET][8];
    auto issue = [&](auto set_) {   // the loads of the cursor's unit into register set S; then the cursor moves on
      constexpr int S = decltype(set_)::value;
      const rsrc_t src = pc.valid ? make_rsrc(in + (size_t)pc.t.b * in_ss, in_ss * 4) : none;
#pragma unroll
      for (int c = 0; c < 8; ++c) R[S][c] = buf_load4(src, voff, ((pc.ch * 8 + c) * cs + pc.zi * HW) * 4);
      if (pc.valid) advance();
    };
    zm_static_for<NSET>([&](auto k_) { issue(k_); });
    // Every iteration issues exactly the same vector-memory operations on every path (beyond the last unit the loads go through the empty descriptor:
    // zeros, no memory access - and the zeros are staged into a buffer nobody reads any more): the compiler's wait-count pass can then count
    // instead of draining everything.  A conditional `issue` - or a branch around the LDS writes - made the number of loads behind a set
    // path-dependent and the only safe wait vmcnt(0): the memory latency was exposed once per unit (shader-clock trace: 2 200-2 500 of a unit's
    // 5 800 cycles in "issue", profiles/r04_conv0_zw_trace.txt).
    // ONE workgroup barrier per unit: iteration n splits unit n with the maxima published one iteration earlier, then takes the maximum of unit
    // n + 1 (its loads were issued NSET - 1 iterations ago) and publishes it (four slots: the consumers read unit n's maxima during iteration
    // n + 1).  With a second barrier between "maximum" and "split" the consumers - who have to join every barrier - serialised their first matrix
    // instructions with the producers' split (trace: 1 400 of a unit's 4 200 cycles waiting).
    auto unit_max = [&](auto set_, int slot) {
      constexpr int S = decltype(set_)::value;
      float m0 = 0.0f, m1 = 0.0f;   // two chains: the maximum of 32 values is one dependent chain otherwise
#pragma unroll
      for (int c = 0; c < 8; c += 2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          m0 = fmaxf(m0, fabsf(R[S][c][j]));
          m1 = fmaxf(m1, fabsf(R[S][c + 1][j]));
        }
      const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, fmaxf(m0, m1)));
      if (lane == 0) wmax[(slot & 3) * 4 + w4] = wm;
    };
    unit_max(std::integral_constant<int, 0>{}, 0);
    __syncthreads();   // P: unit 0's maxima (the consumers join)
    auto stage = [&](auto set_, int n) {
      constexpr int S = decltype(set_)::value;
      const int par = n & 1;
      ZW_STAMP(1);   // p0: top of the iteration
      float mult, inv;
      casmvs::tile_scale(wmax + (n & 3) * 4, mult, inv);
      // threads without a staging item (76 of the 256) split the zeros their out-of-range loads returned into a per-thread scratch slot: no branch
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = R[S][c][j];
        u32x4 o[2];
        casmvs::split8_f16(x, mult, o);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) act[(vox >= 0 ? (par * 2 + sl) * NV + vox + (j ^ vxor) : scratch_unit + sl)] = o[sl];
      }
      ZW_STAMP(1);   // p1: split + LDS writes issued
      unit_max(std::integral_constant<int, (S + 1) % NSET>{}, n + 1);
      ZW_STAMP(1);   // p2: the next unit's loads have landed, its maximum is published
      issue(set_);   // unit n + NSET into the set that has just been emptied
      ZW_STAMP(1);   // p3: before the barrier
      __syncthreads();   // B
    };
    const int iters = (nunits + 1 + NSET - 1) / NSET * NSET;   // nunits + 1 iterations (the last keeps the consumers' final unit company), padded to whole rounds
    for (int n = 0; n < iters; n += NSET) zm_static_for<NSET>([&](auto k_) { stage(k_, n + decltype(k_)::value); });
    return;
  }

  // ---- consumers: unit n out of buffer n & 1 in iteration n + 1 ----
  const float *tail = reinterpret_cast<const float *>(wpk + Cfg::w_bytes(NCH));
  float sc[2], sh[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    sc[h] = tail[2 * u + h];
    sh[h] = tail[8 + 2 * u + h];
  }
  const int wy = Cfg::WIDE ? 4 * (w4 >> 1) : 4 * w4, wx = Cfg::WIDE ? 32 * (w4 & 1) : 0;
  const int vbase = wy * ROW + Cfg::slot(wx + 2 * jcol + u + 3);
  f32x4 acc[3][NT];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // P: the producers publish unit 0's maxima
  __syncthreads();   // B of iteration 0: the producers stage unit 0
  int n = 0;
  for (int item = blockIdx.x; item < total; item += gridDim.x) {
    const ZmItem cur = decode(item);
    const rsrc_t dst = make_rsrc(out + (size_t)cur.b * out_ss, out_ss * 4);
#pragma unroll 1
    for (int zi = cur.zs - 1; zi <= cur.ze; ++zi) {
      if (zi >= 0 && zi < D) {
#pragma unroll 1
        for (int ch = 0; ch < NCH; ++ch, ++n) {
          ZW_STAMP(0);   // c0: top of the unit
          const u32x4 *buf = act + (n & 1) * 2 * NV;
          u32x4 row[NT + 2][2];
#pragma unroll
          for (int yr = 0; yr < NT + 2; ++yr)
#pragma unroll
            for (int s = 0; s < 2; ++s) row[yr][s] = buf[s * NV + vbase + yr * ROW];
          f32x4 part[3][NT];
#pragma unroll
          for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int t = 0; t < NT; ++t) part[kz][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          // the lane images of group g + 1 are read BEFORE the 12 matrix instructions of group g are issued (pinned): this wave is the only one that
          // feeds its SIMD's matrix core, an LDS round trip in front of every group (what the compiler's own placement gave) left it idle 9 x per unit
          u32x4 a[2][2];
#pragma unroll
          for (int s = 0; s < 2; ++s) a[0][s] = wl[((ch * 9) * 2 + s) * 64 + lane];
#pragma unroll
          for (int g = 0; g < 9; ++g) {
            const int kz = g / 3, ky = g % 3;
            if (g + 1 < 9) {
#pragma unroll
              for (int s = 0; s < 2; ++s) a[(g + 1) & 1][s] = wl[((ch * 9 + g + 1) * 2 + s) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
              for (int t = 0; t < NT; ++t) part[kz][t] = zm_mfma(a[g & 1][PA[p]], row[t + ky][PB[p]], part[kz][t]);
            __builtin_amdgcn_sched_barrier(0);
          }
          ZW_STAMP(0);   // c1: all matrix instructions issued
          __builtin_amdgcn_sched_barrier(0);   // the folds are floating-point vector work: after the unit's last matrix instruction
          float mult, inv;
          casmvs::tile_scale(wmax + (n & 3) * 4, mult, inv);
#pragma unroll
          for (int kz = 0; kz < 3; ++kz)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[2 - kz][t][q] = fmaf(part[kz][t][q], inv, acc[2 - kz][t][q]);
          ZW_STAMP(0);   // c2: folds done (the matrix results have arrived)
          __syncthreads();   // B of iteration n + 1: buffer n & 1 is free
        }
      }
      const int zo = zi - 1;
      const bool plane_ok = zo >= cur.zs && zo < cur.ze;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int oy = cur.ty0 + wy + t, ox = cur.tx0 + wx + 2 * jcol;
        const bool ok = plane_ok && oy < H && ox < W;   // W even: the pixel pair is inside or outside
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v0 = fmaf(acc[0][t][2 * h], sc[h], sh[h]), v1 = fmaf(acc[0][t][2 * h + 1], sc[h], sh[h]);
          v0 = v0 > 0.0f ? v0 : v0 * slope;
          v1 = v1 > 0.0f ? v1 : v1 * slope;
          buf_store2(f32x2{v0, v1}, dst, ok ? ((2 * u + h) * cs + (zo * H + oy) * W + ox) * 4 : kOOB, 0);
        }
        acc[0][t] = acc[1][t];
        acc[1][t] = acc[2][t];
        acc[2][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // the producers run whole rounds of CASMVS_ZW_NSET iterations: keep their padding iterations' barriers company
  for (int k = (nunits + 1) % CASMVS_ZW_NSET; k != 0 && k < CASMVS_ZW_NSET; ++k) __syncthreads();
}

// Measured (tools/native/conv0_zm_check.cpp, batch 8, dirtied caches): the first version - two barriers per unit, conditional loads - 651 us at cin 32
// against 704 (two-phase z-march) and 751-771 (tiled), and slower than the two-phase kernel at cin 16 / 8 (profiles/r04_conv0_zw_ab.txt).  A shader-clock
// trace (tools/native/zw_trace.cpp, profiles/r04_conv0_zw_trace.txt) showed why: s_waitcnt vmcnt(0) in front of every unit's loads (conditional vector-memory
// operations) and the consumers' matrix phase serialised with the producers' split by the second barrier.  With unconditional iterations and ONE barrier
// per unit: cin 32 523-530 us (1.44x the tiled kernel), cin 16 685 (two-phase 784, tiled 1109-1120), cin 8 381 (397, 502-522); whole step 7.72 -> 7.57 ms
// (profiles/r04_conv0_zw_single_barrier_ab.txt).  It is the conv0 kernel of every cascade level; conv0_zm_kernel stays in the source for A/B builds
// (-DCASMVS_ZM_WS=0) and is not instantiated by default.
// Tried on top and rejected (profiles/r04_conv0_zw_producer_ab.txt): the sample's buffer descriptor and a running byte offset kept in the producers'
// cursor instead of being recomputed per unit (the descriptor then lives in vector registers: a waterfall loop around every load, 573 / 765 us at
// cin 32 / 16); the two scale multiplies of a pair as one v_pk_mul_f32 (653 / 800 us); raised producer priority (s_setprio 3: +-1 %); three or four
// register sets in flight instead of two (517-523 us at cin 32: the loads are already hidden).
#ifndef CASMVS_ZM_WS
#define CASMVS_ZM_WS 7   // which channel counts run the warp-specialised form: bit 0 cin 8, bit 1 cin 16, bit 2 cin 32 (A/B builds: other values)
#endif
#ifndef CASMVS_ZM_WS32_WIDE
#define CASMVS_ZM_WS32_WIDE 0   // A/B builds: 8 x 64 patches for the warp-specialised cin = 32 kernel
#endif

template <int CIN, bool WIDE>
int launch_zw(const void *packed, const float *in, float *out, int B, int D, int H, int W, float slope, hipStream_t st) {
  using Cfg = ZmCfgT<WIDE>;
  constexpr int NCH = CIN / 8;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY);
  auto kernel = conv0_zw_kernel<CIN, WIDE>;
  const size_t lds = 2 * Cfg::ACT_BYTES + Cfg::w_bytes(NCH) + 64 + 256 * 2 * 16;   // + the producers' scratch units
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), lds, "conv0_zw_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), 512, lds);
  const long patches = (long)B * tiles_x * tiles_y;
  int nseg = 1;
  while (patches * nseg < 4L * resident && D / (nseg + 1) >= 4) ++nseg;
  const int zlen = casmvs::ceil_div(D, nseg);
  nseg = casmvs::ceil_div(D, zlen);
  const long total = patches * nseg;
  CASMVS_REQUIRE(total < (1L << 31), "conv0_zmarch_forward: too many items");
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(512), lds, st, in,
                     reinterpret_cast<const unsigned char *>(packed), out, B, D, H, W, tiles_x, tiles_y, nseg, zlen, slope);
  return casmvs::check_launch("conv0_zw_kernel");
}

template <int CIN, bool WIDE>
int launch_zm(const void *packed, const float *in, float *out, int B, int D, int H, int W, float slope, hipStream_t st) {
  using Cfg = ZmCfgT<WIDE>;
  constexpr int NCH = CIN / 8;
  const int tiles_x = casmvs::ceil_div(W, Cfg::TX), tiles_y = casmvs::ceil_div(H, Cfg::TY);
  auto kernel = conv0_zm_kernel<CIN, WIDE>;
  const size_t lds = Cfg::lds_bytes(NCH);
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), lds, "conv0_zm_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, lds);
  // z segments only where the (y, x) patches alone leave workgroups idle or unevenly loaded: aim at >= 4 items per resident workgroup,
  // segments of at least 4 planes (each one re-stages a halo plane at both ends)
  const long patches = (long)B * tiles_x * tiles_y;
  int nseg = 1;
  while (patches * nseg < 4L * resident && D / (nseg + 1) >= 4) ++nseg;
  const int zlen = casmvs::ceil_div(D, nseg);
  nseg = casmvs::ceil_div(D, zlen);
  const long total = patches * nseg;
  CASMVS_REQUIRE(total < (1L << 31), "conv0_zmarch_forward: too many items");
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), lds, st, in,
                     reinterpret_cast<const unsigned char *>(packed), out, B, D, H, W, tiles_x, tiles_y, nseg, zlen, slope);
  return casmvs::check_launch("conv0_zm_kernel");
}

}  // namespace

extern "C" int casmvs_conv0_zmarch_supported(int cin, int W) { return (cin == 8 || cin == 16 || cin == 32) && W % 4 == 0 && W >= 4; }

extern "C" int casmvs_conv0_zmarch_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int D, int H, int W,
                                               float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "conv0_zmarch_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && casmvs_conv0_zmarch_supported(cin, W), "conv0_zmarch_forward: B=%d cin=%d D=%d H=%d W=%d", B, cin, D, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out) | reinterpret_cast<size_t>(packed)) & 15) == 0, "conv0_zmarch_forward: 16-byte aligned pointers");
  CASMVS_REQUIRE((size_t)cin * D * H * W < ((size_t)1 << 29), "conv0_zmarch_forward: one sample's input tensor must hold < 2^29 floats");
  hipStream_t st = (hipStream_t)stream;
  if (cin == 8) {
    if constexpr ((CASMVS_ZM_WS & 1) != 0) return launch_zw<8, true>(packed, in, out, B, D, H, W, slope, st);
    else return launch_zm<8, true>(packed, in, out, B, D, H, W, slope, st);
  }
  if (cin == 16) {
    if constexpr ((CASMVS_ZM_WS & 2) != 0) return launch_zw<16, true>(packed, in, out, B, D, H, W, slope, st);
    else return launch_zm<16, true>(packed, in, out, B, D, H, W, slope, st);
  }
  if constexpr ((CASMVS_ZM_WS & 4) != 0) return launch_zw<32, CASMVS_ZM_WS32_WIDE != 0>(packed, in, out, B, D, H, W, slope, st);
  else return launch_zm<32, false>(packed, in, out, B, D, H, W, slope, st);   // 95 KiB of LDS: ONE workgroup per CU (measured: 0.74-0.9x the tiled kernel; the engine keeps cin = 32 tiled)
}

#ifdef CASMVS_ZW_TRACE
extern "C" int casmvs_zw_trace_read(unsigned long long *host) {
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_zw_trace), sizeof(unsigned long long) * 2 * 512) == hipSuccess ? 0 : -3;
}
#endif
