// CostRegNet's stride-2 layers conv1 (8 -> 16) and conv3 (16 -> 32): Conv3d k3 s2 p1 + folded ABN + leaky-relu on the f16 matrix cores with
// float32-grade arithmetic - the arithmetic of conv0_splitf16.hip (every float32 operand = two float16 slices behind exact power-of-two scalings,
// three partial products, float32 accumulation), the data flow of conv0_zmarch.hip (input-stationary along z).
//
// Reference semantics: models/mvsnet.py:64-65,67-68,92-93 (`conv1`, `conv3`), models/modules.py:21-31 (ConvBnReLU3D).
//
// Why: on the float32-input MFMA (which issues at the float32 VECTOR rate) these layers sit at 0.4-0.47 of its peak, about twice both their HBM time and
// their matrix time; the f16 instruction does the same product in 3/16 of the matrix-pipe time, which leaves the input volume's one pass
// through HBM as the bound - if every input voxel is fetched about once.  Output o reads inputs 2 o - 1 .. 2 o + 1 per axis: a tile kernel with a
// 2 x 4 x 16 output tile stages 1.5 input voxels per input voxel; marching z with the input plane stationary stages 13 / 12 x 68 / 64 = 1.15.
//
// Formulation: D[16 x 16] += A[16 x 32] B[32 x 16] with
//   rows    i = output channel 16 rb + i
//   columns j = 16 consecutive output x of one output row
//   K       k = (kx = k >> 3, ci = k & 7)   - the 3 x taps (the 4th block: zero weights) x one chunk of 8 input channels; a step = one (kz, ky)
// Lane l = (j = l & 15, kb = l >> 4) supplies the 8 channels of input voxel x = 2 (ox0 + j) - 1 + kb: ONE 16-byte LDS read.  The staged plane keeps even
// and odd x in separate halves of a row, so that the 16 lanes of a kb group read 16 consecutive units.
//
// Workgroup = 256 threads (4 waves) owns an output patch of 6 rows x 32 columns and a segment of output planes; it walks the input planes
// 2 oz0 - 1 .. 2 (oz0 + ZT) - 1 once.  A staged UNIT = (input plane, chunk of 8 channels): 13 rows x 68 x (x from 2 ox0 - 4: whole 16-byte quads) with its
// own power-of-two scale, 27.2 KiB as [slice][row][even x | odd x].  An even plane 2 oz is tap kz = 1 of output plane oz; an odd plane 2 oz + 1 is tap
// kz = 2 of oz and tap kz = 0 of oz + 1 (two accumulator sets; A is stored and B becomes A after every odd plane).  Wave w owns column tiles 3 w .. 3 w + 2
// of the patch's 12 (row = tile >> 1, half = tile & 1).  A register set is refilled with the unit DEPTH ahead as soon as the matrix phase of its unit is done.
// All lane images stay in LDS: conv1 18 KiB (two workgroups per CU), conv3 72 KiB (one per CU); two units in flight per workgroup (two register sets).
#include <cmath>
#include <cstdint>
#include <cstring>

#include "buffer_ops.h"
#include "common.h"
#include "split_f16.h"

#ifndef CASMVS_S2_DEPTH_CONV1
#define CASMVS_S2_DEPTH_CONV1 2   // units in flight per workgroup (register sets of 32): conv1 runs two workgroups per CU,
#endif
#ifndef CASMVS_S2_DEPTH_CONV3
#define CASMVS_S2_DEPTH_CONV3 2   // conv3 (72 KiB of lane images) one; three were slower (profiles/r04_conv_s2_depth_ab.txt)
#endif

namespace {

using namespace casmvs::buf;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int CIN, int COUT>
struct S2Cfg {
  static constexpr int THREADS = 256, WAVES = 4;
  static constexpr int TY = 6, TX = 32, NT = 3;                     // output patch; column tiles per wave
  static constexpr int IY = 2 * TY + 1, IQ = 17;                     // staged rows 2 oy0 - 1 .. 2 oy0 + 11; quads of x from 2 ox0 - 4 (68 floats)
  // a staged row: even x (staged index 2 i) at unit i, odd x (2 i + 1) at unit ODD + i, i < 34.  ODD = 33 = 1 (mod 16) puts the units a 16-byte read's
  // lane groups fetch together - 8 lanes of one kb group, 8 of its neighbour: odd index J + 1 with even index J + 2 - on 16 different 16-byte bank groups;
  // the unit it shares with even index 33 is odd index 0 = x 2 ox0 - 3, which no tap reads and the staging does not write
  static constexpr int ODD = 33, RS = ODD + 2 * IQ;                  // units per row: 67
  static constexpr int NVOX = IY * RS;                               // units per slice: 871
  static constexpr int ITEMS = IY * IQ;                              // (row, quad) staging items: 221 of the 256 threads
  static constexpr int NCH = CIN / 8, RB = COUT / 16;
  static constexpr int WUNITS = NCH * 9 * RB * 2 * 64;               // lane images [chunk][kz][ky][row block][slice][lane]
  static constexpr int NWL = (WUNITS + THREADS - 1) / THREADS;
  static constexpr size_t ACT_BYTES = (size_t)2 * NVOX * 16, W_BYTES = (size_t)WUNITS * 16;   // 27 872 + 18 432 (conv1) / 73 728 (conv3)
  // At most TWO workgroups per CU, as every split-f16 kernel of the library (conv2d_ci_splitf16.hip: the configuration with three waves per SIMD is the
  // one in which round 3 saw wrong float32 results): the requested LDS is padded to 56 KiB.  Measured: three workgroups with one unit in flight each
  // 117 / 230 / 232 us at the three levels (batch 8), two with two units each 120 / 236 / 233 (profiles/r04_conv_s2_depth_ab.txt).
  static constexpr size_t LDS_USED = ACT_BYTES + W_BYTES + 16;
  static constexpr size_t LDS_BYTES = LDS_USED < CASMVS_SF_LDS_FLOOR ? (size_t)CASMVS_SF_LDS_FLOOR : LDS_USED;
  static constexpr int WG_PER_CU = LDS_BYTES * 2 <= (size_t)160 * 1024 ? 2 : 1;
  static_assert(CIN % 8 == 0 && COUT % 16 == 0 && TY * (TX / 16) == WAVES * NT && ITEMS <= THREADS, "shape");
};

__device__ __forceinline__ f32x4 s2_mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct S2Item {   // an output patch and its segment of output planes
  int ox0, oy0, oz0, b;
};
template <typename Cfg>
__device__ __forceinline__ S2Item s2_decode(int v, int total, int tiles_x, int tiles_y, int segs, int zt) {
  int item = xcd_major(v, total);   // x fastest, then y (neighbouring patches share halo rows inside an XCD's L2), then the z segment
  S2Item t;
  t.ox0 = (item % tiles_x) * Cfg::TX;
  item /= tiles_x;
  t.oy0 = (item % tiles_y) * Cfg::TY;
  item /= tiles_y;
  t.oz0 = (item % segs) * zt;
  t.b = item / segs;
  return t;
}

// in (B, CIN, D, H, W) float32, W % 4 == 0, 16-byte aligned; wpk: [chunk][kz][ky][row block][slice][lane] 16-byte lane images, then
// scale[COUT] (ABN scale x 2^-kw), shift[COUT]; out (B, COUT, Do, Ho, Wo), o = (i - 1) / 2 + 1 per axis.
template <int CIN, int COUT, int DEPTH>
__global__ __launch_bounds__(256, (S2Cfg<CIN, COUT>::WG_PER_CU)) void conv_s2_sf_kernel(const float *__restrict__ in, const unsigned char *__restrict__ wpk,
                                                                                   float *__restrict__ out, int B, int D, int H, int W, int tiles_x,
                                                                                   int tiles_y, int segs, int zt, float slope) {
  using Cfg = S2Cfg<CIN, COUT>;
  constexpr int NCH = Cfg::NCH, RB = Cfg::RB, NT = Cfg::NT, NWL = Cfg::NWL, NVOX = Cfg::NVOX, RS = Cfg::RS, ODD = Cfg::ODD, IQ = Cfg::IQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4 *act = reinterpret_cast<u32x4 *>(smem_raw);                                          // [slice][row][even x: 34 | odd x from unit 33]
  u32x4 *wl = reinterpret_cast<u32x4 *>(smem_raw + Cfg::ACT_BYTES);                          // [chunk][kz][ky][rb][slice][64]
  unsigned *wmax = reinterpret_cast<unsigned *>(smem_raw + Cfg::ACT_BYTES + Cfg::W_BYTES);   // [4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jcol = lane & 15, kb = lane >> 4;
  const int total = tiles_x * tiles_y * segs * B;
  if ((int)blockIdx.x >= total) return;
  const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int HW = H * W, cs = D * HW, ocs = Do * Ho * Wo;
  const size_t in_ss = (size_t)CIN * cs, out_ss = (size_t)COUT * ocs;
  const float *tail = reinterpret_cast<const float *>(wpk + Cfg::W_BYTES);
  float sc[RB][4], sh[RB][4];   // lane holds rows 4 kb + r of every row block
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sc[rb][r] = tail[rb * 16 + 4 * kb + r];
      sh[rb][r] = tail[COUT + rb * 16 + 4 * kb + r];
    }
  // ---- every lane image into LDS, once per workgroup (the first unit's barriers publish it) ----
  {
    const rsrc_t wsrc = make_rsrc(reinterpret_cast<const float *>(wpk), Cfg::W_BYTES);
    u32x4 WR[NWL];   // all loads first: one round trip, not NWL
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int unit = tid + i * Cfg::THREADS;
      WR[i] = __builtin_bit_cast(u32x4, buf_load4(wsrc, unit < Cfg::WUNITS ? unit * 16 : kOOB, 0));
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int unit = tid + i * Cfg::THREADS;
      if (Cfg::WUNITS % Cfg::THREADS == 0 || i + 1 < NWL || unit < Cfg::WUNITS) wl[unit] = WR[i];
    }
  }
  const rsrc_t none = make_rsrc(in, 0);

  // lane's B unit (slice 0, step ky = 0) of column tile t: staged x index 2 (16 hf + j) + 3 + kb -> (parity, index) = kb 0: (1, J + 1), 1: (0, J + 2),
  // 2: (1, J + 2); kb 3 (zero weights) reads kb 2's kind of unit: staged, finite data, and no bank shared with the kb 2 lanes of its group
  int vb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int tile = wave * NT + t, row = tile >> 1, hf = tile & 1;
    vb[t] = 2 * row * RS + (kb == 1 ? 0 : ODD) + 16 * hf + jcol + (kb == 0 ? 1 : 2);
  }

  // staging item of this thread: (row, quad of x)
  const bool staged = tid < Cfg::ITEMS;
  const int srow = tid / IQ, sq = tid - srow * IQ;
  const int sunit = srow * RS + 2 * sq;   // voxel v of the quad -> parity v & 1, index 2 q + (v >> 1)

  // a unit = (item, input plane p, chunk); DEPTH register sets hold the units in flight (the loads of a unit: 8 channels x one quad of x per staging thread)
  struct Cursor {
    S2Item it;
    int item, p, ch;
    bool valid;
  };
  auto advance = [&](const Cursor &c) {
    Cursor n = c;
    if (!c.valid) return n;
    n.ch = c.ch + 1;
    if (n.ch == NCH) {
      n.ch = 0;
      n.p = c.p + 1;
      if (n.p > 2 * (c.it.oz0 + zt) - 1) {   // past the odd plane that completes the segment's last output plane (planes >= D read as zeros)
        n.item = c.item + gridDim.x;
        n.valid = n.item < total;
        if (n.valid) n.it = s2_decode<Cfg>(n.item, total, tiles_x, tiles_y, segs, zt);
        n.p = n.it.oz0 == 0 ? 0 : 2 * n.it.oz0 - 1;
      }
    }
    return n;
  };
  f32x4 R[DEPTH][8];
  auto load = [&](f32x4 (&Rs)[8], const Cursor &c) {
    const int gy = 2 * c.it.oy0 - 1 + srow, gx = 2 * c.it.ox0 - 4 + 4 * sq;
    const bool ok = staged && gy >= 0 && gy < H && gx >= 0 && gx < W;   // W % 4 == 0: whole quads
    const int voff = ok ? (gy * W + gx) * 4 : kOOB;
    const rsrc_t src = c.valid && c.p < D ? make_rsrc(in + (size_t)c.it.b * in_ss, in_ss * 4) : none;
    const int soff = (c.ch * 8 * cs + c.p * HW) * 4;
#pragma unroll
    for (int k = 0; k < 8; ++k) Rs[k] = __builtin_bit_cast(f32x4, buf_load4(src, voff, soff + k * cs * 4));
  };

  f32x4 accA[NT][RB], accB[NT][RB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) accA[t][rb] = accB[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};

  Cursor q[DEPTH];   // q[d]: the unit whose loads set d holds
  q[0].item = blockIdx.x;
  q[0].it = s2_decode<Cfg>(q[0].item, total, tiles_x, tiles_y, segs, zt);
  q[0].p = q[0].it.oz0 == 0 ? 0 : 2 * q[0].it.oz0 - 1;
  q[0].ch = 0;
  q[0].valid = true;
#pragma unroll
  for (int d = 1; d < DEPTH; ++d) q[d] = advance(q[d - 1]);
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(R[d], q[d]);
  // ONE exit, behind the last set's unit: an exit edge from the middle of the body goes through the loop's latch after structurisation, and the compiler's
  // wait-count pass then merges "set d was just issued" into the head (every wait becomes vmcnt(0)).  Units past the end of the stream (at most DEPTH - 1 per
  // workgroup) load zeros and store nothing.
  for (;;) {
#pragma unroll
   for (int d = 0; d < DEPTH; ++d) {
    const Cursor c = q[d];
    const Cursor ahead = advance(q[(d + DEPTH - 1) % DEPTH]);      // the unit DEPTH ahead: set d's next load
    const Cursor nx = DEPTH == 1 ? ahead : q[(d + 1) % DEPTH];     // the unit consumed next
    const S2Item cur = c.it;
    const int p = c.p, ch = c.ch;
    f32x4 (&Rd)[8] = R[d];
    // ---- the staged unit's largest magnitude (this thread's loads -> wave -> workgroup) ----
    float m = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) m = casmvs::absmax3(casmvs::absmax3(m, Rd[c][0], Rd[c][1]), Rd[c][2], Rd[c][3]);
    const unsigned wm = casmvs::wave_max_bits(__builtin_bit_cast(unsigned, m));
    if (lane == 0) wmax[wave] = wm;
    __syncthreads();   // every wave is done with the previous unit's LDS; the four maxima are visible
    float mult, inv;
    casmvs::tile_scale(wmax, mult, inv);
    if (staged) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = Rd[c][v];
        casmvs::split_u32x4 o[2];
        casmvs::split8_f16(x, mult, o);
        u32x4 *pl = act + sunit + ((v & 1) ? ODD : 0) + (v >> 1);
        if (v != 1 || sq != 0) {   // (odd index 0 shares its unit with even index 33)
          pl[0] = o[0];
          pl[NVOX] = o[1];
        }
      }
    }
    __syncthreads();
    if (DEPTH == 1) load(Rd, ahead);   // in flight behind the matrix phase (DEPTH > 1: the next unit's loads already are; this set is refilled after it)
    // ---- matrix phase: an even plane is tap kz = 1 of A's output plane; an odd plane tap kz = 2 of A's and tap kz = 0 of B's ----
    const u32x4 *wch = wl + ch * (9 * RB * 2 * 64) + lane;
    constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 0};
    if (!(p & 1)) {
      f32x4 part[NT][RB];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) part[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        u32x4 bv[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int s = 0; s < 2; ++s) bv[t][s] = act[s * NVOX + vb[t] + ky * RS];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          u32x4 a[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) a[s] = wch[(((1 * 3 + ky) * RB + rb) * 2 + s) * 64];
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) part[t][rb] = s2_mfma(a[PA[q]], bv[t][PB[q]], part[t][rb]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // the folds stay behind the last matrix instruction (no floating-point vector work inside a matrix phase)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) accA[t][rb][r] = fmaf(part[t][rb][r], inv, accA[t][rb][r]);
    } else {
      f32x4 pa[NT][RB], pb[NT][RB];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) pa[t][rb] = pb[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        u32x4 bv[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int s = 0; s < 2; ++s) bv[t][s] = act[s * NVOX + vb[t] + ky * RS];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          u32x4 a2[2], a0[2];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            a2[s] = wch[(((2 * 3 + ky) * RB + rb) * 2 + s) * 64];
            a0[s] = wch[(((0 * 3 + ky) * RB + rb) * 2 + s) * 64];
          }
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              pa[t][rb] = s2_mfma(a2[PA[q]], bv[t][PB[q]], pa[t][rb]);
              pb[t][rb] = s2_mfma(a0[PA[q]], bv[t][PB[q]], pb[t][rb]);
            }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            accA[t][rb][r] = fmaf(pa[t][rb][r], inv, accA[t][rb][r]);
            accB[t][rb][r] = fmaf(pb[t][rb][r], inv, accB[t][rb][r]);
          }
      if (ch == NCH - 1) {
        // ---- output plane oz = (p - 1) / 2 is complete: y = lrelu(acc * scale + shift); lane holds channels 16 rb + 4 kb + r, column j ----
        const int oz = (p - 1) >> 1;
        const bool zok = c.valid && oz >= cur.oz0 && oz < Do;   // (the segment's first odd plane also feeds the previous segment's last output plane: not ours)
        const rsrc_t dst = zok ? make_rsrc(out + (size_t)cur.b * out_ss, out_ss * 4) : make_rsrc(out, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int tile = wave * NT + t, oy = cur.oy0 + (tile >> 1), ox = cur.ox0 + 16 * (tile & 1) + jcol;
          const bool ok = oy < Ho && ox < Wo;
          const int o0 = ok ? (4 * kb * ocs + (oz * Ho + oy) * Wo + ox) * 4 : kOOB;
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = fmaf(accA[t][rb][r], sc[rb][r], sh[rb][r]);
              v = v > 0.0f ? v : v * slope;
              buf_store(v, dst, o0, (rb * 16 + r) * ocs * 4);
            }
            accA[t][rb] = accB[t][rb];
            accB[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
    if (DEPTH > 1) load(Rd, ahead);   // behind the epilogue's stores: the next unit's loads stay the oldest outstanding operations
    q[d] = ahead;
    if (d == DEPTH - 1 && !nx.valid) goto done;
    if (nx.item != c.item) {   // a new item starts with empty accumulators (A holds what the last odd plane fed the next segment's first output plane)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) accA[t][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
   }
  }
done:;
}

inline uint16_t f16_bits_s2(float x) {   // round to nearest even (host)
  const _Float16 h = (_Float16)x;
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

// the number of z segments: the one that minimises (rounds of resident workgroups) x (input planes an item walks)
inline int s2_pick_segments(int Do, long patches, int resident) {
  int best = 1;
  long best_cost = -1;
  for (int s = 1; s <= Do; ++s) {
    const int zt = casmvs::ceil_div(Do, s), real = casmvs::ceil_div(Do, zt);
    if (real != s) continue;
    const long total = patches * real, rounds = (total + resident - 1) / resident;
    const long cost = rounds * (2 * zt + 1);
    if (best_cost < 0 || cost < best_cost) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}

#ifdef HIPEMU_LDS_BYTES
int g_s2_emu_depth = 2;
#endif

template <int CIN, int COUT>
int launch_s2(const void *packed, const float *in, float *out, int B, int D, int H, int W, float slope, hipStream_t st) {
  using Cfg = S2Cfg<CIN, COUT>;
  const int Do = (D - 1) / 2 + 1, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const int tiles_x = casmvs::ceil_div(Wo, Cfg::TX), tiles_y = casmvs::ceil_div(Ho, Cfg::TY);
#ifdef HIPEMU_LDS_BYTES   // tests/hipemu runs every depth
  auto kernel = g_s2_emu_depth == 1 ? conv_s2_sf_kernel<CIN, COUT, 1> : g_s2_emu_depth == 2 ? conv_s2_sf_kernel<CIN, COUT, 2> : conv_s2_sf_kernel<CIN, COUT, 3>;
#else
  auto kernel = conv_s2_sf_kernel<CIN, COUT, (CIN == 8 ? CASMVS_S2_DEPTH_CONV1 : CASMVS_S2_DEPTH_CONV3)>;
#endif
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), Cfg::LDS_BYTES, "conv_s2_sf_kernel")) return rc;
  const int resident = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), Cfg::THREADS, Cfg::LDS_BYTES);
  const int segs = s2_pick_segments(Do, (long)tiles_x * tiles_y * B, resident), zt = casmvs::ceil_div(Do, segs);
  const long total = (long)tiles_x * tiles_y * segs * B;
  CASMVS_REQUIRE(total < (1L << 31), "conv_s2_splitf16_forward: too many patches");
  hipLaunchKernelGGL(kernel, dim3((unsigned)(total < resident ? total : resident)), dim3(Cfg::THREADS), Cfg::LDS_BYTES, st, in,
                     reinterpret_cast<const unsigned char *>(packed), out, B, D, H, W, tiles_x, tiles_y, segs, zt, slope);
  return casmvs::check_launch("conv_s2_sf_kernel");
}

template <int CIN, int COUT>
size_t s2_packed_bytes() { return S2Cfg<CIN, COUT>::W_BYTES + 2 * COUT * sizeof(float); }

}  // namespace

extern "C" size_t casmvs_conv_s2_splitf16_packed_bytes(int cin, int cout) {
  if (cin == 8 && cout == 16) return s2_packed_bytes<8, 16>();
  if (cin == 16 && cout == 32) return s2_packed_bytes<16, 32>();
  return 0;
}

// HOST-side packing: weight (cout, cin, 3, 3, 3) float32 -> w' = 2^kw w (max |w'| in [2^13, 2^14)); per chunk of 8 input channels, (kz, ky), block of 16
// output channels, slice (f16(w'), f16(w' - f16(w'))), per lane the 8 float16 values
//   A[i = lane & 15][k = 8 (lane >> 4) + e] = slice(w'[co = 16 rb + i][ci = 8 chunk + e][kz][ky][kx = lane >> 4])   (kx = 3: zeros);
// then scale[cout] * 2^-kw, shift[cout].
extern "C" int casmvs_conv_s2_splitf16_pack(int cin, int cout, const float *weight, const float *scale, const float *shift, void *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "conv_s2_splitf16_pack: null pointer");
  CASMVS_REQUIRE(casmvs_conv_s2_splitf16_packed_bytes(cin, cout) != 0, "conv_s2_splitf16_pack: %d -> %d (8 -> 16 or 16 -> 32)", cin, cout);
  float wmax = 0.0f;
  for (int i = 0; i < cout * cin * 27; ++i) {
    CASMVS_REQUIRE(std::isfinite(weight[i]), "conv_s2_splitf16_pack: weight %d is not finite", i);
    wmax = std::fmax(wmax, std::fabs(weight[i]));
  }
  int ex = 14;
  if (wmax > 0.0f) (void)std::frexp(wmax, &ex);
  const int kw = 14 - ex;
  const int nch = cin / 8, nrb = cout / 16;
  uint16_t *p = reinterpret_cast<uint16_t *>(packed);
  for (int chunk = 0; chunk < nch; ++chunk)
    for (int r9 = 0; r9 < 9; ++r9)
      for (int rb = 0; rb < nrb; ++rb) {
        uint16_t img[2][64][8];
        for (int l = 0; l < 64; ++l) {
          const int co = 16 * rb + (l & 15), kx = l >> 4;
          for (int e = 0; e < 8; ++e) {
            const int ci = 8 * chunk + e;
            const float w = kx < 3 ? std::ldexp(weight[(((size_t)co * cin + ci) * 9 + r9) * 3 + kx], kw) : 0.0f;
            const float a = (float)(_Float16)w;
            img[0][l][e] = f16_bits_s2(w);
            img[1][l][e] = f16_bits_s2(w - a);
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

extern "C" int casmvs_conv_s2_splitf16_supported(int cin, int cout, int W) {
  return casmvs_conv_s2_splitf16_packed_bytes(cin, cout) != 0 && W % 4 == 0 && W >= 4;
}

extern "C" int casmvs_conv_s2_splitf16_forward_f32(const void *packed, const float *in, float *out, int B, int cin, int cout, int D, int H, int W, float slope,
                                                   void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "conv_s2_splitf16_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && D > 0 && H > 0 && casmvs_conv_s2_splitf16_supported(cin, cout, W),
                 "conv_s2_splitf16_forward: B=%d %d -> %d D=%d H=%d W=%d (8 -> 16 or 16 -> 32, W a multiple of 4)", B, cin, cout, D, H, W);
  CASMVS_REQUIRE(((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(packed)) & 15) == 0 && (reinterpret_cast<size_t>(out) & 3) == 0,
                 "conv_s2_splitf16_forward: 16-byte aligned input and image");
  CASMVS_REQUIRE((size_t)cin * D * H * W < ((size_t)1 << 29), "conv_s2_splitf16_forward: one sample's input tensor must hold < 2^29 floats");
  if (cin == 8) return launch_s2<8, 16>(packed, in, out, B, D, H, W, slope, (hipStream_t)stream);
  return launch_s2<16, 32>(packed, in, out, B, D, H, W, slope, (hipStream_t)stream);
}
