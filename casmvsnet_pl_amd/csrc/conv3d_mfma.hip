// CostRegNet 3D convolutions on the gfx950 matrix cores, fp32 in / fp32 accumulate.
//
// Reference semantics: models/mvsnet.py:60-104 (CostRegNet), models/modules.py:21-31
// (ConvBnReLU3D): Conv3d / ConvTranspose3d (k3, p1) -> eval-mode ABN -> (+ skip).
//
// Instruction choice (measured on MI355X with casmvs_selftest_mfma_rate): the
// fp32 MFMA that runs at the full 64 FLOP/clk/SIMD is v_mfma_f32_16x16x4_f32 (32 cycles, 149-155
// TFLOP/s chip-wide); the 16-block v_mfma_f32_4x4x1_16b_f32 form reaches 135 and is not used by any
// kernel (the 1-channel `prob` head is a VALU kernel); it survives only in the rate probes.
//
// Formulation.  D[16 rows][16 cols] += A[16][4] * B[4][16] per instruction with
//   cols = 16 output voxels that are consecutive along x (one "column tile"),
//   rows = 16 output features, K = 4 contraction steps.  Two row/K assignments are used:
//   CI  (Cout % 16 == 0): rows = 16 output channels, K = 4 input channels of one tap.
//   PX  (Cout == 8, the full-resolution layers that hold most FLOPs): rows = (co, s) = 8 output
//       channels x 2 x-phases - the column tile covers 32 consecutive x as (x0 + 2j + s) - and
//       K = 4 input x-offsets u of one (ci, kz, ky): out[x0+2j+s] += in[x0+2j+u-1] * w[kx = u-s].
//       3 of the 4 K steps are non-zero for every row: 75 % of the matrix pipe does useful work
//       (padding Cout 8 -> 16 would give 50 %).
//   The transposed convolutions use the same two forms with rows = (co, x-parity) for Cout = 8.
// Lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15] and receives
// D[row = 4 * (l >> 4) + reg][col = l & 15] (reg = 0..3).
//
// Data flow per workgroup (256 threads = 4 wavefronts, NT column tiles per wavefront):
//   for each chunk of CK input channels: the zero-padded halo tile of the chunk and the chunk's
//   weight images are staged in LDS (global -> registers one chunk ahead -> LDS), then per
//   (tap | row-tap) iteration: NA A-images (ds_read_b32, lane-linear) and NA*NT B operands
//   (ds_read_b32, bank-conflict-free by construction of the channel stride) feed NA*NT MFMAs.
//
// Packed parameter image (built on the host by casmvs_conv3d_pack_f32):
//   [slice][chunk][NW floats of 64-lane A images]  scale[slices*coutb]  shift[slices*coutb]
//   zero[64] (target of out-of-range staging loads).  Image order inside a chunk, and the
//   lane -> (row, k) -> weight mapping, are documented at each format in pack_weight().
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "buffer_ops.h"
#include "common.h"

namespace {

#ifndef CASMVS_MFMA_DRAIN_NOPS
#define CASMVS_MFMA_DRAIN_NOPS 0
#endif
using namespace casmvs::buf;  // rsrc_t, kOOB, make_rsrc, buf_load*, buf_store*, xcd_major, f32x2, f32x4v, u32x2, u32x4
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 4x4x1 with A-block broadcast: acc[r] (lane) += a[4*ABID + r] * b[lane]   (lane-mapping probe only)
template <int ABID>
__device__ __forceinline__ f32x4 mfma_bcast(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, /*cbsz=*/4, /*abid=*/ABID, /*blgp=*/0);
}

// ---- layer formats -------------------------------------------------------------------------------
enum Fmt { FMT_P1 = 0, FMT_CI = 1, FMT_PX = 2, FMT_TCI = 3, FMT_TPX = 4 };

struct LayerCfg {
  int fmt;
  int coutb;        // output channels per slice: 16 (CI, TCI), 8 (PX, TPX), 4 (P1: 1 used)
  int slices;       // ceil(cout / coutb), blockIdx.z
  int units;        // contraction units per slice (padded so that any kernel chunking stays in range)
  int unit_floats;  // floats per unit
  int kz, ks;       // kernel extent along z and along y / x (3,3 for the 3D layers; 1,{1,3,5} for the 2D ones)
  int taps() const { return kz * ks * ks; }
  size_t per_slice() const { return (size_t)units * unit_floats; }
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// The weight image of a slice is a sequence of contraction UNITS, so that a kernel may chunk the
// input channels by any CK it likes (chunk s = units [s*CK/q, (s+1)*CK/q)):
//   CI / TCI : unit = 4 input channels (one "ci quad"), kz*ks*ks tap images -> [quad][tap][64]
//   PX       : unit = 1 input channel, kz*ks (kz,ky) images                 -> [ci][kz*3+ky][64]
//   TPX      : unit = 4 input channels, 9 (kz,ky) x 2 (dx) images           -> [quad][kz*3+ky][dx][64]
//   P1       : unit = 2 input channels, [tap (27 + 5 zeros)][channel of the pair]: no lane images - the 1-channel
//              `prob` head is a VALU kernel that reads its weights as wave-uniform scalars, a pair per packed FMA
// The 2D layers of FeatureNet (kinds CASMVS_CONV2D_*) are the same formats with kz = 1.
inline bool layer_cfg(int kind, int cin, int cout, LayerCfg &c) {
  if (cin < 1 || cout < 1) return false;
  c.kz = 3;
  c.ks = 3;
  const int quads = round_up((cin + 3) / 4, 4);
  if (kind == CASMVS_CONV_S1) {
    if (cout == 1) { c.fmt = FMT_P1; c.coutb = 4; c.units = round_up(cin, 8) / 2; c.unit_floats = 64; }
    else if (cout == 8) { c.fmt = FMT_PX; c.coutb = 8; c.units = round_up(cin, 8); c.unit_floats = 9 * 64; }
    else if (cout % 16 == 0) { c.fmt = FMT_CI; c.coutb = 16; c.units = quads; c.unit_floats = 27 * 64; }
    else return false;
  } else if (kind == CASMVS_CONV_S2) {
    if (cout % 16 != 0) return false;
    c.fmt = FMT_CI; c.coutb = 16; c.units = quads; c.unit_floats = 27 * 64;
  } else if (kind == CASMVS_CONV_T2) {
    if (cout == 8) { c.fmt = FMT_TPX; c.coutb = 8; c.units = quads; c.unit_floats = 18 * 64; }
    else if (cout % 16 == 0) { c.fmt = FMT_TCI; c.coutb = 16; c.units = quads; c.unit_floats = 27 * 64; }
    else return false;
  } else if (kind == CASMVS_CONV2D_K3) {
    c.kz = 1;
    if (cout == 8) { c.fmt = FMT_PX; c.coutb = 8; c.units = round_up(cin, 8); c.unit_floats = 3 * 64; }
    else if (cout % 16 == 0) { c.fmt = FMT_CI; c.coutb = 16; c.units = quads; c.unit_floats = 9 * 64; }
    else return false;
  } else if (kind == CASMVS_CONV2D_K5S2) {
    if (cout % 16 != 0) return false;
    c.kz = 1; c.ks = 5; c.fmt = FMT_CI; c.coutb = 16; c.units = quads; c.unit_floats = 25 * 64;
  } else if (kind == CASMVS_CONV2D_K1 || kind == CASMVS_CONV2D_K1_UP) {
    if (cout % 16 != 0) return false;
    c.kz = 1; c.ks = 1; c.fmt = FMT_CI; c.coutb = 16; c.units = quads; c.unit_floats = 64;
  } else {
    return false;
  }
  c.slices = (cout + c.coutb - 1) / c.coutb;
  return true;
}

// Value of lane `l` of image `img` of unit `unit`, slice `sl` (host side).
inline float pack_weight(const LayerCfg &c, int kind, int cin, int cout, const float *w, int sl,
                         int unit, int img, int l) {
  auto conv_w = [&](int co, int ci, int tap) -> float {
    if (co >= cout || ci >= cin || tap < 0) return 0.0f;
    const int nt = c.taps();
    return kind == CASMVS_CONV_T2 ? w[((size_t)ci * cout + co) * nt + tap]   // (cin, cout, 3,3,3)
                                  : w[((size_t)co * cin + ci) * nt + tap];  // (cout, cin, [kz,] ks, ks)
  };
  const int i = l & 15, k = l >> 4;
  switch (c.fmt) {
    case FMT_P1:     // one 64-float row per PAIR of input channels: l = 2 * tap + (channel & 1)
      return (l >> 1) < 27 ? conv_w(0, 2 * unit + (l & 1), l >> 1) : 0.0f;
    case FMT_CI:     // img = tap; row i = co, k = input channel of the quad
    case FMT_TCI:
      return conv_w(sl * 16 + i, unit * 4 + k, img);
    case FMT_PX: {   // img = kz*3+ky; row i = (co, s), k = x-offset u, kx = u - s
      const int co = i >> 1, s = i & 1, kx = k - s;
      return (kx >= 0 && kx <= 2) ? conv_w(co, unit, img * 3 + kx) : 0.0f;
    }
    case FMT_TPX: {  // img = (kz*3+ky) * 2 + dx; row i = (co, px), k = input channel of the quad
      const int dx = img % 2, r9 = img / 2;
      const int co = i >> 1, px = i & 1;
      // even output x = 2m takes (kx = 1, cell m); odd x = 2m+1 takes (kx = 2, m), (kx = 0, m+1)
      const int kx = dx == 0 ? (px == 0 ? 1 : 2) : (px == 1 ? 0 : -1);
      return kx >= 0 ? conv_w(co, unit * 4 + k, r9 * 3 + kx) : 0.0f;
    }
  }
  return 0.0f;
}

// ---- staging: global -> registers -> LDS, software-pipelined one chunk ahead -------------------
// A chunk = CK input channels of the zero-padded halo tile (CK*IZ planes of IY*IX floats) plus
// the chunk's NW weight floats.  A thread copies the same NPASS in-plane positions of every
// plane, so the (iy, ix) decode, the bounds tests and the in-plane global offset are computed
// ONCE per kernel (StagePlan); a plane then costs one add + load per position.  The loads of
// chunk s+1 are issued right after the barrier that publishes chunk s and land in registers
// while the MFMA loop of chunk s runs; they are written to LDS after the next barrier.  Weights
// go through LDS too, so that the MFMA loop contains no vector-memory instruction (an in-loop
// global load would make the compiler's in-order vmcnt wait drain the whole prefetch).

// Stager<VEC = 1>: one float per load.  A thread copies the same NPASS in-plane positions of
// every plane; per-lane byte offsets are tile constants, the plane offset is scalar.
// Stager<VEC = 4>: 16-byte loads / ds_write_b128 (needs Wi % 4 == 0 and 4-aligned tile origins; rows
// are widened to aligned 16-byte groups).  The chunk's float4 groups are flattened over the 256
// threads (group e = tid + 256 k), so a chunk costs ceil(groups / 256) vector-memory instructions
// per thread - 4x fewer than VEC = 1, which is what bounds the staging phase: the texture
// addresser retires ~1 wave-instruction per 16 cycles regardless of its width (measured).
template <int VEC, int CK, int IZ, int IY, int IXR, int SC, int NW>
struct Stager;

template <int CK, int IZ, int IY, int IXR, int SC, int NW>
struct Stager<1, CK, IZ, IY, IXR, SC, NW> {
  static constexpr int PLANE = IY * IXR, NPL = CK * IZ;
  static constexpr int NPASS = (PLANE + kThreads - 1) / kThreads;
  static constexpr int NWR = (NW + kThreads - 1) / kThreads;
  int in_cs, HiWi, Di, iz0;
  int voff[NPASS];   // byte offset (gy * Wi + gx) * 4 of this thread's position, or kOOB
  float v[NPL][NPASS];
  float w[NWR];

  __device__ __forceinline__ void init_kernel(int in_cs_, int HiWi_, int Di_) {
    in_cs = in_cs_;
    HiWi = HiWi_;
    Di = Di_;
  }
  __device__ __forceinline__ void init_tile(int iz0_, int iy0, int ix0, int Hi, int Wi) {
    iz0 = iz0_;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int pe = threadIdx.x + p * kThreads;
      const int iy = pe / IXR, ix = pe - iy * IXR;
      const int gy = iy0 + iy, gx = ix0 + ix;
      const bool inb = pe < PLANE && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
      voff[p] = inb ? (gy * Wi + gx) * 4 : kOOB;
    }
  }
  // issue every load of the chunk that starts at input channel ci0; nothing here waits
  __device__ __forceinline__ void load(rsrc_t src, int cin, int ci0, const float *__restrict__ wchunk) {
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int e = threadIdx.x + i * kThreads;
      w[i] = wchunk[e < NW ? e : 0];
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const int cil = pl / IZ, iz = pl - cil * IZ;
      const int ci = ci0 + cil, gz = iz0 + iz;
      const bool plane_ok = ci < cin && gz >= 0 && gz < Di;  // wave-uniform
      const int soff = plane_ok ? (ci * in_cs + gz * HiWi) * 4 : 0;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) v[pl][p] = buf_load(src, plane_ok ? voff[p] : kOOB, soff);
    }
  }
  // tile layout: [cil][iz][iy][ix] with channel stride SC (>= IZ*PLANE, padded for the banks)
  __device__ __forceinline__ void store(float *tile, float *wts) const {
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int e = threadIdx.x + i * kThreads;
      if (e < NW) wts[e] = w[i];
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const int cil = pl / IZ, iz = pl - cil * IZ;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int pe = threadIdx.x + p * kThreads;
        if (pe < PLANE) tile[cil * SC + iz * PLANE + pe] = v[pl][p];
      }
    }
  }
};

template <int CK, int IZ, int IY, int IXR, int SC, int NW>
struct Stager<4, CK, IZ, IY, IXR, SC, NW> {
  static_assert(IXR % 4 == 0 && SC % 4 == 0 && NW % 4 == 0, "16-byte groups");
  static constexpr int ROWV = IXR / 4, PLV = IY * ROWV, TOTV = CK * IZ * PLV;
  static constexpr int NK = (TOTV + kThreads - 1) / kThreads;
  static constexpr int NWV = NW / 4, NWR = (NWV + kThreads - 1) / kThreads;
  int in_cs, HiWi, Di;
  int voff[NK];      // byte offset of group k inside the sample for chunk channel 0, or kOOB (tile constant)
  f32x4v v[NK];
  f32x4v w[NWR];

  // group e = tid + 256 k -> (cil, iz, iy, xv); compile-time divisors, recomputed where needed
  // instead of being kept in registers
  struct Pos {
    int cil, iz, iy, xv;
    bool valid;
  };
  static __device__ __forceinline__ Pos pos_of(int k) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // opaque: keeps the compiler from hoisting 4 * NK decoded ints into registers
    const int e = tid + k * kThreads;
    const int pl = e / PLV, r = e - pl * PLV;
    Pos p;
    p.cil = pl / IZ;
    p.iz = pl - p.cil * IZ;
    p.iy = r / ROWV;
    p.xv = r - p.iy * ROWV;
    p.valid = e < TOTV;
    return p;
  }

  __device__ __forceinline__ void init_kernel(int in_cs_, int HiWi_, int Di_) {
    in_cs = in_cs_;
    HiWi = HiWi_;
    Di = Di_;
  }
  __device__ __forceinline__ void init_tile(int iz0, int iy0, int ix0, int Hi, int Wi) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const Pos p = pos_of(k);
      const int gz = iz0 + p.iz, gy = iy0 + p.iy, gx = ix0 + 4 * p.xv;
      const bool inb = p.valid && gz >= 0 && gz < Di && gy >= 0 && gy < Hi && gx >= 0 && gx + 3 < Wi;
      voff[k] = inb ? (p.cil * in_cs + gz * HiWi + gy * Wi + gx) * 4 : kOOB;
    }
  }
  __device__ __forceinline__ void load(rsrc_t src, int cin, int ci0, const float *__restrict__ wchunk) {
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int e = threadIdx.x + i * kThreads;
      w[i] = *reinterpret_cast<const f32x4v *>(wchunk + 4 * (e < NWV ? e : 0));
    }
    const int soff = ci0 * in_cs * 4;
    const bool full = ci0 + CK <= cin;  // wave-uniform: only the last chunk can be channel-padded
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const bool ok = full || ci0 + pos_of(k).cil < cin;
      v[k] = buf_load4(src, ok ? voff[k] : kOOB, soff);
    }
  }
  __device__ __forceinline__ void store(float *tile, float *wts) const {
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int e = threadIdx.x + i * kThreads;
      if (e < NWV) *reinterpret_cast<f32x4v *>(wts + 4 * e) = w[i];
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const Pos p = pos_of(k);
      if (p.valid) *reinterpret_cast<f32x4v *>(tile + p.cil * SC + p.iz * (IY * IXR) + p.iy * IXR + 4 * p.xv) = v[k];
    }
  }
  // Single-operation forms, used by the double-buffered kernel to spread the staging work over the
  // issue slots of the MFMA loop: op j in [0, NOPS) is one 16-byte load (resp. one ds_write_b128).
  static constexpr int NOPS = NK + NWR;
  template <int J>
  __device__ __forceinline__ void load_op(rsrc_t src, int cin, int ci0, const float *__restrict__ wchunk) {
    if constexpr (J < NWR) {
      const int e = threadIdx.x + J * kThreads;
      w[J] = *reinterpret_cast<const f32x4v *>(wchunk + 4 * (e < NWV ? e : 0));
    } else {
      constexpr int k = J - NWR;
      const bool ok = ci0 + CK <= cin || ci0 + pos_of(k).cil < cin;
      v[k] = buf_load4(src, ok ? voff[k] : kOOB, ci0 * in_cs * 4);
    }
  }
  template <int J>
  __device__ __forceinline__ void store_op(float *tile, float *wts) const {
    if constexpr (J < NWR) {
      const int e = threadIdx.x + J * kThreads;
      if (e < NWV) *reinterpret_cast<f32x4v *>(wts + 4 * e) = w[J];
    } else {
      constexpr int k = J - NWR;
      const Pos p = pos_of(k);
      if (p.valid) *reinterpret_cast<f32x4v *>(tile + p.cil * SC + p.iz * (IY * IXR) + p.iy * IXR + 4 * p.xv) = v[k];
    }
  }
};

// Stager<VEC = 5>: the 16-byte loads of Stager<4>, but every row is stored DE-INTERLEAVED in x -
// [even columns | odd columns], IXR / 2 each - for the stride-2 layers: tap kx of output column j
// reads input column 2 j + kx - 1, i.e. a unit-stride walk through one parity half (bank-conflict free
// like the stride-1 layers; the interleaved row gives 4-way conflicts once the channel stride is a
// multiple of 4, which 16-byte LDS writes need).
template <int CK, int IZ, int IY, int IXR, int SC, int NW>
struct Stager<5, CK, IZ, IY, IXR, SC, NW> : Stager<4, CK, IZ, IY, IXR, SC, NW> {
  using Base = Stager<4, CK, IZ, IY, IXR, SC, NW>;
  static_assert(IXR % 8 == 0 || (IXR / 2) % 2 == 0, "8-byte aligned halves");
  __device__ __forceinline__ void store(float *tile, float *wts) const {
#pragma unroll
    for (int i = 0; i < Base::NWR; ++i) {
      const int e = threadIdx.x + i * kThreads;
      if (e < Base::NWV) *reinterpret_cast<f32x4v *>(wts + 4 * e) = this->w[i];
    }
#pragma unroll
    for (int k = 0; k < Base::NK; ++k) {
      const typename Base::Pos p = Base::pos_of(k);
      if (p.valid) {
        float *row = tile + p.cil * SC + p.iz * (IY * IXR) + p.iy * IXR + 2 * p.xv;
        *reinterpret_cast<f32x2 *>(row) = f32x2{this->v[k][0], this->v[k][2]};            // columns 4 xv, 4 xv + 2
        *reinterpret_cast<f32x2 *>(row + IXR / 2) = f32x2{this->v[k][1], this->v[k][3]};  // columns 4 xv + 1, + 3
      }
    }
  }
};

constexpr int round_up_to_16_mod_32(int x) { return x + ((16 - x % 32) + 32) % 32; }

// ---- Conv3d k3 p1 (stride 1 or 2) on 16x16x4 ----------------------------------------------------
template <int MODE, int STRIDE, int CK, int NT, int TZ, int TY, int TX, int VEC, int KZ = 3, int KS = 3>
struct Conv16Cfg {
  static_assert(MODE == FMT_CI || MODE == FMT_PX, "conv16: CI or PX");
  static_assert(VEC == 1 || (VEC == 4 && KS == 3 && (STRIDE == 1 || MODE == FMT_CI)), "16-byte staging: k3 layers only");
  static constexpr bool DEINT = VEC == 4 && STRIDE == 2;  // rows stored [even x | odd x] (Stager<5>)
  static_assert((KZ == 3 || (KZ == 1 && TZ == 1)) && (KS == 1 || KS == 3 || KS == 5), "kernel extents");
  static_assert(MODE != FMT_PX || KS == 3, "PX form: 3 taps along x");
  static constexpr int PZ = KZ / 2, PS = KS / 2;       // "same" padding
  static constexpr int XW = MODE == FMT_PX ? 32 : 16;  // output voxels along x per column tile
  static constexpr int NXG = TX / XW;
  static_assert(TX % XW == 0 && TZ * TY * NXG == 4 * NT, "tile = 4 waves x NT column tiles");
  static constexpr int IZ = STRIDE * (TZ - 1) + KZ, IY = STRIDE * (TY - 1) + KS;
  // staged row: VEC 1: x in [S x0 - PS, S (x0 + TX - 1) + PS]; VEC 4: widened to the aligned [x0 - 4, x0 + TX + 4)
  static constexpr int IX = VEC == 4 ? (STRIDE == 1 ? TX + 8 : (4 + 2 * (TX - 1) + 2 + 3) / 4 * 4) : STRIDE * (TX - 1) + KS;
  static constexpr int XLO = VEC == 4 ? 4 : PS;   // tile row starts at global x = S * x0 - XLO
  static constexpr int XOFF = XLO - PS;           // local x of (output x0, tap kx = 0)
  static constexpr int SY = IX, SZ = IY * IX;
  // channel stride: B lanes k = 0..3 read 4 channels (CI) -> k * SC must land on disjoint banks:
  // stride 1: 16 consecutive words per k -> SC == 16 (mod 32); stride 2: even words -> SC odd.
  static constexpr int SC = MODE == FMT_PX ? IZ * SZ
                            : ((STRIDE == 1 || DEINT) ? round_up_to_16_mod_32(IZ * SZ) : (IZ * SZ) | 1);
  static constexpr int NA = MODE == FMT_PX ? CK : CK / 4;              // A images per iteration
  static constexpr int NITER = MODE == FMT_PX ? KZ * KS : KZ * KS * KS;  // (kz,ky) | (kz,ky,kx)
  static constexpr int ASTEP = MODE == FMT_PX ? SC : 4 * SC;           // B offset between A images
  static constexpr int NW = NITER * NA * 64;
  static constexpr size_t LDS_BYTES = (size_t)(CK * SC + NW) * sizeof(float);
};

#ifdef CASMVS_TRACE
// Profiling build only (tools/gpu_trace.sh): wave 0 of the every 16th workgroup (of the first 1024) of conv16_kernel
// stamps the shader clock at phase boundaries into a device buffer read back by casmvs_trace_read.
__device__ unsigned long long g_trace[64 * 128];
#define TRACE_STAMP()                                                                 \
  do {                                                                                \
    if (threadIdx.x == 0 && (blockIdx.x & 15) == 0 && blockIdx.x < 1024 && tr_n < 128) \
      g_trace[(blockIdx.x >> 4) * 128 + tr_n++] = __builtin_readcyclecounter();       \
  } while (0)
#else
#define TRACE_STAMP() do {} while (0)
#endif

// Work item (= one output tile of one slice of one sample) of the v-th (block, iteration) pair,
// v = blockIdx.x + n * gridDim.x < total.  Workgroup b runs on XCD b % 8 and each XCD has its own
// 4 MiB L2, so the items are dealt XCD-major: XCD x owns the contiguous range [start(x), start(x+1)) and
// the workgroups resident on it at one time walk neighbouring tiles (z fastest, then x, then y:
// ~80-100 concurrent tiles = one z-x slab, whose halos are then shared inside that L2).  With the
// plain v -> tile map neighbouring tiles ran on 8 different XCDs and every L2 fetched every halo:
// PMC FETCH_SIZE of conv0 was 5x the algorithmic input bytes (profiles/r01_pmc_traffic.md).
#ifndef CASMVS_DB_ORDER
#define CASMVS_DB_ORDER 0   // A/B builds: 1 = x-fastest tile order, 2 = x fastest + CU pairing (buffer_ops.h: cu_pair_remap)
#endif
struct TileCoord {
  int tx0, ty0, tz0, b, slice;
};
// Tile order inside an XCD's contiguous item range: z fastest, then x, then y.  (Tried in round 2: NB = 4 consecutive
// tile rows in y before x - a 16-row x 128-px block per XCD instead of a full-width 4-row slab.  PMC FETCH_SIZE of
// conv0 went UP, 561 -> 753 MB per launch, no change in time: neither working set fits the 4 MiB L2 and the slab's
// streaming order along x re-uses the z-halo planes better.  Reverted.)
template <int TZ, int TY, int TX>
__device__ __forceinline__ TileCoord decode_tile(int v, int total, int tiles_x, int tiles_y, int tiles_z, int B) {
#if CASMVS_DB_ORDER == 2
  v = cu_pair_remap(v, total);
#endif
  int item = xcd_major(v, total);
  TileCoord c;
#if CASMVS_DB_ORDER >= 1   // A/B builds: x fastest, then z, then y (conv0_splitf16.hip: +2.6 %)
  c.tx0 = (item % tiles_x) * TX;
  item /= tiles_x;
  c.tz0 = (item % tiles_z) * TZ;
  item /= tiles_z;
#else
  c.tz0 = (item % tiles_z) * TZ;
  item /= tiles_z;
  c.tx0 = (item % tiles_x) * TX;
  item /= tiles_x;
#endif
  c.ty0 = (item % tiles_y) * TY;
  item /= tiles_y;
  c.b = item % B;
  c.slice = item / B;
  return c;
}

// Persistent workgroups: the grid is sized to the number of resident workgroups and each one walks
// tiles item, item + gridDim.x, ...  The first chunk of the NEXT tile is prefetched during the last
// chunk of the current one, so the global-load latency, the address set-up and the epilogue
// stores of a tile all overlap MFMA work - measured per-workgroup fixed cost before: ~13 us.
// ABL (ablation, profiling only): 0 = normal, 1 = staging only (no MFMA loop), 2 = MFMA loop only.
// KZ / KS: kernel extent along z and along y, x ("same" padding).  KZ = 1 (with TZ = 1, D = 1) turns
// the kernel into the 2D convolutions of FeatureNet; UPS = 1 makes the epilogue add the bilinear x2
// upsampling (align_corners = True) of `skip` (B, Cout, Ho/2, Wo/2) instead of `skip` itself - the
// FPN top-down step F.interpolate(coarse) + lateral(x) of mvsnet.py:36-38,53-54.
// OUT2 = 1 (2D layers, no skip input): `skip` is instead a SECOND output that receives the same result
// pixel-major, (B, Ho, Wo, Cout) - the layout the cost-volume gather wants - next to the NCHW one.
template <int MODE, int STRIDE, int CK, int NT, int TZ, int TY, int TX, int VEC, int ABL = 0, int KZ = 3,
          int KS = 3, int UPS = 0, int OUT2 = 0>
__global__ __launch_bounds__(kThreads, 3) void conv16_kernel(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ skip,
    float *__restrict__ out, int B, int cin, int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
    int per_slice, int slices, int tiles_x, int tiles_y, int tiles_z, float slope) {
  using Cfg = Conv16Cfg<MODE, STRIDE, CK, NT, TZ, TY, TX, VEC, KZ, KS>;
  constexpr int PZ = Cfg::PZ, PS = Cfg::PS;
  const int nstages = (cin + CK - 1) / CK;
  constexpr int IZ = Cfg::IZ, IY = Cfg::IY, IX = Cfg::IX, SY = Cfg::SY, SZ = Cfg::SZ, SC = Cfg::SC;
  constexpr int XLO = Cfg::XLO, XOFF = Cfg::XOFF;
  constexpr int NA = Cfg::NA, NITER = Cfg::NITER, ASTEP = Cfg::ASTEP, NW = Cfg::NW, NXG = Cfg::NXG;
  constexpr int COUTB = MODE == FMT_PX ? 8 : 16;
  extern __shared__ float smem[];
  float *tile = smem;            // [CK][IZ][IY][IX], channel stride SC
  float *wts = smem + CK * SC;   // [NA][NITER][64]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jcol = lane & 15, kq = lane >> 4;
  const int total = tiles_x * tiles_y * tiles_z * B * slices;
  int item = blockIdx.x;
  if (item >= total) return;

  // column tile t of this wave -> (cz, cy, cx) inside the block tile and the lane's LDS base
  int base[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ct = wave * NT + t;
    const int cx = ct % NXG, cy = (ct / NXG) % TY, cz = ct / (NXG * TY);
    if (MODE == FMT_PX) base[t] = cz * SZ + cy * SY + cx * 32 + 2 * jcol + kq + XOFF;                   // k = x-offset u
    else if (Cfg::DEINT) base[t] = kq * SC + (cz * 2) * SZ + (cy * 2) * SY + cx * 16 + jcol;             // parity halves: unit stride
    else base[t] = kq * SC + (cz * STRIDE) * SZ + (cy * STRIDE) * SY + (cx * 16 + jcol) * STRIDE + XOFF;  // k = channel
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int in_cs = Di * Hi * Wi;    // input / output channel strides (floats); one sample of
  const int out_cs = Do * Ho * Wo;   // either tensor is < 2^29 floats (checked on the host)
  const size_t in_ss = (size_t)cin * in_cs, out_ss = (size_t)cout * out_cs;  // sample strides
  const float *tail = wpk + (size_t)slices * per_slice;     // scale | shift | zero

  // MFMA loop of one chunk, written out in issue order and pinned with sched_barrier.  Step
  // s = (a, t) of an iteration needs one B operand (ds_read_b32, immediate offset a * ASTEP) and
  // feeds one MFMA.  The read of step s + P is issued right before the MFMA of step s (wrapping
  // into the next iteration), so P * 32 MFMA cycles of LDS latency are always covered and the
  // waitcnt pass emits counted lgkmcnt waits.  The A images of the next iteration are read a
  // whole iteration ahead.
  constexpr int NS = NA * NT;              // steps per iteration
  constexpr int P0 = NS % 8 == 0 ? 8 : (NS % 4 == 0 ? 4 : NS);  // read-ahead distance (steps)
  constexpr int P = (ABL == 16 || ABL == 32) ? (NS % ABL == 0 ? ABL : P0) : P0;  // profiling override
  static_assert(NS % P == 0, "ring slots must line up across iterations");
  auto it_off = [&](int it) -> int {
    if (Cfg::DEINT) {
      // row starts at input column 2 x0 - 4: tap kx of output j reads column 2 j + kx + 3 of the row:
      // kx = 1 -> even half, index j + 2; kx = 0 / 2 -> odd half, index j + 1 / j + 2
      const int kx = it % 3;
      return (it / 9) * SZ + ((it / 3) % 3) * SY + (kx == 1 ? 2 : IX / 2 + (kx == 0 ? 1 : 2));
    }
    return MODE == FMT_PX ? (it / KS) * SZ + (it % KS) * SY : (it / (KS * KS)) * SZ + ((it / KS) % KS) * SY + (it % KS);
  };

#ifdef CASMVS_TRACE
  int tr_n = 0;
#endif
  TRACE_STAMP();  // kernel start
  TileCoord cur = decode_tile<TZ, TY, TX>(item, total, tiles_x, tiles_y, tiles_z, B);
  Stager<Cfg::DEINT ? 5 : VEC, CK, IZ, IY, IX, SC, NW> regs;
  regs.init_kernel(in_cs, Hi * Wi, Di);
  regs.init_tile(cur.tz0 * STRIDE - PZ, cur.ty0 * STRIDE - PS, cur.tx0 * STRIDE - XLO, Hi, Wi);
  regs.load(make_rsrc(in + cur.b * in_ss, in_ss * 4), cin, 0, wpk + (size_t)cur.slice * per_slice);
  for (;;) {
    const int next_item = item + gridDim.x;
    TileCoord nxt = cur;
    // this lane's folded-ABN coefficients, fetched now so that the epilogue issues no load (the
    // in-order vmcnt wait of an epilogue load would also wait for the next tile's prefetch)
    constexpr int NCO = MODE == FMT_PX ? 2 : 4;
    float sc[NCO], sh[NCO];
    {
      const float *scale = tail + cur.slice * COUTB;
      const float *shift = scale + slices * COUTB;
#pragma unroll
      for (int r = 0; r < NCO; ++r) {
        const int col = (MODE == FMT_PX ? 2 : 4) * kq + r;
        sc[r] = scale[col];
        sh[r] = shift[col];
      }
    }
    for (int s = 0; s < nstages; ++s) {
      if (ABL != 2 || s == 0) {
        TRACE_STAMP();  // chunk begin (before barrier 1)
        __syncthreads();  // every wave is done reading the previous chunk
        TRACE_STAMP();  // after barrier 1
        regs.store(tile, wts);
        __syncthreads();
        TRACE_STAMP();  // after store + barrier 2
        if (ABL != 2) {
          // prefetch the next chunk - or chunk 0 of the next tile - through ONE load site;
          // it is consumed after the next barrier
          int n_ci0 = (s + 1) * CK;
          bool have_next = true;
          if (s + 1 == nstages) {
            n_ci0 = 0;
            have_next = next_item < total;
            if (have_next) {
              nxt = decode_tile<TZ, TY, TX>(next_item, total, tiles_x, tiles_y, tiles_z, B);
              regs.init_tile(nxt.tz0 * STRIDE - PZ, nxt.ty0 * STRIDE - PS, nxt.tx0 * STRIDE - XLO, Hi, Wi);
            }
          }
          if (have_next)
            regs.load(make_rsrc(in + nxt.b * in_ss, in_ss * 4), cin, n_ci0,
                      wpk + (size_t)nxt.slice * per_slice + (size_t)(n_ci0 / CK) * NW);
        }
      }
      if (ABL == 1) {
        acc[0][0] += tile[lane + s];  // keep the staged data live
        continue;
      }

      TRACE_STAMP();  // prefetch issued, MFMA loop begins
      float a_cur[NA], a_nxt[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) a_cur[a] = wts[(a * NITER) * 64 + lane];
      int ad_c[NT], ad_n[NT];  // per-tile LDS word address of the current / next iteration
#pragma unroll
      for (int t = 0; t < NT; ++t) ad_c[t] = base[t] + it_off(0);
      float ring[P];
#pragma unroll
      for (int i = 0; i < P; ++i) ring[i] = tile[ad_c[i % NT] + (i / NT) * ASTEP];
      for (int it = 0; it < NITER; ++it) {
        const int itn = it < NITER - 1 ? it + 1 : NITER - 1;  // last iteration: harmless re-read
#pragma unroll
        for (int a = 0; a < NA; ++a) a_nxt[a] = wts[(a * NITER + itn) * 64 + lane];
        const int off_n = it_off(itn);
#pragma unroll
        for (int t = 0; t < NT; ++t) ad_n[t] = base[t] + off_n;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int a = i / NT, t = i % NT;
          const float bcur = ring[i % P];
          const int ii = i + P;
          if (ii < NS) ring[i % P] = tile[ad_c[ii % NT] + (ii / NT) * ASTEP];
          else ring[i % P] = tile[ad_n[(ii - NS) % NT] + ((ii - NS) / NT) * ASTEP];
          acc[t] = mfma16(a_cur[a], bcur, acc[t]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int a = 0; a < NA; ++a) a_cur[a] = a_nxt[a];
#pragma unroll
        for (int t = 0; t < NT; ++t) ad_c[t] = ad_n[t];
      }
    }

    TRACE_STAMP();  // last chunk's MFMA loop done, epilogue begins
    // epilogue of tile `cur`: y = lrelu(acc * scale + shift) (+ skip); the lane holds rows
    // 4*kq + r of column jcol.  The accumulators are cleared for the next tile.
    // Buffer stores: per-lane byte offset = (this lane's first channel, voxel), scalar offset =
    // remaining channel stride; lanes outside the volume / beyond cout carry kOOB and are dropped.
    const rsrc_t dst = make_rsrc(out + cur.b * out_ss, out_ss * 4);
    const rsrc_t skp = make_rsrc((skip && !UPS && !OUT2) ? skip + cur.b * out_ss : out, out_ss * 4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ct = wave * NT + t;
      const int cx = ct % NXG, cy = (ct / NXG) % TY, cz = ct / (NXG * TY);
      const int oz = cur.tz0 + cz, oy = cur.ty0 + cy;
      const f32x4 av = acc[t];
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (MODE == FMT_PX) {
        const int ox = cur.tx0 + cx * 32 + 2 * jcol;  // rows (co, s): r = 2 * h + s; channels 2*kq + h
        const bool ok = oz < Do && oy < Ho && ox < Wo;
        const int voff = ok ? (2 * kq * out_cs + (oz * Ho + oy) * Wo + ox) * 4 : kOOB;
        [[maybe_unused]] float o2[2][2];  // [x phase][channel 2*kq + h]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v0 = fmaf(av[2 * h], sc[h % NCO], sh[h % NCO]);
          float v1 = fmaf(av[2 * h + 1], sc[h % NCO], sh[h % NCO]);
          v0 = v0 > 0.0f ? v0 : v0 * slope;
          v1 = v1 > 0.0f ? v1 : v1 * slope;
          const int soff = h * out_cs * 4;
          if constexpr (OUT2) {
            o2[0][h] = v0;
            o2[1][h] = v1;
          }
          if ((Wo & 1) == 0) {  // ox even and Wo even: 8-byte aligned pair, both in range
            if (skip && !OUT2) {
              const f32x2 sk = buf_load2(skp, voff, soff);
              v0 += sk[0];
              v1 += sk[1];
            }
            buf_store2(f32x2{v0, v1}, dst, voff, soff);
          } else {
            const int voff1 = (ok && ox + 1 < Wo) ? voff + 4 : kOOB;
            if (skip && !OUT2) {
              v0 += buf_load(skp, voff, soff);
              v1 += buf_load(skp, voff1, soff);
            }
            buf_store(v0, dst, voff, soff);
            buf_store(v1, dst, voff1, soff);
          }
        }
        if constexpr (OUT2) {  // pixel-major copy: channels (2 kq, 2 kq + 1) of pixels ox, ox + 1
          const rsrc_t d2 = make_rsrc(const_cast<float *>(skip) + cur.b * out_ss, out_ss * 4);
          const int pbase = (((oz * Ho + oy) * Wo + ox) * cout + 2 * kq) * 4;
          buf_store2(f32x2{o2[0][0], o2[0][1]}, d2, ok ? pbase : kOOB, 0);
          buf_store2(f32x2{o2[1][0], o2[1][1]}, d2, (ok && ox + 1 < Wo) ? pbase + cout * 4 : kOOB, 0);
        }
      } else {
        const int ox = cur.tx0 + cx * 16 + jcol;  // channels slice*16 + 4*kq + r
        const bool ok = oz < Do && oy < Ho && ox < Wo;
        const int vbase = ((cur.slice * 16 + 4 * kq) * out_cs + (oz * Ho + oy) * Wo + ox) * 4;
        if constexpr (UPS) {
          // ATen upsample_bilinear2d, align_corners: src = dst * (in - 1) / (out - 1); the coarse
          // tensor is (B, cout, Ho/2, Wo/2); value = hy0 * (hx0 v00 + hx1 v01) + hy1 * (hx0 v10 + hx1 v11)
          const int hc = Ho >> 1, wc = Wo >> 1;
          const float sy = Ho > 1 ? (float)(hc - 1) / (float)(Ho - 1) : 0.0f;
          const float sx = Wo > 1 ? (float)(wc - 1) / (float)(Wo - 1) : 0.0f;
          const float fy = sy * (float)oy, fx = sx * (float)ox;
          const int y0 = (int)fy, x0 = (int)fx;
          const int y1 = y0 + (y0 < hc - 1 ? 1 : 0), x1 = x0 + (x0 < wc - 1 ? 1 : 0);
          const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1, lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
          const rsrc_t cs = make_rsrc(skip + (size_t)cur.b * cout * hc * wc, (size_t)cout * hc * wc * 4);
          const int cbase = (cur.slice * 16 + 4 * kq) * hc * wc;
          // the two columns x0, x1 = x0 + 1 of a row come in ONE 8-byte load (half the gather instructions,
          // which bound this layer); at the last column x1 == x0 and the pair is (x0 - 1, x0)
          const bool pair = wc >= 2;  // uniform
          const int xl = x0 < wc - 1 ? x0 : wc - 2;
          const bool last = x0 != xl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool okr = ok && cur.slice * 16 + 4 * kq + r < cout;
            const int soff = r * hc * wc * 4;
            float v00, v01, v10, v11;
            if (pair) {
              const f32x2 p0 = buf_load2(cs, okr ? (cbase + y0 * wc + xl) * 4 : kOOB, soff);
              const f32x2 p1 = buf_load2(cs, okr ? (cbase + y1 * wc + xl) * 4 : kOOB, soff);
              v00 = last ? p0[1] : p0[0];
              v01 = p0[1];
              v10 = last ? p1[1] : p1[0];
              v11 = p1[1];
            } else {
              v00 = buf_load(cs, okr ? (cbase + y0 * wc + x0) * 4 : kOOB, soff);
              v01 = buf_load(cs, okr ? (cbase + y0 * wc + x1) * 4 : kOOB, soff);
              v10 = buf_load(cs, okr ? (cbase + y1 * wc + x0) * 4 : kOOB, soff);
              v11 = buf_load(cs, okr ? (cbase + y1 * wc + x1) * 4 : kOOB, soff);
            }
            float v = fmaf(av[r], sc[r % NCO], sh[r % NCO]);
            v = v > 0.0f ? v : v * slope;
            v += ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
            buf_store(v, dst, okr ? vbase : kOOB, r * out_cs * 4);
          }
        } else {
          [[maybe_unused]] f32x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int voff = (ok && cur.slice * 16 + 4 * kq + r < cout) ? vbase : kOOB;
            float v = fmaf(av[r], sc[r % NCO], sh[r % NCO]);
            v = v > 0.0f ? v : v * slope;
            if (skip && !OUT2) v += buf_load(skp, voff, r * out_cs * 4);
            if constexpr (OUT2) o4[r] = v;
            buf_store(v, dst, voff, r * out_cs * 4);
          }
          if constexpr (OUT2) {  // pixel-major copy: this lane's 4 consecutive channels in one store (cout % 16 == 0)
            const rsrc_t d2 = make_rsrc(const_cast<float *>(skip) + cur.b * out_ss, out_ss * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), d2,
                ok ? (((oz * Ho + oy) * Wo + ox) * cout + cur.slice * 16 + 4 * kq) * 4 : kOOB, 0, 0);
          }
        }
      }
    }
    TRACE_STAMP();  // epilogue done
    if (next_item >= total) break;
    item = next_item;
    cur = nxt;
  }
}

// ---- double-buffered variant: staging hidden inside the MFMA loop --------------------------------
// Measured on the single-buffer kernel above (tools/gpu_trace.py, PMC): the MFMA + ds_read loop in
// isolation sustains ~95 % of the matrix pipe (casmvs_selftest_mfma_rate shapes 4..7), but every
// VALU instruction issued by ANY wave of the SIMD - staging address math, predicate selects,
// register copies at iteration boundaries - takes MFMA issue time (SQ_VALU_MFMA_COEXEC_CYCLES = 0),
// and at ~2 VALU per MFMA the pipe was only ~65 % busy.  This variant is built to issue almost no
// VALU in steady state:
//   * the tile lives in two LDS buffers; while chunk w is multiplied out of buffer w & 1, the
//     registers holding chunk w + 1 are written to the other buffer (ds_write_b128, address register
//     precomputed per thread, buffer selected by an immediate) and the loads of chunk w + 2 are
//     issued (buffer_load_dwordx4: per-thread offset register + scalar offset) from the spare issue
//     slots of the MFMA steps: one barrier per chunk, no staging phase, no address arithmetic;
//   * the whole chunk loop is one flattened, fully unrolled sequence (all LDS offsets immediates,
//     no loop-carried register copies), instantiated once per buffer parity.
// VEC = 4 staging only (Wi % 4 == 0).
template <int MODE, int CK, int NT, int TZ, int TY, int TX, int STRIDE = 1, int KZ = 3>
struct Conv16DbCfg : Conv16Cfg<MODE, STRIDE, CK, NT, TZ, TY, TX, 4, KZ, 3> {
  using Base = Conv16Cfg<MODE, STRIDE, CK, NT, TZ, TY, TX, 4, KZ, 3>;
  static constexpr int BUF = CK * Base::SC + Base::NW;  // floats per buffer
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * sizeof(float);
};

// Staging registers of the double-buffered kernel: like Stager<4>, but every address is a
// precomputed register so that a load / store operation is exactly one memory instruction.
// DEINT (stride-2 layers): rows are stored [even columns | odd columns] (see Stager<5>): a staged 16-byte
// group becomes two 8-byte LDS writes.
template <int CK, int IZ, int IY, int IXR, int SC, int NW, bool DEINT = false>
struct DbStager {
  static_assert(IXR % 4 == 0 && SC % 4 == 0 && NW % 4 == 0, "16-byte groups");
  static constexpr int ROWV = IXR / 4, PLV = IY * ROWV, TOTV = CK * IZ * PLV;
  static constexpr int NK = (TOTV + kThreads - 1) / kThreads;
  static constexpr int NWV = NW / 4, NWR = (NWV + kThreads - 1) / kThreads;
  static constexpr int NOPS = NK + NWR;
  int in_cs, HiWi, Di;
  float *lds_t[NK];   // destination of group k in LDS buffer 0 (kernel constant)
  bool t_ok[NK];      // group k exists (the last pass is partial); wave-level masks, live in SGPRs
  bool w_ok[NWR];
  int cil[NK];        // local channel of group k (kernel constant; only read for channel-padded chunks)
  int voff[2][NK];    // [set] byte offset of group k inside the sample for chunk channel 0, or kOOB
  float *lds_w[NWR];  // destination of weight group i in LDS buffer 0
  int woff[NWR];      // byte offset of weight group i inside a chunk's weight block
  // two register sets: while set s is being written to LDS, the loads of the chunk after are
  // already in flight into set 1 - s, i.e. every load has a whole chunk (~4 us) to land
  f32x4v v[2][NK];
  f32x4v w[2][NWR];

  __device__ __forceinline__ void init_kernel(float *tile0, float *wts0, int in_cs_, int HiWi_, int Di_) {
    in_cs = in_cs_;
    HiWi = HiWi_;
    Di = Di_;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int e = threadIdx.x + k * kThreads;
      const int pl = e / PLV, r = e - pl * PLV;
      const int c = pl / IZ, iz = pl - c * IZ, iy = r / ROWV, xv = r - iy * ROWV;
      cil[k] = c;
      t_ok[k] = e < TOTV;
      lds_t[k] = tile0 + (e < TOTV ? c * SC + iz * (IY * IXR) + iy * IXR + (DEINT ? 2 : 4) * xv : 0);
    }
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int e = threadIdx.x + i * kThreads;
      woff[i] = (e < NWV ? e : 0) * 16;
      w_ok[i] = e < NWV;
      lds_w[i] = wts0 + (e < NWV ? 4 * e : 0);
    }
  }
  template <int S>
  __device__ __forceinline__ void init_tile(int iz0, int iy0, int ix0, int Hi, int Wi) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int e = threadIdx.x + k * kThreads;
      const int pl = e / PLV, r = e - pl * PLV;
      const int c = pl / IZ, iz = pl - c * IZ, iy = r / ROWV, xv = r - iy * ROWV;
      const int gz = iz0 + iz, gy = iy0 + iy, gx = ix0 + 4 * xv;
      const bool inb = e < TOTV && gz >= 0 && gz < Di && gy >= 0 && gy < Hi && gx >= 0 && gx + 3 < Wi;
      voff[S][k] = inb ? (c * in_cs + gz * HiWi + gy * Wi + gx) * 4 : kOOB;
    }
  }
  template <int S>
  __device__ __forceinline__ void kill_plan() {
#pragma unroll
    for (int k = 0; k < NK; ++k) voff[S][k] = kOOB;
  }
  template <int S>
  __device__ __forceinline__ void copy_plan_from_other() {  // same tile, next chunk: the plan carries over
#pragma unroll
    for (int k = 0; k < NK; ++k) voff[S][k] = voff[1 - S][k];
  }
  template <int S, int J>
  __device__ __forceinline__ void load_op(rsrc_t src, rsrc_t wsrc, int cin, int ci0, int wsoff) {
    if constexpr (J < NWR) {
      w[S][J] = buf_load4(wsrc, woff[J], wsoff);
    } else {
      constexpr int k = J - NWR;
      if (ci0 + CK <= cin) {  // wave-uniform; false only for a channel-padded last chunk
        v[S][k] = buf_load4(src, voff[S][k], ci0 * in_cs * 4);
      } else {
        v[S][k] = buf_load4(src, ci0 + cil[k] < cin ? voff[S][k] : kOOB, ci0 * in_cs * 4);
      }
    }
  }
  template <int S, int J, int BUFOFF>  // BUFOFF: float offset of the destination buffer (immediate)
  __device__ __forceinline__ void store_op() const {
    if constexpr (J < NWR) {
      if (w_ok[J]) *reinterpret_cast<f32x4v *>(lds_w[J] + BUFOFF) = w[S][J];
    } else {
      constexpr int k = J - NWR;
      if constexpr (DEINT) {
        if (t_ok[k]) {
          *reinterpret_cast<f32x2 *>(lds_t[k] + BUFOFF) = f32x2{v[S][k][0], v[S][k][2]};            // columns 4 xv, + 2
          *reinterpret_cast<f32x2 *>(lds_t[k] + BUFOFF + IXR / 2) = f32x2{v[S][k][1], v[S][k][3]};  // columns 4 xv + 1, + 3
        }
      } else {
        if (t_ok[k]) *reinterpret_cast<f32x4v *>(lds_t[k] + BUFOFF) = v[S][k];
      }
    }
  }
};

// KZ = 1 (TZ = 1, D = 1): the 3x3 2D layers of FeatureNet; OUT2 = 1: `skip` is a second, pixel-major output
// (as in conv16_kernel).
template <int MODE, int CK, int NT, int TZ, int TY, int TX, int STRIDE = 1, int KZ = 3, int OUT2 = 0>
__global__ __launch_bounds__(kThreads, 2) void conv16db_kernel(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ skip,
    float *__restrict__ out, int B, int cin, int cout, int Di, int Hi, int Wi, int per_slice, int slices,
    int tiles_x, int tiles_y, int tiles_z, float slope) {
  using Cfg = Conv16DbCfg<MODE, CK, NT, TZ, TY, TX, STRIDE, KZ>;
  constexpr int PZ = Cfg::PZ;
  static_assert(STRIDE == 1 || (STRIDE == 2 && MODE == FMT_CI), "stride 2: CI form only");
  constexpr bool DEINT = Cfg::DEINT;
  constexpr int IZ = Cfg::IZ, IY = Cfg::IY, IX = Cfg::IX, SY = Cfg::SY, SZ = Cfg::SZ, SC = Cfg::SC;
  constexpr int NA = Cfg::NA, NITER = Cfg::NITER, ASTEP = Cfg::ASTEP, NW = Cfg::NW, NXG = Cfg::NXG;
  constexpr int XLO = Cfg::XLO, XOFF = Cfg::XOFF, BUF = Cfg::BUF;
  constexpr int COUTB = MODE == FMT_PX ? 8 : 16;
  const int Do = Di / STRIDE, Ho = Hi / STRIDE, Wo = Wi / STRIDE;
  const int nstages = (cin + CK - 1) / CK;
  extern __shared__ float smem[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jcol = lane & 15, kq = lane >> 4;
  const int total = tiles_x * tiles_y * tiles_z * B * slices;
  if ((int)blockIdx.x >= total) return;

  // lane's LDS read pointers (buffer 0): B operand of column tile t, A image 0
  const float *bptr[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ct = wave * NT + t;
    const int cx = ct % NXG, cy = (ct / NXG) % TY, cz = ct / (NXG * TY);
    if (MODE == FMT_PX) bptr[t] = smem + cz * SZ + cy * SY + cx * 32 + 2 * jcol + kq + XOFF;
    else if (DEINT) bptr[t] = smem + kq * SC + (cz * 2) * SZ + (cy * 2) * SY + cx * 16 + jcol;  // parity halves: unit stride
    else bptr[t] = smem + kq * SC + cz * SZ + cy * SY + cx * 16 + jcol + XOFF;
  }
  const float *aptr = smem + CK * SC + lane;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int in_cs = Di * Hi * Wi, out_cs = Do * Ho * Wo;
  const size_t in_ss = (size_t)cin * in_cs, out_ss = (size_t)cout * out_cs;
  const float *tail = wpk + (size_t)slices * per_slice;
  const rsrc_t wsrc = make_rsrc(wpk, ((size_t)slices * per_slice) * 4);

  constexpr int NS = NA * NT;
  constexpr auto it_off = [](int it) constexpr -> int {
    if (DEINT) {  // see conv16_kernel: tap kx of output j reads column 2 j + kx + 3 of the row
      const int kx = it % 3;
      return (it / 9) * SZ + ((it / 3) % 3) * SY + (kx == 1 ? 2 : IX / 2 + (kx == 0 ? 1 : 2));
    }
    return MODE == FMT_PX ? (it / 3) * SZ + (it % 3) * SY : (it / 9) * SZ + ((it / 3) % 3) * SY + (it % 3);
  };
  using St = DbStager<CK, IZ, IY, IX, SC, NW, DEINT>;
  constexpr int NOPS = St::NOPS;
  // side-work schedule of work item w (register set / LDS buffer parity PAR = w & 1):
  //   steps 1 .. NOPS        : load op j of item w + 2  -> register set PAR (free since item w - 1)
  //   steps ST0 + j * SST     : store op j of item w + 1 (set 1 - PAR, loaded during item w - 1)
  //                             -> LDS buffer 1 - PAR
  constexpr int TOTAL_STEPS = NITER * NS;
  // B-operand ring: the ds_read of step g + P is issued right before the MFMA of step g.  The loop is
  // flattened, so the look-ahead may span iterations: the small tiles (NS = 1, 2) get 8 steps too.
  // (NS = 4 keeps P = 4: 8 measured no faster and costs the wide CI tile a wave of occupancy.)
  constexpr int P = NS % 8 == 0 ? 8 : (NS % 4 == 0 ? 4 : (TOTAL_STEPS >= 16 ? 8 : 2));
  constexpr auto b_tile = [](int g) constexpr -> int { return ((g < TOTAL_STEPS ? g : TOTAL_STEPS - 1) % NS) % NT; };
  constexpr auto b_off = [it_off](int g) constexpr -> int {
    const int gg = g < TOTAL_STEPS ? g : TOTAL_STEPS - 1;  // beyond the chunk: harmless re-read of the last operand
    return it_off(gg / NS) + ((gg % NS) / NT) * ASTEP;
  };
  static_assert(2 * NOPS + 4 <= TOTAL_STEPS, "not enough MFMA steps to hide the staging operations");
  constexpr int ST0 = NOPS + 2, SST = (TOTAL_STEPS - ST0 - 1) / NOPS;

  // prefetch cursor: the (tile, chunk) most recently put in flight
  struct Cursor {
    TileCoord tc;
    int item, chunk;
    bool valid;
  };
  St regs;
  auto advance = [&](Cursor &c, auto set_) {  // -> next work item; plan of register set S for it
    constexpr int S = decltype(set_)::value;
    if (++c.chunk == nstages) {
      c.chunk = 0;
      c.item += gridDim.x;
      c.valid = c.item < total;
      if (c.valid) {
        c.tc = decode_tile<TZ, TY, TX>(c.item, total, tiles_x, tiles_y, tiles_z, B);
        regs.template init_tile<S>(c.tc.tz0 * STRIDE - PZ, c.tc.ty0 * STRIDE - 1, c.tc.tx0 * STRIDE - XLO, Hi, Wi);
      } else {
        regs.template kill_plan<S>();  // no more work: the (unconditional) loads of this set read nothing
      }
    } else {
      regs.template copy_plan_from_other<S>();
    }
  };
  auto wsoff_of = [&](const Cursor &c) { return (int)(((size_t)c.tc.slice * per_slice + (size_t)c.chunk * NW) * 4); };

  regs.init_kernel(smem, smem + CK * SC, in_cs, Hi * Wi, Di);
  Cursor pf;  // prefetch cursor
  pf.item = blockIdx.x;
  pf.chunk = 0;
  pf.valid = true;
  pf.tc = decode_tile<TZ, TY, TX>(pf.item, total, tiles_x, tiles_y, tiles_z, B);
  regs.template init_tile<0>(pf.tc.tz0 * STRIDE - PZ, pf.tc.ty0 * STRIDE - 1, pf.tc.tx0 * STRIDE - XLO, Hi, Wi);
  TileCoord cur = pf.tc;  // tile being computed
  int cur_chunk = 0, tiles_done = 0;
  // prologue: work item 0 -> set 0 -> buffer 0; work item 1 -> set 1 (in flight)
  {
    const rsrc_t src = make_rsrc(in + pf.tc.b * in_ss, in_ss * 4);
    const int ws = wsoff_of(pf);
    static_for<NOPS>([&](auto j_) { regs.template load_op<0, decltype(j_)::value>(src, wsrc, cin, 0, ws); });
    static_for<NOPS>([&](auto j_) { regs.template store_op<0, decltype(j_)::value, 0>(); });
  }
  __syncthreads();
  advance(pf, std::integral_constant<int, 1>{});
  bool next_valid = pf.valid;  // work item w + 1 exists (its loads are in flight / landed)
  if (pf.valid) {
    const rsrc_t src = make_rsrc(in + pf.tc.b * in_ss, in_ss * 4);
    const int ws = wsoff_of(pf), ci0 = pf.chunk * CK;
    static_for<NOPS>([&](auto j_) { regs.template load_op<1, decltype(j_)::value>(src, wsrc, cin, ci0, ws); });
  }

  constexpr int NCO = MODE == FMT_PX ? 2 : 4;
  float sc[NCO], sh[NCO];
  auto load_coeffs = [&](int slice) {
    const float *scale = tail + slice * COUTB;
    const float *shift = scale + slices * COUTB;
#pragma unroll
    for (int r = 0; r < NCO; ++r) {
      sc[r] = scale[NCO * kq + r];
      sh[r] = shift[NCO * kq + r];
    }
  };
  load_coeffs(cur.slice);

  // one work item (chunk) computed out of buffer PAR; returns false after the last tile
  auto run_item = [&](auto par_) -> bool {
    constexpr int PAR = decltype(par_)::value;
    constexpr int RD = PAR * BUF, WR = (1 - PAR) * BUF;  // float offsets of the read / write buffers
    const bool store_next = next_valid;  // register set 1 - PAR holds work item w + 1
    // put work item w + 2 in flight into register set PAR right away (a whole chunk to land)
    bool load_next = false;
    if (store_next) {
      advance(pf, std::integral_constant<int, PAR>{});
      load_next = pf.valid;
    }
    const TileCoord ltc = load_next ? pf.tc : cur;
    const rsrc_t lsrc = make_rsrc(in + ltc.b * in_ss, in_ss * 4);
    const int lws = load_next ? wsoff_of(pf) : 0, lci0 = load_next ? pf.chunk * CK : 0;
    next_valid = load_next;

    float a_cur[NA], a_nxt[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) a_cur[a] = aptr[RD + (a * NITER) * 64];
    float ring[P];
    static_for<P>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      ring[i] = bptr[b_tile(i)][RD + b_off(i)];
    });
    // one flattened, fully unrolled loop over the NITER * NS steps: every index and every LDS
    // offset below is a compile-time constant
    static_for<TOTAL_STEPS>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      constexpr int it = g / NS, i = g % NS;
      constexpr int a = i / NT, t = i % NT;
      constexpr int itn = it < NITER - 1 ? it + 1 : NITER - 1;
      if constexpr (i == 0) {
#pragma unroll
        for (int aa = 0; aa < NA; ++aa) a_nxt[aa] = aptr[RD + (aa * NITER + itn) * 64];
      }
      const float bcur = ring[g % P];
      ring[g % P] = bptr[b_tile(g + P)][RD + b_off(g + P)];
      acc[t] = mfma16(a_cur[a], bcur, acc[t]);
      // ---- side work in this step's spare issue slots ----
      // (issued unconditionally - a dead set loads with out-of-range offsets and its stores land in
      // the buffer nobody reads - so that the code stays branch-free and the compiler can emit
      // counted vmcnt waits instead of draining the loads it has just issued)
      if constexpr (g >= 1 && g <= NOPS) regs.template load_op<PAR, g - 1>(lsrc, wsrc, cin, lci0, lws);
      if constexpr (g >= ST0 && (g - ST0) % SST == 0 && (g - ST0) / SST < NOPS)
        regs.template store_op<1 - PAR, (g - ST0) / SST, WR>();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (i == NS - 1) {
#pragma unroll
        for (int aa = 0; aa < NA; ++aa) a_cur[aa] = a_nxt[aa];
      }
    });

    bool more = true;
    if (++cur_chunk == nstages) {  // tile finished: epilogue, then switch to the next tile
#pragma unroll
      for (int dn = 0; dn < CASMVS_MFMA_DRAIN_NOPS; ++dn) asm volatile("s_nop 15");   // debug builds (co-residency experiment)
      const rsrc_t dst = make_rsrc(out + cur.b * out_ss, out_ss * 4);
      const rsrc_t skp = make_rsrc((skip && !OUT2) ? skip + cur.b * out_ss : out, out_ss * 4);
      [[maybe_unused]] const rsrc_t d2 = make_rsrc(OUT2 ? const_cast<float *>(skip) + cur.b * out_ss : out, out_ss * 4);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int ct = wave * NT + t;
        const int cx = ct % NXG, cy = (ct / NXG) % TY, cz = ct / (NXG * TY);
        const int oz = cur.tz0 + cz, oy = cur.ty0 + cy;
        const f32x4 av = acc[t];
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (MODE == FMT_PX) {
          const int ox = cur.tx0 + cx * 32 + 2 * jcol;
          const bool ok = oz < Do && oy < Ho && ox < Wo;  // Wo % 4 == 0 here: the pair is in range
          const int voff = ok ? (2 * kq * out_cs + (oz * Ho + oy) * Wo + ox) * 4 : kOOB;
          [[maybe_unused]] float o2[2][2];  // [x phase][channel 2*kq + h]
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float v0 = fmaf(av[2 * h], sc[h % NCO], sh[h % NCO]);
            float v1 = fmaf(av[2 * h + 1], sc[h % NCO], sh[h % NCO]);
            v0 = v0 > 0.0f ? v0 : v0 * slope;
            v1 = v1 > 0.0f ? v1 : v1 * slope;
            const int soff = h * out_cs * 4;
            if constexpr (OUT2) {
              o2[0][h] = v0;
              o2[1][h] = v1;
            } else if (skip) {
              const f32x2 sk = buf_load2(skp, voff, soff);
              v0 += sk[0];
              v1 += sk[1];
            }
            buf_store2(f32x2{v0, v1}, dst, voff, soff);
          }
          if constexpr (OUT2) {  // pixel-major copy: channels (2 kq, 2 kq + 1) of pixels ox, ox + 1
            const int pbase = (((oz * Ho + oy) * Wo + ox) * cout + 2 * kq) * 4;
            buf_store2(f32x2{o2[0][0], o2[0][1]}, d2, ok ? pbase : kOOB, 0);
            buf_store2(f32x2{o2[1][0], o2[1][1]}, d2, ok ? pbase + cout * 4 : kOOB, 0);
          }
        } else {
          const int ox = cur.tx0 + cx * 16 + jcol;
          const bool ok = oz < Do && oy < Ho && ox < Wo;
          const int vbase = ((cur.slice * 16 + 4 * kq) * out_cs + (oz * Ho + oy) * Wo + ox) * 4;
          [[maybe_unused]] f32x4 o4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int voff = (ok && cur.slice * 16 + 4 * kq + r < cout) ? vbase : kOOB;
            float v = fmaf(av[r], sc[r % NCO], sh[r % NCO]);
            v = v > 0.0f ? v : v * slope;
            if constexpr (OUT2) o4[r] = v;
            else if (skip) v += buf_load(skp, voff, r * out_cs * 4);
            buf_store(v, dst, voff, r * out_cs * 4);
          }
          if constexpr (OUT2)  // pixel-major copy: this lane's 4 consecutive channels in one store (cout % 16 == 0)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o4), d2,
                ok ? (((oz * Ho + oy) * Wo + ox) * cout + cur.slice * 16 + 4 * kq) * 4 : kOOB, 0, 0);
        }
      }
      cur_chunk = 0;
      ++tiles_done;
      if (!store_next) {
        more = false;  // no further work item: this was the last tile
      } else {
        cur = decode_tile<TZ, TY, TX>(blockIdx.x + tiles_done * gridDim.x, total, tiles_x, tiles_y, tiles_z, B);
        load_coeffs(cur.slice);
      }
    }
    __syncthreads();  // the other buffer is published, this one is free
    return more;
  };
  for (;;) {
    if (!run_item(std::integral_constant<int, 0>{})) break;
    if (!run_item(std::integral_constant<int, 1>{})) break;
  }
}

// ---- ConvTranspose3d k3 s2 p1 op1 on 16x16x4 -------------------------------------------------------
// out[2i - 1 + k] += in[i] * w[k] per axis: even outputs o = 2m take (k = 1, i = m); odd outputs
// o = 2m + 1 take (k = 2, i = m) and (k = 0, i = m + 1).  A column tile is 16 input cells m along
// x; for every cell a workgroup produces all 8 output parities: the 2 x 2 x 2 neighbourhood
// (m + d) of a cell is read ONCE from LDS (8 B operands per channel quad) and feeds the 27 (TCI)
// / 18 (TPX) MFMAs of the four (pz, py) passes.  TCI keeps two accumulators per pass (x parity),
// TPX folds the x parity into the rows (co, px).
#ifndef CASMVS_DECONV_PREFETCH
#define CASMVS_DECONV_PREFETCH 1   // 0: A/B builds without the skip prefetch under the last chunk
#endif
template <int MODE, int CK, int NT, int TZ, int TY, int TX, int VEC>
struct Deconv16Cfg {
  static_assert(MODE == FMT_TCI || MODE == FMT_TPX, "deconv16: TCI or TPX");
  static constexpr int NXG = TX / 16;
  static_assert(TX % 16 == 0 && TZ * TY * NXG == 4 * NT, "tile = 4 waves x NT column tiles");
  static constexpr int IZ = TZ + 1, IY = TY + 1, IX = VEC == 4 ? TX + 4 : TX + 1;  // cells m .. m + 1
  static constexpr int SY = IX, SZ = IY * IX;
  static constexpr int SC = round_up_to_16_mod_32(IZ * SZ);
  static constexpr int NQ = CK / 4;
  static constexpr int UI = MODE == FMT_TCI ? 27 : 18;  // images per channel quad
  static constexpr int NW = NQ * UI * 64;
  static constexpr size_t LDS_BYTES = (size_t)(CK * SC + NW) * sizeof(float);
};

template <int MODE, int CK, int NT, int TZ, int TY, int TX, int VEC>
__global__ __launch_bounds__(kThreads) void deconv16_kernel(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ skip,
    float *__restrict__ out, int cin, int cout, int Di, int Hi, int Wi, int per_slice, int tiles_x,
    int tiles_y, float slope) {
  using Cfg = Deconv16Cfg<MODE, CK, NT, TZ, TY, TX, VEC>;
  constexpr int IZ = Cfg::IZ, IY = Cfg::IY, IX = Cfg::IX, SY = Cfg::SY, SZ = Cfg::SZ, SC = Cfg::SC;
  constexpr int NQ = Cfg::NQ, NW = Cfg::NW, NXG = Cfg::NXG, UI = Cfg::UI;
  constexpr int COUTB = MODE == FMT_TPX ? 8 : 16;
  constexpr int NACC = MODE == FMT_TCI ? 2 : 1;
  extern __shared__ float smem[];
  float *tile = smem;
  float *wts = smem + CK * SC;  // [NQ][UI][64]
  const int nstages = (cin + CK - 1) / CK;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jcol = lane & 15, kq = lane >> 4;
  // XCD-major tile order, z fastest (see decode_tile): the halos of neighbouring tiles share an L2
  const int tiles_z = gridDim.x / (tiles_x * tiles_y);
  const int bid = xcd_major(blockIdx.x, gridDim.x);
  const int tz0 = (bid % tiles_z) * TZ;
  const int tx0 = ((bid / tiles_z) % tiles_x) * TX;
  const int ty0 = (bid / (tiles_z * tiles_x)) * TY;
  const int b = blockIdx.y, slice = blockIdx.z;
  const int slices = gridDim.z;

  int base[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ct = wave * NT + t;
    const int cx = ct % NXG, cy = (ct / NXG) % TY, cz = ct / (NXG * TY);
    base[t] = kq * SC + cz * SZ + cy * SY + cx * 16 + jcol;
  }
  f32x4 acc[4][NACC][NT];  // [pass = 2*pz + py][x parity (TCI)][tile]
#pragma unroll
  for (int ps = 0; ps < 4; ++ps)
#pragma unroll
    for (int p = 0; p < NACC; ++p)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[ps][p][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int in_cs = Di * Hi * Wi;
  const size_t in_ss = (size_t)cin * in_cs;
  const rsrc_t src = make_rsrc(in + b * in_ss, in_ss * 4);
  const float *wslice = wpk + (size_t)slice * per_slice;
  const float *scale = wpk + (size_t)slices * per_slice + slice * COUTB;
  const float *shift = scale + slices * COUTB;
  Stager<VEC, CK, IZ, IY, IX, SC, NW> regs;
  regs.init_kernel(in_cs, Hi * Wi, Di);
#ifdef CASMVS_TRACE
  int tr_n = 0;
#endif
  TRACE_STAMP();  // kernel start
  regs.init_tile(tz0, ty0, tx0, Hi, Wi);
  regs.load(src, cin, 0, wslice);
  TRACE_STAMP();  // first loads issued

  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const int out_cs = Do * Ho * Wo;
  const size_t out_ss = (size_t)cout * out_cs;
  constexpr int NCO = MODE == FMT_TCI ? 4 : 2;
  // (Tried: warming the skip tensor's cache lines with fire-and-forget loads issued here - the
  // in-order vmcnt made the first stage wait for them and the epilogue did not get faster.)

  // Epilogue operands fetched BEFORE the last chunk's MFMA loop (round 3): the per-lane coefficients at kernel start, and
  // - where the registers allow: TPX (32 registers) and the one-tile TCI form - every skip value of the tile right after
  // the last staging barrier, so that the epilogue's global round trip runs under the MFMAs instead of after them
  // (conv11 was at MFMA time + HBM time: the two did not overlap inside a workgroup).  No staging load is younger than
  // them, so the in-order vmcnt wait of the epilogue waits for nothing else.
  constexpr bool PREFETCH_SKIP = CASMVS_DECONV_PREFETCH && (MODE == FMT_TPX || NT == 1);
  const rsrc_t dst = make_rsrc(out + b * out_ss, out_ss * 4);
  const rsrc_t skp = make_rsrc(skip ? skip + b * out_ss : out, out_ss * 4);
  float sc[NCO], sh[NCO];
#pragma unroll
  for (int h = 0; h < NCO; ++h) {
    sc[h] = scale[NCO * kq + h];
    sh[h] = shift[NCO * kq + h];
  }
  int vcell[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ct = wave * NT + t;
    const int cx = ct % NXG, cy = (ct / NXG) % TY, cz = ct / (NXG * TY);
    const int mz = tz0 + cz, my = ty0 + cy, mx = tx0 + cx * 16 + jcol;
    const bool ok = mz < Di && my < Hi && mx < Wi;
    // per-lane part: first channel of the lane (slice*COUTB + NCO*kq) and the cell's even-corner voxel
    vcell[t] = ok ? ((slice * COUTB + NCO * kq) * out_cs + (2 * mz * Ho + 2 * my) * Wo + 2 * mx) * 4 : kOOB;
  }
  [[maybe_unused]] f32x2 skall[PREFETCH_SKIP ? NT * 4 : 1][NCO];

  for (int s = 0; s < nstages; ++s) {
    __syncthreads();
    TRACE_STAMP();  // after barrier 1
    regs.store(tile, wts);
    __syncthreads();
    TRACE_STAMP();  // after store + barrier 2
    if (s + 1 < nstages) {
      regs.load(src, cin, (s + 1) * CK, wslice + (size_t)(s + 1) * NW);
    } else if (PREFETCH_SKIP && skip) {
#pragma unroll
      for (int bi = 0; bi < NT * 4; ++bi)
#pragma unroll
        for (int h = 0; h < NCO; ++h) {
          const int t = bi / 4, ps = bi % 4;
          const int voff = (slice * COUTB + NCO * kq + h < cout) ? vcell[t] : kOOB;
          skall[PREFETCH_SKIP ? bi : 0][h] = buf_load2(skp, voff, (h * out_cs + ((ps >> 1) * Ho + (ps & 1)) * Wo) * 4);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float aw[UI];  // this quad's images
#pragma unroll
      for (int i = 0; i < UI; ++i) aw[i] = wts[(q * UI + i) * 64 + lane];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float bv[2][2][2];  // [dz][dy][dx]: cells m + d
#pragma unroll
        for (int d = 0; d < 8; ++d)
          bv[d >> 2][(d >> 1) & 1][d & 1] = tile[base[t] + q * 4 * SC + (d >> 2) * SZ + ((d >> 1) & 1) * SY + (d & 1)];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int pz = ps >> 1, py = ps & 1;
#pragma unroll
          for (int zt = 0; zt < (pz ? 2 : 1); ++zt) {
            const int kz = pz ? (zt == 0 ? 2 : 0) : 1, dz = (pz && zt == 1) ? 1 : 0;
#pragma unroll
            for (int yt = 0; yt < (py ? 2 : 1); ++yt) {
              const int ky = py ? (yt == 0 ? 2 : 0) : 1, dy = (py && yt == 1) ? 1 : 0;
              const int r9 = kz * 3 + ky;
              if (MODE == FMT_TCI) {
                acc[ps][0][t] = mfma16(aw[(r9 * 3 + 1) % UI], bv[dz][dy][0], acc[ps][0][t]);                // px 0: k = 1, i = m
                acc[ps][NACC - 1][t] = mfma16(aw[(r9 * 3 + 2) % UI], bv[dz][dy][0], acc[ps][NACC - 1][t]);  // px 1: k = 2, i = m
                acc[ps][NACC - 1][t] = mfma16(aw[(r9 * 3 + 0) % UI], bv[dz][dy][1], acc[ps][NACC - 1][t]);  // px 1: k = 0, i = m+1
              } else {
                acc[ps][0][t] = mfma16(aw[(r9 * 2 + 0) % UI], bv[dz][dy][0], acc[ps][0][t]);
                acc[ps][0][t] = mfma16(aw[(r9 * 2 + 1) % UI], bv[dz][dy][1], acc[ps][0][t]);
              }
            }
          }
        }
      }
    }
  }

  TRACE_STAMP();  // MFMA loops done
  // Without the prefetch (wide TCI tile: its two x-parity accumulators leave no room for 64 more registers) the skip
  // loads are issued in batches ahead of the stores.  Interleaved (load, wait, add, store per element) the in-order vmcnt
  // wait of load i also waits for store i-1: 32 serial memory round trips per wave - measured 38k of the 61k cycles of a
  // conv11 workgroup (tools/gpu_trace2.py).
  constexpr int BP = PREFETCH_SKIP ? NT * 4 : 2;  // (tile, pass) pairs per batch
  static_assert((NT * 4) % BP == 0, "batches cover the tile x pass pairs");
#pragma unroll
  for (int b0 = 0; b0 < NT * 4; b0 += BP) {
    [[maybe_unused]] f32x2 sk[PREFETCH_SKIP ? 1 : BP][NCO];
    if constexpr (!PREFETCH_SKIP) {
      if (skip) {
#pragma unroll
        for (int bi = 0; bi < BP; ++bi)
#pragma unroll
          for (int h = 0; h < NCO; ++h) {
            const int t = (b0 + bi) / 4, ps = (b0 + bi) % 4;
            const int voff = (slice * COUTB + NCO * kq + h < cout) ? vcell[t] : kOOB;
            sk[bi][h] = buf_load2(skp, voff, (h * out_cs + ((ps >> 1) * Ho + (ps & 1)) * Wo) * 4);
          }
      }
    }
#pragma unroll
    for (int bi = 0; bi < BP; ++bi) {
      const int t = (b0 + bi) / 4, ps = (b0 + bi) % 4;
      const int pz = ps >> 1, py = ps & 1;
#pragma unroll
      for (int h = 0; h < NCO; ++h) {
        // TCI: h = row r -> channel 4*kq + r, pair = (parity 0, parity 1) accumulators
        // TPX: h -> channel 2*kq + h, pair = rows (2h, 2h+1) of the single accumulator
        const int voff = (slice * COUTB + NCO * kq + h < cout) ? vcell[t] : kOOB;
        const int soff = (h * out_cs + (pz * Ho + py) * Wo) * 4;
        float v0 = MODE == FMT_TCI ? acc[ps][0][t][h] : acc[ps][0][t][(2 * h) & 3];
        float v1 = MODE == FMT_TCI ? acc[ps][NACC - 1][t][h] : acc[ps][0][t][(2 * h + 1) & 3];
        v0 = fmaf(v0, sc[h], sh[h]);
        v1 = fmaf(v1, sc[h], sh[h]);
        v0 = v0 > 0.0f ? v0 : v0 * slope;
        v1 = v1 > 0.0f ? v1 : v1 * slope;
        if (skip) {
          const f32x2 sv = PREFETCH_SKIP ? skall[PREFETCH_SKIP ? b0 + bi : 0][h] : sk[PREFETCH_SKIP ? 0 : bi][h];
          v0 += sv[0];
          v1 += sv[1];
        }
        buf_store2(f32x2{v0, v1}, dst, voff, soff);
      }
    }
  }
  TRACE_STAMP();  // epilogue issued
#ifdef CASMVS_TRACE
  __builtin_amdgcn_s_waitcnt(0);
#endif
  TRACE_STAMP();  // epilogue drained
}

// ---- `prob` head (Cout = 1, with bias, no activation): VALU kernel -----------------------------------
// One output channel is not a matrix problem (the MFMA forms would waste >= 75 % of their rows and
// were measured at 5 TFLOP/s): each thread produces 4 consecutive x of one (z, y) with fp32 FMAs.
// Per (ci, kz, ky) it reads 6 consecutive inputs from the LDS halo tile as one ds_read_b128 + one
// ds_read_b64 (rows are stored so that the group starts 16-byte aligned) and does 12 FMAs with 3
// weights that live in SGPRs (wide uniform scalar loads from the channel's contiguous 27-tap row).  Memory-bound by design:
// 8 input channels + 1 output per voxel.
template <int CK, int TZ, int TY, int TX, int VEC>
struct ProbCfg {
  static_assert(TX == 32 && TY == 8 && TZ == 4, "thread map: 8 x-groups x 8 y x 4 z");
  static constexpr int IZ = TZ + 2, IY = TY + 2;
  // VEC 1: rows start at x0 - 1 (TX + 2 needed, padded to TX + 4); VEC 4: at the aligned x0 - 4 (TX + 8)
  static constexpr int IX = VEC == 4 ? TX + 8 : TX + 4;
  static constexpr int XLO = VEC == 4 ? 4 : 1;
  static constexpr int SY = IX, SZ = IY * IX, SC = IZ * SZ;
  static constexpr int NW = 64;  // Stager carries a (dummy) weight block; the weights are read as scalars
  static constexpr size_t LDS_BYTES = (size_t)(CK * SC + NW) * sizeof(float);
};

template <int CK, int TZ, int TY, int TX, int VEC>
__global__ __launch_bounds__(kThreads, 4) void prob_valu_kernel(
    const float *__restrict__ in, const float *__restrict__ wpk, float *__restrict__ out, int cin,
    int Di, int Hi, int Wi, int tiles_x, int tiles_y, float slope) {
  using Cfg = ProbCfg<CK, TZ, TY, TX, VEC>;
  constexpr int IZ = Cfg::IZ, IY = Cfg::IY, IX = Cfg::IX, SY = Cfg::SY, SZ = Cfg::SZ, SC = Cfg::SC, NW = Cfg::NW;
  extern __shared__ float smem[];
  float *tile = smem;
  float *wts = smem + CK * SC;
  const int nstages = (cin + CK - 1) / CK;
  const int tiles_z = gridDim.x / (tiles_x * tiles_y);
  const int bid = xcd_major(blockIdx.x, gridDim.x);  // XCD-major tile order, z fastest (see decode_tile)
  const int tz0 = (bid % tiles_z) * TZ;
  const int tx0 = ((bid / tiles_z) % tiles_x) * TX;
  const int ty0 = (bid / (tiles_z * tiles_x)) * TY;
  const int b = blockIdx.y;
  const int xi = threadIdx.x & 7, yi = (threadIdx.x >> 3) & 7, zi = threadIdx.x >> 6;
  // (kz, ky, cil) = 0; VEC 1: the 6 inputs x-1 .. x+4 start at local 4 xi (16-byte aligned);
  // VEC 4: they start at local 4 xi + 3, i.e. one word, the aligned group 4 xi + 4, one word
  const float *row0 = tile + zi * SZ + yi * SY + 4 * xi;

  const int in_cs = Di * Hi * Wi;
  const size_t in_ss = (size_t)cin * in_cs;
  const rsrc_t src = make_rsrc(in + b * in_ss, in_ss * 4);
  const int rows = (cin + 7) / 8 * 8;  // packed image: [channel pair][tap (32)][2]
  const float *scale = wpk + (size_t)rows * 32;
  const float *shift = scale + 4;
  Stager<VEC, CK, IZ, IY, IX, SC, NW> regs;
  regs.init_kernel(in_cs, Hi * Wi, Di);
#ifdef CASMVS_TRACE
  int tr_n = 0;
#endif
  TRACE_STAMP();  // kernel start
  regs.init_tile(tz0 - 1, ty0 - 1, tx0 - Cfg::XLO, Hi, Wi);
  regs.load(src, cin, 0, wpk);
  TRACE_STAMP();  // first loads issued
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // VEC 4: the LDS offset of each staged 16-byte group is a kernel constant: keep it in a register instead
  // of re-deriving it (divisions by constants) at every stage - this kernel has the registers to spare
  [[maybe_unused]] int lds_off[VEC == 4 ? Stager<4, CK, IZ, IY, IX, SC, NW>::NK : 1];
  if constexpr (VEC == 4) {
    using S4 = Stager<4, CK, IZ, IY, IX, SC, NW>;
#pragma unroll
    for (int k = 0; k < S4::NK; ++k) {
      const auto p = S4::pos_of(k);
      lds_off[k] = p.valid ? p.cil * SC + p.iz * (IY * IX) + p.iy * IX + 4 * p.xv : -1;
    }
  }
  for (int s = 0; s < nstages; ++s) {
    __syncthreads();
    TRACE_STAMP();  // after barrier 1
    if constexpr (VEC == 4) {
#pragma unroll
      for (int k = 0; k < Stager<4, CK, IZ, IY, IX, SC, NW>::NK; ++k)
        if (lds_off[k] >= 0) *reinterpret_cast<f32x4v *>(tile + lds_off[k]) = regs.v[k];
    } else {
      regs.store(tile, wts);
    }
    __syncthreads();
    TRACE_STAMP();  // after store + barrier 2
    if (s + 1 < nstages) regs.load(src, cin, (s + 1) * CK, wpk);
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const int ci = s * CK + c;  // channels >= cin: staged zeros x zero weights
      const float *wci = wpk + (ci >> 1) * 64 + (ci & 1);  // wave-uniform scalar loads; tap t of this channel at [2 t]
#pragma unroll
      for (int r9 = 0; r9 < 9; ++r9) {
        const float *row = row0 + c * SC + (r9 / 3) * SZ + (r9 % 3) * SY;
        float i0, i5;
        f32x4v m;
        if constexpr (VEC == 4) {
          // The two edge inputs x - 1 and x + 4 are the neighbouring lanes' m[3] / m[0]: fetched with DPP row
          // shifts.  (As single-dword LDS reads at word 4 xi + 3 they hit 8 of the 32 banks from all 64 lanes:
          // 8-way conflicts that made the kernel LDS-bound - halving its VALU count changed nothing.)  Only
          // the first / last lane of an 8-lane row needs the tile's halo column: one broadcast read per row.
          m = *reinterpret_cast<const f32x4v *>(row + 4);
          const float *rb = row - 4 * xi;  // the row's first staged word (x0 - 4)
          const float h0 = rb[3], h5 = rb[36];
          // (the empty asm pins the two elements in their own registers: hipcc 7.2 otherwise folds BOTH DPP
          //  sources to element 0 of the 16-byte load - reproduced in a 10-line kernel)
          float m3 = m[3], m0 = m[0];
          asm volatile("" : "+v"(m3), "+v"(m0));
          const float l0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m3), 0x111, 0xf, 0xf, false));  // row_shr:1
          const float l5 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m0), 0x101, 0xf, 0xf, false));  // row_shl:1
          i0 = xi == 0 ? h0 : l0;
          i5 = xi == 7 ? h5 : l5;
        } else {
          const f32x4v a = *reinterpret_cast<const f32x4v *>(row);
          const f32x2 e = *reinterpret_cast<const f32x2 *>(row + 4);
          i0 = a[0];
          m = f32x4v{a[1], a[2], a[3], e[0]};
          i5 = e[1];
        }
        const float w0 = wci[(r9 * 3 + 0) * 2], w1 = wci[(r9 * 3 + 1) * 2], w2 = wci[(r9 * 3 + 2) * 2];
        acc[0] = fmaf(m[1], w2, fmaf(m[0], w1, fmaf(i0, w0, acc[0])));
        acc[1] = fmaf(m[2], w2, fmaf(m[1], w1, fmaf(m[0], w0, acc[1])));
        acc[2] = fmaf(m[3], w2, fmaf(m[2], w1, fmaf(m[1], w0, acc[2])));
        acc[3] = fmaf(i5, w2, fmaf(m[3], w1, fmaf(m[2], w0, acc[3])));
      }
    }
  }
  TRACE_STAMP();  // FMA loops done
  const int oz = tz0 + zi, oy = ty0 + yi, ox = tx0 + 4 * xi;
  const rsrc_t dst = make_rsrc(out + (size_t)b * in_cs, (size_t)in_cs * 4);
  const float sc0 = scale[0], sh0 = shift[0];
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = fmaf(acc[i], sc0, sh0);
    v[i] = v[i] > 0.0f ? v[i] : v[i] * slope;
  }
  const bool ok = oz < Di && oy < Hi;
  const int vbase = ((oz * Hi + oy) * Wi + ox) * 4;
  if ((Wi & 3) == 0) {  // 16-byte aligned group, entirely inside or outside the row
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4v{v[0], v[1], v[2], v[3]}), dst,
                                           (ok && ox < Wi) ? vbase : kOOB, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) buf_store(v[i], dst, (ok && ox + i < Wi) ? vbase + 4 * i : kOOB, 0);
  }
}

// ---- `prob` head, packed-math form (W % 4 == 0, aligned input) ---------------------------------------------------------
// The kernel above issues 19 instructions per 12 FMAs (one b128 + two b32 LDS reads, two DPP moves, two selects per
// input row and channel): it is instruction-bound at 3.6x its HBM time.  Here the LDS tile interleaves the two channels
// of a PAIR ([x][2]), rows start at x0 - 1 so that the 6 x 2 inputs a thread needs per (pair, kz, ky) are three aligned
// ds_read_b128, and every FMA is a v_pk_fma_f32 over the channel pair with the weight pair as a scalar operand: 15
// instructions per 24 FMAs.  The two channel-parity partial sums of an output are added at the end.
struct ProbPkCfg {
  static constexpr int TZ = 4, TY = 8, TX = 32, IZ = TZ + 2, IY = TY + 2;
  static constexpr int SY = 76;             // floats per row: 38 channel pairs (37 used: x0 - 1 .. x0 + 35), 16-byte multiple
  static constexpr int SZ = IY * SY, SP = IZ * SZ;   // per plane, per channel pair
  static constexpr int CKP = 2;             // channel pairs per stage
  static constexpr int GROUPS = 2 * CKP * IZ * IY * 10;   // 16-byte global groups per stage (10 per row: x0 - 4 .. x0 + 35)
  static constexpr int NK = (GROUPS + kThreads - 1) / kThreads;
  static constexpr size_t LDS_BYTES = (size_t)CKP * SP * sizeof(float);
};

__global__ __launch_bounds__(kThreads, 4) void prob_pk_kernel(const float *__restrict__ in, const float *__restrict__ wpk,
                                                             float *__restrict__ out, int cin, int Di, int Hi, int Wi,
                                                             int tiles_x, int tiles_y, float slope) {
  using Cfg = ProbPkCfg;
  constexpr int IZ = Cfg::IZ, IY = Cfg::IY, SY = Cfg::SY, SZ = Cfg::SZ, SP = Cfg::SP, NK = Cfg::NK;
  extern __shared__ float smem[];
  float *tile = smem;
  const int nstages = (cin + 2 * Cfg::CKP - 1) / (2 * Cfg::CKP);
  const int tiles_z = gridDim.x / (tiles_x * tiles_y);
  const int bid = xcd_major(blockIdx.x, gridDim.x);
  const int tz0 = (bid % tiles_z) * Cfg::TZ;
  const int tx0 = ((bid / tiles_z) % tiles_x) * Cfg::TX;
  const int ty0 = (bid / (tiles_z * tiles_x)) * Cfg::TY;
  const int b = blockIdx.y;
  const int xi = threadIdx.x & 7, yi = (threadIdx.x >> 3) & 7, zi = threadIdx.x >> 6;
  const int in_cs = Di * Hi * Wi;
  const size_t in_ss = (size_t)cin * in_cs;
  const rsrc_t src = make_rsrc(in + b * in_ss, in_ss * 4);
  const int rows = (cin + 7) / 8 * 8;
  const float *scale = wpk + (size_t)rows * 32;
  const float *shift = scale + 4;

  // staging plan (tile constants): group e = tid + 256 k -> (channel of the stage, plane, row, 16-byte group of the row)
  int voff[NK], loff[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int e = threadIdx.x + k * kThreads;
    const int cl = e / (IZ * IY * 10), r = e - cl * (IZ * IY * 10);
    const int iz = r / (IY * 10), r2 = r - iz * (IY * 10), iy = r2 / 10, g = r2 - iy * 10;
    const int gz = tz0 - 1 + iz, gy = ty0 - 1 + iy, gx = tx0 - 4 + 4 * g;
    const bool ok = e < Cfg::GROUPS && gz >= 0 && gz < Di && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;   // Wi % 4 == 0
    voff[k] = ok ? (cl * in_cs + (gz * Hi + gy) * Wi + gx) * 4 : kOOB;
    // element j of the group is column L = 4 g + j of the row (L = 0 is x0 - 4); columns 3 .. 39 are kept at [L - 3][parity]
    loff[k] = (cl >> 1) * SP + iz * SZ + iy * SY + 2 * (4 * g - 3) + (cl & 1);
  }
  f32x4v v[NK];
  auto load_stage = [&](int s) {
    const int soff = s * 2 * Cfg::CKP * in_cs * 4;
#pragma unroll
    for (int k = 0; k < NK; ++k) v[k] = buf_load4(src, voff[k], soff);   // channels >= cin lie beyond the sample: zeros
  };
  load_stage(0);
  f32x2 acc[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
  for (int s = 0; s < nstages; ++s) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      // the row's first group (g == 0) holds columns 0 .. 3: only column 3 is staged
      const int e = threadIdx.x + k * kThreads;
      const bool g0 = (e % 10) == 0;
      if (e < Cfg::GROUPS) {
        float *q = tile + loff[k];
        if (!g0) { q[0] = v[k][0]; q[2] = v[k][1]; q[4] = v[k][2]; }
        q[6] = v[k][3];
      }
    }
    __syncthreads();
    if (s + 1 < nstages) load_stage(s + 1);
#pragma unroll
    for (int pp = 0; pp < Cfg::CKP; ++pp) {
      const float *wq = wpk + (size_t)(s * Cfg::CKP + pp) * 64;   // wave-uniform: [tap][2]
#pragma unroll
      for (int r9 = 0; r9 < 9; ++r9) {
        const float *row = tile + pp * SP + (zi + r9 / 3) * SZ + (yi + r9 % 3) * SY + 8 * xi;
        const f32x4v A = *reinterpret_cast<const f32x4v *>(row), Bq = *reinterpret_cast<const f32x4v *>(row + 4),
                     Cq = *reinterpret_cast<const f32x4v *>(row + 8);
        const f32x2 P[6] = {f32x2{A[0], A[1]}, f32x2{A[2], A[3]}, f32x2{Bq[0], Bq[1]}, f32x2{Bq[2], Bq[3]}, f32x2{Cq[0], Cq[1]}, f32x2{Cq[2], Cq[3]}};
        const f32x2 W0{wq[r9 * 6 + 0], wq[r9 * 6 + 1]}, W1{wq[r9 * 6 + 2], wq[r9 * 6 + 3]}, W2{wq[r9 * 6 + 4], wq[r9 * 6 + 5]};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_elementwise_fma(P[j + 2], W2, __builtin_elementwise_fma(P[j + 1], W1, __builtin_elementwise_fma(P[j], W0, acc[j])));
      }
    }
  }
  const int oz = tz0 + zi, oy = ty0 + yi, ox = tx0 + 4 * xi;
  const rsrc_t dst = make_rsrc(out + (size_t)b * in_cs, (size_t)in_cs * 4);
  const float sc0 = scale[0], sh0 = shift[0];
  float o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = fmaf(acc[i][0] + acc[i][1], sc0, sh0);
    o[i] = o[i] > 0.0f ? o[i] : o[i] * slope;
  }
  const bool ok = oz < Di && oy < Hi && ox < Wi;
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4v{o[0], o[1], o[2], o[3]}), dst,
                                         ok ? ((oz * Hi + oy) * Wi + ox) * 4 : kOOB, 0, 0);
}

// ---- FPN top-down step: 1x1 lateral conv + bilinear x2 upsample-add (mvsnet.py:36-38, 49-50) -----------
// out = (conv1x1(x) * scale + shift) + upsample2x(up), align_corners = True.  The MFMA form of this layer
// (conv16_kernel<..., KS = 1, UPS = 1>) spends its time in the epilogue's per-element gathers: measured 172 us
// for lat0 at batch 2 against ~75 us of HBM time.  Here a thread owns 4 consecutive x of one row and walks
// the output channels: the 8 / 16 input channels of its 4 pixels stay in registers, the 4-column source
// window of the coarser map comes as one 16-byte load per row, the horizontal interpolation is a 4 x 4
// "tent" weight matrix computed once per thread (zeros for the columns a pixel does not use: adding exact
// zeros keeps ATen's lambda0 * v0 + lambda1 * v1), and every store is 16 bytes, 1 KiB contiguous per wave.
typedef float f32x4u4 __attribute__((ext_vector_type(4), aligned(4)));

template <int CIN>
__global__ __launch_bounds__(kThreads) void fpn_lateral_kernel(
    const float *__restrict__ x, const float *__restrict__ wpk, const float *__restrict__ up,
    float *__restrict__ out, int cout, int H, int W, int units, int slices) {
  extern __shared__ float smem[];
  float *wl = smem;               // [cout][CIN]
  float *sc = smem + cout * CIN;  // [cout] scale, then [cout] shift
  for (int e = threadIdx.x; e < cout * CIN; e += kThreads) {
    const int co = e / CIN, ci = e - co * CIN;  // CI-form image: lane = (ci % 4) * 16 + co % 16 of unit ci / 4, slice co / 16
    wl[e] = wpk[((size_t)(co >> 4) * units + (ci >> 2)) * 64 + (ci & 3) * 16 + (co & 15)];
  }
  const float *tail = wpk + (size_t)slices * units * 64;
  for (int e = threadIdx.x; e < cout; e += kThreads) {
    sc[e] = tail[e];
    sc[cout + e] = tail[slices * 16 + e];
  }
  __syncthreads();
  const int n = blockIdx.y;
  const int wq = W >> 2;
  const int q = blockIdx.x * kThreads + threadIdx.x;
  if (q >= H * wq) return;
  const int y = q / wq, ox = (q - y * wq) * 4;
  f32x4 c[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci)
    c[ci] = *reinterpret_cast<const f32x4 *>(x + (((size_t)n * CIN + ci) * H + y) * W + ox);
  // ATen upsample_bilinear2d, align_corners: src = dst * (in - 1) / (out - 1)
  const int hc = H >> 1, wc = W >> 1;
  const float sy = H > 1 ? (float)(hc - 1) / (float)(H - 1) : 0.0f;
  const float sx = W > 1 ? (float)(wc - 1) / (float)(W - 1) : 0.0f;
  const float fy = sy * (float)y;
  const int y0 = (int)fy, y1 = y0 + (y0 < hc - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
  int xb = 0;
  float T[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float fx = sx * (float)(ox + k);
    const int x0 = (int)fx, x1 = x0 + (x0 < wc - 1 ? 1 : 0);
    const float lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
    if (k == 0) xb = x0 < wc - 4 ? x0 : wc - 4;  // 4-column window [xb, xb + 4) holds every column the 4 pixels use
#pragma unroll
    for (int m = 0; m < 4; ++m) T[k][m] = (m == x0 - xb ? lx0 : 0.0f) + (m == x1 - xb ? lx1 : 0.0f);
  }
  const float *u0 = up + ((size_t)n * cout * hc + y0) * wc + xb;
  const float *u1 = up + ((size_t)n * cout * hc + y1) * wc + xb;
  float *op = out + ((size_t)n * cout * H + y) * W + ox;
#pragma unroll 2
  for (int co = 0; co < cout; ++co) {
    const f32x4 r0 = *reinterpret_cast<const f32x4u4 *>(u0 + (size_t)co * hc * wc);
    const f32x4 r1 = *reinterpret_cast<const f32x4u4 *>(u1 + (size_t)co * hc * wc);
    const float *wr = wl + co * CIN;
    f32x4 a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float wv = wr[ci];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = fmaf(c[ci][k], wv, a[k]);
    }
    const float scale = sc[co], shift = sc[cout + co];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float h0 = fmaf(T[k][3], r0[3], fmaf(T[k][2], r0[2], fmaf(T[k][1], r0[1], T[k][0] * r0[0])));
      const float h1 = fmaf(T[k][3], r1[3], fmaf(T[k][2], r1[2], fmaf(T[k][1], r1[1], T[k][0] * r1[0])));
      o[k] = fmaf(a[k], scale, shift) + (ly0 * h0 + ly1 * h1);
    }
    *reinterpret_cast<f32x4 *>(op + (size_t)co * H * W) = o;
  }
}

// ---- MFMA probes ----------------------------------------------------------------------------------
// Lane-mapping probe: D = A * B for A[i][k] = 1 + i + 16 k, B[k][j] = (1 + k) * (3 + j) on 16x16x4
// (dump rows 0..3 = the 4 accumulator registers), then the 4x4x1 broadcast form with ABID 0/5/15.
__global__ void mfma_probe_kernel(float *out) {
  const int lane = threadIdx.x;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const int i = lane & 15, k = lane >> 4;
  const f32x4 d16 = mfma16((float)(1 + i + 16 * k), (float)((1 + k) * (3 + i)), z);
  const float a = (float)(lane + 1), b = (float)(100 + lane);
  const f32x4 d0 = mfma_bcast<0>(a, b, z), d5 = mfma_bcast<5>(a, b, z), d15 = mfma_bcast<15>(a, b, z);
  for (int r = 0; r < 4; ++r) {
    out[(0 * 4 + r) * 64 + lane] = d16[r];
    out[(1 * 4 + r) * 64 + lane] = d0[r];
    out[(2 * 4 + r) * 64 + lane] = d5[r];
    out[(3 * 4 + r) * 64 + lane] = d15[r];
  }
}

// Issue-rate probe: every wave runs `iters` x 16 MFMAs on 8 independent accumulators.
template <int SHAPE>
__global__ __launch_bounds__(kThreads) void mfma_rate_kernel(float *out, int iters) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
  float s = 0.f;
  if constexpr (SHAPE == 0 || SHAPE == 1) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (SHAPE == 0) acc[i & 7] = mfma_bcast<1>(a, b, acc[i & 7]);
        else acc[i & 7] = mfma16(a, b, acc[i & 7]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (SHAPE == 2) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
        else acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[i & 3], 2, 1, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
  }
  if (s == 123.456f) out[0] = s;  // keep the chains live
}

// Shape 4: the conv inner loop in isolation - one ds_read_b32 (ring, 8 steps ahead) per 16x16x4 MFMA.
__global__ __launch_bounds__(kThreads) void mfma_lds_rate_kernel(float *out, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += kThreads) lds[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-6f;
  const int lane_off = (threadIdx.x & 63) + (threadIdx.x >> 6) * 1024;
  float ring[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ring[i] = lds[lane_off + i * 64];
  for (int it = 0; it < iters; ++it) {
    const int o = lane_off + ((it & 7) << 6);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float b = ring[i & 7];
      ring[i & 7] = lds[o + ((i * 37) & 511)];
      acc[i & 7] = mfma16(a, b, acc[i & 7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}

// Shapes 5..7: the same loop with the PX operand pattern (lane (j, u) reads word 2j + u), with a
// 42 KiB tile (3 workgroups per CU like the real kernel), and with random-ish data.
template <int VARIANT>
__global__ __launch_bounds__(kThreads) void mfma_lds_rate2_kernel(float *out, int iters) {
  extern __shared__ float lds2[];
  constexpr int N = VARIANT >= 6 ? 10496 : 8192;
  for (int i = threadIdx.x; i < N; i += kThreads) {
    unsigned h = (i * 2654435761u) ^ (blockIdx.x * 40503u);
    lds2[i] = VARIANT >= 7 ? (float)((int)(h >> 9) - (1 << 22)) * (1.0f / (1 << 22)) : 1.0f + i * 1e-6f;
  }
  __syncthreads();
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  float a = VARIANT >= 7 ? lds2[lane + 64] : 1.0f + threadIdx.x * 1e-6f;
  const int lane_off = 2 * (lane & 15) + (lane >> 4) + (threadIdx.x >> 6) * 1024;
  float ring[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) ring[i] = lds2[lane_off + i * 34];
  for (int it = 0; it < iters; ++it) {
    const int o = lane_off + ((it & 7) * 204);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float b = ring[i & 7];
      ring[i & 7] = lds2[o + (i & 7) * 34 + (i >> 3) * 2040];
      acc[i & 7] = mfma16(a, b, acc[i & 7]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}

// Per-kernel host-side caches, keyed by the kernel's ADDRESS (all instantiations of one kernel template have
// the same function-pointer type, so a function-local static in a template <class K> helper would be shared
// between them - it was: the first instantiation launched decided the grid of all the others).
// A/B switches exist in the profiling build only (tools/build_trace_lib.sh, -DCASMVS_TRACE): the production
// library never reads the environment.
inline int trace_env_int(const char *name, int dflt) {
#ifdef CASMVS_TRACE
  if (const char *e = getenv(name)) return atoi(e);
#else
  (void)name;
#endif
  return dflt;
}
inline bool trace_env_set(const char *name) {
#ifdef CASMVS_TRACE
  return getenv(name) != nullptr;
#else
  (void)name;
  return false;
#endif
}

template <class K>
int ensure_lds(K kernel, size_t bytes, const char *what) {
  return casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), bytes, what);
}
template <class K>
int resident_blocks(K kernel, size_t lds_bytes) {
  const int r = casmvs::resident_blocks(reinterpret_cast<const void *>(kernel), kThreads, lds_bytes);
  // profiling build: CASMVS_GRID_PCT = percentage of the resident capacity a persistent kernel launches (experiment:
  // two concurrent forwards sharing the chip side by side instead of kernel after kernel)
  static const int pct = trace_env_int("CASMVS_GRID_PCT", 100);
  return pct >= 100 ? r : (r * pct / 100 < 8 ? 8 : r * pct / 100);
}

template <int MODE, int STRIDE, int CK, int NT, int TZ, int TY, int TX, int VEC, int KZ = 3, int KS = 3, int UPS = 0, int OUT2 = 0>
int launch_conv16_v(const LayerCfg &c, const float *packed, const float *in, const float *skip,
                    float *out, int B, int cin, int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                    float slope, hipStream_t st) {
  using Cfg = Conv16Cfg<MODE, STRIDE, CK, NT, TZ, TY, TX, VEC, KZ, KS>;
  auto kernel = conv16_kernel<MODE, STRIDE, CK, NT, TZ, TY, TX, VEC, 0, KZ, KS, UPS, OUT2>;
#ifdef CASMVS_TRACE  // profiling build only (tools/build_trace_lib.sh): ablations of the PX kernel - their RESULTS ARE WRONG
  if constexpr (MODE == FMT_PX && KZ == 3) {
    static const int abl = getenv("CASMVS_ABLATE") ? atoi(getenv("CASMVS_ABLATE")) : 0;
    if (abl == 1) kernel = conv16_kernel<MODE, STRIDE, CK, NT, TZ, TY, TX, VEC, 1>;
    if (abl == 2) kernel = conv16_kernel<MODE, STRIDE, CK, NT, TZ, TY, TX, VEC, 2>;
    if (abl == 16) kernel = conv16_kernel<MODE, STRIDE, CK, NT, TZ, TY, TX, VEC, 16>;
    if (abl == 32) kernel = conv16_kernel<MODE, STRIDE, CK, NT, TZ, TY, TX, VEC, 32>;
  }
#endif
  if (int rc = ensure_lds(kernel, Cfg::LDS_BYTES, "conv16_kernel")) return rc;
  const int tiles_x = casmvs::ceil_div(Wo, TX), tiles_y = casmvs::ceil_div(Ho, TY),
            tiles_z = casmvs::ceil_div(Do, TZ);
  const long total = (long)tiles_x * tiles_y * tiles_z * B * c.slices;
  CASMVS_REQUIRE(total < (1L << 31), "conv_forward: too many tiles");
  const int resident = resident_blocks(conv16_kernel<MODE, STRIDE, CK, NT, TZ, TY, TX, VEC, 0, KZ, KS, UPS, OUT2>, Cfg::LDS_BYTES);
  dim3 grid((unsigned)(total < resident ? total : resident));
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), Cfg::LDS_BYTES, st, in, packed, skip, out, B, cin, cout,
                     Di, Hi, Wi, Do, Ho, Wo, (int)c.per_slice(), c.slices, tiles_x, tiles_y, tiles_z, slope);
  return casmvs::check_launch("conv16_kernel");
}

template <int MODE, int CK, int NT, int TZ, int TY, int TX, int STRIDE = 1, int KZ = 3, int OUT2 = 0>
int launch_conv16db(const LayerCfg &c, const float *packed, const float *in, const float *skip, float *out,
                    int B, int cin, int cout, int D, int H, int W, float slope, hipStream_t st) {
  using Cfg = Conv16DbCfg<MODE, CK, NT, TZ, TY, TX, STRIDE, KZ>;
  auto kernel = conv16db_kernel<MODE, CK, NT, TZ, TY, TX, STRIDE, KZ, OUT2>;
  if (int rc = ensure_lds(kernel, Cfg::LDS_BYTES, "conv16db_kernel")) return rc;
  // D, H, W are the INPUT dims; tiles cover the output
  const int tiles_x = casmvs::ceil_div(W / STRIDE, TX), tiles_y = casmvs::ceil_div(H / STRIDE, TY),
            tiles_z = casmvs::ceil_div(D / STRIDE, TZ);
  const long total = (long)tiles_x * tiles_y * tiles_z * B * c.slices;
  CASMVS_REQUIRE(total < (1L << 31), "conv3d_forward: too many tiles");
  const int resident = resident_blocks(kernel, Cfg::LDS_BYTES);
  dim3 grid((unsigned)(total < resident ? total : resident));
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), Cfg::LDS_BYTES, st, in, packed, skip, out, B, cin, cout,
                     D, H, W, (int)c.per_slice(), c.slices, tiles_x, tiles_y, tiles_z, slope);
  return casmvs::check_launch("conv16db_kernel");
}

// 16-byte staging needs rows that start 16-byte aligned: Wi % 4 == 0 (and 16-byte aligned tensors).
inline bool vec4_ok(const float *in, int Wi) {
  static const bool disabled = trace_env_set("CASMVS_NO_VEC4");  // A/B switch (profiling)
  return !disabled && Wi % 4 == 0 && (reinterpret_cast<size_t>(in) & 15) == 0;
}

template <int MODE, int STRIDE, int CK, int NT, int TZ, int TY, int TX>
int launch_conv16(const LayerCfg &c, const float *packed, const float *in, const float *skip,
                  float *out, int B, int cin, int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                  float slope, hipStream_t st) {
  // 16-byte staging widens the LDS rows (TX + 8): measured slower for the wide CI tile (one
  // workgroup fewer per CU) and faster for the deep one
  if constexpr (STRIDE == 1 && NT == 1) {
    if (vec4_ok(in, Wi))
      return launch_conv16_v<MODE, STRIDE, CK, NT, TZ, TY, TX, 4>(c, packed, in, skip, out, B, cin, cout, Di, Hi, Wi, Do, Ho, Wo, slope, st);
  }
  if constexpr (STRIDE == 2) {
    // 16-byte staging into x-de-interleaved rows (Stager<5>): correct and bank-conflict free, but measured
    // 10-15 % slower on conv1 / conv3 (wider rows -> one resident workgroup fewer; the per-tile set-up
    // of the flattened stager costs more VALU next to the MFMAs).  Kept for A/B runs: CASMVS_S2_VEC4=1.
    static const bool s2_vec4 = trace_env_set("CASMVS_S2_VEC4");
    if (s2_vec4 && vec4_ok(in, Wi))
      return launch_conv16_v<MODE, STRIDE, CK, NT, TZ, TY, TX, 4>(c, packed, in, skip, out, B, cin, cout, Di, Hi, Wi, Do, Ho, Wo, slope, st);
  }
  return launch_conv16_v<MODE, STRIDE, CK, NT, TZ, TY, TX, 1>(c, packed, in, skip, out, B, cin, cout, Di, Hi, Wi, Do, Ho, Wo, slope, st);
}

template <int MODE, int CK, int NT, int TZ, int TY, int TX, int VEC>
int launch_deconv16_v(const LayerCfg &c, const float *packed, const float *in, const float *skip,
                      float *out, int B, int cin, int cout, int Di, int Hi, int Wi, float slope,
                      hipStream_t st) {
  using Cfg = Deconv16Cfg<MODE, CK, NT, TZ, TY, TX, VEC>;
  auto kernel = deconv16_kernel<MODE, CK, NT, TZ, TY, TX, VEC>;
  if (int rc = ensure_lds(kernel, Cfg::LDS_BYTES, "deconv16_kernel")) return rc;
  const int tiles_x = casmvs::ceil_div(Wi, TX), tiles_y = casmvs::ceil_div(Hi, TY),
            tiles_z = casmvs::ceil_div(Di, TZ);
  dim3 grid((unsigned)(tiles_x * tiles_y * tiles_z), (unsigned)B, (unsigned)c.slices);
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), Cfg::LDS_BYTES, st, in, packed, skip, out, cin, cout,
                     Di, Hi, Wi, (int)c.per_slice(), tiles_x, tiles_y, slope);
  return casmvs::check_launch("deconv16_kernel");
}

template <int MODE, int CK, int NT, int TZ, int TY, int TX>
int launch_deconv16(const LayerCfg &c, const float *packed, const float *in, const float *skip,
                    float *out, int B, int cin, int cout, int Di, int Hi, int Wi, float slope,
                    hipStream_t st) {
  if (vec4_ok(in, Wi))
    return launch_deconv16_v<MODE, CK, NT, TZ, TY, TX, 4>(c, packed, in, skip, out, B, cin, cout, Di, Hi, Wi, slope, st);
  return launch_deconv16_v<MODE, CK, NT, TZ, TY, TX, 1>(c, packed, in, skip, out, B, cin, cout, Di, Hi, Wi, slope, st);
}

template <int VEC>
int launch_prob_v(const float *packed, const float *in, float *out, int B, int cin, int D, int H, int W,
                  float slope, hipStream_t st) {
  using Cfg = ProbCfg<4, 4, 8, 32, VEC>;
  auto kernel = prob_valu_kernel<4, 4, 8, 32, VEC>;
  if (int rc = ensure_lds(kernel, Cfg::LDS_BYTES, "prob_valu_kernel")) return rc;
  const int tiles_x = casmvs::ceil_div(W, 32), tiles_y = casmvs::ceil_div(H, 8), tiles_z = casmvs::ceil_div(D, 4);
  dim3 grid((unsigned)(tiles_x * tiles_y * tiles_z), (unsigned)B, 1);
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), Cfg::LDS_BYTES, st, in, packed, out, cin, D, H, W,
                     tiles_x, tiles_y, slope);
  return casmvs::check_launch("prob_valu_kernel");
}

int launch_prob(const LayerCfg &, const float *packed, const float *in, float *out, int B, int cin,
                int D, int H, int W, float slope, hipStream_t st) {
  // depth-walking kernel (prob_regress.hip): cin == 8, 16-byte aligned rows; the tile kernels below serve the rest
  static const bool no_zwalk = trace_env_set("CASMVS_NO_PROB_ZWALK");  // A/B switch (profiling build)
  if (!no_zwalk && casmvs_prob_regress_supported(cin, W) && vec4_ok(in, W) && (reinterpret_cast<size_t>(out) & 15) == 0)
    return casmvs_prob_regress_f32(packed, in, nullptr, out, nullptr, nullptr, nullptr, B, cin, D, H, W, slope,
                                   trace_env_int("CASMVS_PROB_ZCHUNK", 0), st);
  static const bool no_pk = trace_env_set("CASMVS_NO_PROB_PK");  // A/B switch (profiling build)
  if (!no_pk && vec4_ok(in, W)) {
    if (int rc = ensure_lds(prob_pk_kernel, ProbPkCfg::LDS_BYTES, "prob_pk_kernel")) return rc;
    const int tiles_x = casmvs::ceil_div(W, 32), tiles_y = casmvs::ceil_div(H, 8), tiles_z = casmvs::ceil_div(D, 4);
    hipLaunchKernelGGL(prob_pk_kernel, dim3((unsigned)(tiles_x * tiles_y * tiles_z), (unsigned)B, 1), dim3(kThreads), ProbPkCfg::LDS_BYTES, st,
                       in, packed, out, cin, D, H, W, tiles_x, tiles_y, slope);
    return casmvs::check_launch("prob_pk_kernel");
  }
  if (vec4_ok(in, W)) return launch_prob_v<4>(packed, in, out, B, cin, D, H, W, slope, st);
  return launch_prob_v<1>(packed, in, out, B, cin, D, H, W, slope, st);
}

}  // namespace

namespace {
inline bool is_3d_kind(int kind) { return kind >= CASMVS_CONV_S1 && kind <= CASMVS_CONV_T2; }
inline bool is_2d_kind(int kind) { return kind >= CASMVS_CONV2D_K3 && kind <= CASMVS_CONV2D_K1_UP; }

size_t packed_floats(int kind, int cin, int cout) {
  LayerCfg c;
  if (!layer_cfg(kind, cin, cout, c)) return 0;
  return (size_t)c.slices * c.per_slice() + 2 * (size_t)c.slices * c.coutb + 64;
}

int pack_layer(const char *who, int kind, int cin, int cout, const float *weight, const float *scale,
               const float *shift, float *packed) {
  CASMVS_REQUIRE(weight && packed, "%s: null pointer", who);
  LayerCfg c;
  if (!layer_cfg(kind, cin, cout, c))
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "%s: kind=%d cin=%d cout=%d", who, kind, cin, cout);
  float *p = packed;
  const int nimg = c.unit_floats / 64;
  for (int sl = 0; sl < c.slices; ++sl)
    for (int u = 0; u < c.units; ++u) {
      for (int img = 0; img < nimg; ++img)
        for (int l = 0; l < 64; ++l) *p++ = pack_weight(c, kind, cin, cout, weight, sl, u, img, l);
      for (int l = 0; l < c.unit_floats - nimg * 64; ++l) *p++ = pack_weight(c, kind, cin, cout, weight, sl, u, 0, l);  // P1 rows
    }
  const int cp = c.slices * c.coutb;
  for (int co = 0; co < cp; ++co) p[co] = (co < cout) ? (scale ? scale[co] : 1.0f) : 0.0f;
  for (int co = 0; co < cp; ++co) p[cp + co] = (co < cout) ? (shift ? shift[co] : 0.0f) : 0.0f;
  for (int i = 0; i < 64; ++i) p[2 * cp + i] = 0.0f;  // zero words the staging loads point out-of-range lanes at
  return CASMVS_OK;
}
}  // namespace

extern "C" size_t casmvs_conv3d_packed_floats(int kind, int cin, int cout) {
  return is_3d_kind(kind) ? packed_floats(kind, cin, cout) : 0;
}

extern "C" int casmvs_conv3d_pack_f32(int kind, int cin, int cout, const float *weight,
                                      const float *scale, const float *shift, float *packed) {
  casmvs::clear_error();
  if (!is_3d_kind(kind)) return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "conv3d_pack: kind=%d", kind);
  return pack_layer("conv3d_pack", kind, cin, cout, weight, scale, shift, packed);
}

// Whole CostRegNet (models/mvsnet.py:60-89): the eleven layers of casmvs_costreg_forward_f32's `packed_layers`, in that order
namespace {
struct CostRegLayer { int kind, cin, cout; };
inline void costreg_layers(int cin, CostRegLayer (&L)[11]) {
  const CostRegLayer t[11] = {{CASMVS_CONV_S1, cin, 8},  {CASMVS_CONV_S2, 8, 16},  {CASMVS_CONV_S1, 16, 16}, {CASMVS_CONV_S2, 16, 32},
                              {CASMVS_CONV_S1, 32, 32},  {CASMVS_CONV_S2, 32, 64}, {CASMVS_CONV_S1, 64, 64}, {CASMVS_CONV_T2, 64, 32},
                              {CASMVS_CONV_T2, 32, 16},  {CASMVS_CONV_T2, 16, 8},  {CASMVS_CONV_S1, 8, 1}};
  for (int i = 0; i < 11; ++i) L[i] = t[i];
}
}  // namespace

extern "C" size_t casmvs_costreg_packed_floats(int cin, size_t *layer_offsets) {
  if (cin < 1) return 0;
  CostRegLayer L[11];
  costreg_layers(cin, L);
  size_t total = 0;
  for (int i = 0; i < 11; ++i) {
    if (layer_offsets) layer_offsets[i] = total;
    total += (packed_floats(L[i].kind, L[i].cin, L[i].cout) + 3) & ~(size_t)3;   // every image starts 16-byte aligned
  }
  return total;
}

extern "C" int casmvs_costreg_pack_f32(int cin, const float *const *weights, const float *const *scales, const float *const *shifts, float *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(cin >= 1 && weights && packed, "costreg_pack: cin=%d or null pointer", cin);
  CostRegLayer L[11];
  costreg_layers(cin, L);
  size_t off[11];
  const size_t total = casmvs_costreg_packed_floats(cin, off);
  for (size_t i = 0; i < total; ++i) packed[i] = 0.0f;
  for (int i = 0; i < 11; ++i) {
    CASMVS_REQUIRE(weights[i], "costreg_pack: weights[%d] is null", i);
    if (int rc = pack_layer("costreg_pack", L[i].kind, L[i].cin, L[i].cout, weights[i], scales ? scales[i] : nullptr, shifts ? shifts[i] : nullptr, packed + off[i])) return rc;
  }
  return CASMVS_OK;
}

extern "C" size_t casmvs_conv2d_packed_floats(int kind, int cin, int cout) {
  return is_2d_kind(kind) ? packed_floats(kind, cin, cout) : 0;
}

extern "C" int casmvs_conv2d_pack_f32(int kind, int cin, int cout, const float *weight,
                                      const float *scale, const float *shift, float *packed) {
  casmvs::clear_error();
  if (!is_2d_kind(kind)) return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "conv2d_pack: kind=%d", kind);
  return pack_layer("conv2d_pack", kind, cin, cout, weight, scale, shift, packed);
}

extern "C" int casmvs_conv3d_forward_f32(int kind, const float *packed, const float *in,
                                         const float *skip, float *out, int B, int cin, int cout,
                                         int D, int H, int W, float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "conv3d_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0, "conv3d_forward: bad shape B=%d D=%d H=%d W=%d", B, D, H, W);
  LayerCfg c;
  if (!is_3d_kind(kind) || !layer_cfg(kind, cin, cout, c))
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "conv3d_forward: kind=%d cin=%d cout=%d", kind, cin, cout);
  // buffer descriptors address one sample's tensor with 31-bit byte offsets
  {
    const size_t in_floats = (size_t)cin * D * H * W;
    const size_t out_floats = (size_t)cout * D * H * W * (kind == CASMVS_CONV_T2 ? 8 : 1) / (kind == CASMVS_CONV_S2 ? 8 : 1);
    CASMVS_REQUIRE(in_floats < ((size_t)1 << 29) && out_floats < ((size_t)1 << 29),
                   "conv3d_forward: one sample's input / output tensor must hold < 2^29 floats");
  }
  hipStream_t st = (hipStream_t)stream;
  // Workgroup shapes.  "wide" variants (many column tiles per wave, small CK) maximise operand
  // reuse for the big full-resolution layers; "deep" variants (1 column tile per wave, CK = 16:
  // a 4x shorter stage chain and 4x more workgroups) serve the low-resolution layers, whose
  // cost is the latency of the chain, not FLOPs.
  if (kind == CASMVS_CONV_S1) {
    if (c.fmt == FMT_P1) {
      CASMVS_REQUIRE(skip == nullptr, "conv3d_forward: the 1-channel head takes no skip input");
      return launch_prob(c, packed, in, out, B, cin, D, H, W, slope, st);
    }
    if (c.fmt == FMT_PX) {
      static const bool no_db = trace_env_set("CASMVS_NO_DB");  // A/B switch (profiling)
      if (!no_db && W % 4 == 0 && (reinterpret_cast<size_t>(in) & 15) == 0)
        return launch_conv16db<FMT_PX, 4, 4, 4, 4, 32>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
      return launch_conv16<FMT_PX, 1, 4, 8, 8, 4, 32>(c, packed, in, skip, out, B, cin, cout, D, H, W, D, H, W, slope, st);
    }
    const long wide_blocks = (long)casmvs::ceil_div(W, 16) * casmvs::ceil_div(H, 8) * casmvs::ceil_div(D, 4) * c.slices * B;
    static const int db_ci = trace_env_int("CASMVS_DB_CI", 3);  // A/B switch (profiling); default: double-buffered
    const bool al = W % 4 == 0 && (reinterpret_cast<size_t>(in) & 15) == 0;
    if (wide_blocks >= 512 && D >= 3) {  // the wide tile is 4 deep: with D <= 2 half of every tile would be padding
      if ((db_ci & 1) && al) return launch_conv16db<FMT_CI, 4, 4, 4, 4, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
      return launch_conv16<FMT_CI, 1, 8, 8, 4, 8, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, D, H, W, slope, st);
    }
    if ((db_ci & 2) && al) return launch_conv16db<FMT_CI, 8, 1, 1, 4, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
    return launch_conv16<FMT_CI, 1, 16, 1, 1, 4, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, D, H, W, slope, st);
  }
  if (kind == CASMVS_CONV_S2) {
    CASMVS_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_forward(S2): odd input dims %dx%dx%d", D, H, W);
    const long wide_blocks = (long)casmvs::ceil_div(W / 2, 16) * casmvs::ceil_div(H / 2, 8) * casmvs::ceil_div(D / 2, 2) * c.slices * B;
    // wide tile (2, 4, 16), 2 column tiles per wave: A/B-tested against (2, 8, 16) x 4, (1, 8, 16) with CK = 8 and
    // (1, 16, 16) x 4 - the small tile wins by 7 % (5 resident workgroups per CU instead of 3)
    static const int db_s2 = trace_env_int("CASMVS_DB_S2", 3);  // A/B switch (profiling); default: double-buffered form
    const bool al2 = W % 4 == 0 && (reinterpret_cast<size_t>(in) & 15) == 0;
#ifndef CASMVS_S2_ONE_PLANE_SMALL
#define CASMVS_S2_ONE_PLANE_SMALL 1   // A/B builds: 0 = the two-plane tile also for a single output plane (rounds 1-5)
#endif
    // ONE output plane (conv5 at cascade level 0: D / 4 = 2 -> 1): the wide tile's second z plane would be computed and masked - the one-plane tile instead
    const bool one_plane = CASMVS_S2_ONE_PLANE_SMALL && D / 2 == 1;
    if (wide_blocks >= 512 && !one_plane && (db_s2 & 1) && al2) return launch_conv16db<FMT_CI, 4, 2, 2, 4, 16, 2>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
    if ((wide_blocks < 512 || one_plane) && (db_s2 & 2) && al2) return launch_conv16db<FMT_CI, 4, 1, 1, 4, 16, 2>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
    if (wide_blocks >= 512) return launch_conv16<FMT_CI, 2, 4, 2, 2, 4, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, D / 2, H / 2, W / 2, slope, st);
    return launch_conv16<FMT_CI, 2, 8, 1, 1, 4, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, D / 2, H / 2, W / 2, slope, st);
  }
  if (c.fmt == FMT_TPX) return launch_deconv16<FMT_TPX, 8, 2, 2, 4, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
  const long wide_blocks = (long)casmvs::ceil_div(W, 16) * casmvs::ceil_div(H, 8) * casmvs::ceil_div(D, 1) * c.slices * B;
  if (wide_blocks >= 512) return launch_deconv16<FMT_TCI, 8, 2, 1, 8, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
  return launch_deconv16<FMT_TCI, 16, 1, 1, 4, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
}

// ---- FeatureNet (2D) -------------------------------------------------------------------------------
// Tile shapes: TZ = 1 (the N images are the batch), 8 rows x 32 / 64 columns per workgroup.  The
// layers are small (<= 1.5 GFLOP per image), so the shapes favour many work items over operand reuse.
namespace {
// out2: optional pixel-major (N, H, W, cout) copy of the result (K3 and K1 only)
int conv2d_forward(int kind, const float *packed, const float *in, const float *up, float *out, float *out2,
                   int N, int cin, int cout, int H, int W, float slope, void *stream) {
  CASMVS_REQUIRE(packed && in && out, "conv2d_forward: null pointer");
  CASMVS_REQUIRE(N > 0 && H > 0 && W > 0, "conv2d_forward: bad shape N=%d H=%d W=%d", N, H, W);
  LayerCfg c;
  if (!is_2d_kind(kind) || !layer_cfg(kind, cin, cout, c))
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "conv2d_forward: kind=%d cin=%d cout=%d", kind, cin, cout);
  CASMVS_REQUIRE((size_t)cin * H * W < ((size_t)1 << 29) && (size_t)cout * H * W < ((size_t)1 << 29),
                 "conv2d_forward: one image's input / output tensor must hold < 2^29 floats");
  CASMVS_REQUIRE((kind == CASMVS_CONV2D_K1_UP) == (up != nullptr), "conv2d_forward: `up` goes with CASMVS_CONV2D_K1_UP only");
  CASMVS_REQUIRE(!out2 || kind == CASMVS_CONV2D_K3 || kind == CASMVS_CONV2D_K1, "conv2d_forward: pixel-major copy: K3 / K1 layers only");
  CASMVS_REQUIRE(!out2 || (reinterpret_cast<size_t>(out2) & 15) == 0, "conv2d_forward: out2 must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  switch (kind) {
    case CASMVS_CONV2D_K3: {
      // double-buffered kernel (16-byte staging: W % 4 == 0, aligned); otherwise the single-buffer one below
      // A/B-tested (3 repeats on one box): the Cout = 8 (PX) layers gain 11 % (smooth0 132 -> 117 us), the CI layers
      // nothing -> default 1 (bit 0 = PX layers, bit 1 = CI layers)
      static const int db2d = trace_env_int("CASMVS_DB_2D", 1);
      const bool al = W % 4 == 0 && (reinterpret_cast<size_t>(in) & 15) == 0;
      if (al && (db2d & 1) && c.fmt == FMT_PX) {
        if (out2) return launch_conv16db<FMT_PX, 4, 4, 1, 8, 64, 1, 1, 1>(c, packed, in, out2, out, N, cin, cout, 1, H, W, slope, st);
        return launch_conv16db<FMT_PX, 4, 4, 1, 8, 64, 1, 1, 0>(c, packed, in, nullptr, out, N, cin, cout, 1, H, W, slope, st);
      }
      if (al && (db2d & 2) && c.fmt == FMT_CI) {
        if (out2) return launch_conv16db<FMT_CI, 8, 4, 1, 8, 32, 1, 1, 1>(c, packed, in, out2, out, N, cin, cout, 1, H, W, slope, st);
        return launch_conv16db<FMT_CI, 8, 4, 1, 8, 32, 1, 1, 0>(c, packed, in, nullptr, out, N, cin, cout, 1, H, W, slope, st);
      }
      if (c.fmt == FMT_PX) {
        if (out2) return launch_conv16_v<FMT_PX, 1, 4, 4, 1, 8, 64, 1, 1, 3, 0, 1>(c, packed, in, out2, out, N, cin, cout, 1, H, W, 1, H, W, slope, st);
        return launch_conv16_v<FMT_PX, 1, 4, 4, 1, 8, 64, 1, 1, 3>(c, packed, in, nullptr, out, N, cin, cout, 1, H, W, 1, H, W, slope, st);
      }
      if (out2) return launch_conv16_v<FMT_CI, 1, 8, 4, 1, 8, 32, 1, 1, 3, 0, 1>(c, packed, in, out2, out, N, cin, cout, 1, H, W, 1, H, W, slope, st);
      return launch_conv16_v<FMT_CI, 1, 8, 4, 1, 8, 32, 1, 1, 3>(c, packed, in, nullptr, out, N, cin, cout, 1, H, W, 1, H, W, slope, st);
    }
    case CASMVS_CONV2D_K5S2:
      CASMVS_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv2d_forward(K5S2): odd input dims %dx%d", H, W);
      return launch_conv16_v<FMT_CI, 2, 4, 4, 1, 8, 32, 1, 1, 5>(c, packed, in, nullptr, out, N, cin, cout, 1, H, W, 1, H / 2, W / 2, slope, st);
    case CASMVS_CONV2D_K1:
      if (out2) return launch_conv16_v<FMT_CI, 1, 8, 4, 1, 8, 32, 1, 1, 1, 0, 1>(c, packed, in, out2, out, N, cin, cout, 1, H, W, 1, H, W, slope, st);
      return launch_conv16_v<FMT_CI, 1, 8, 4, 1, 8, 32, 1, 1, 1>(c, packed, in, nullptr, out, N, cin, cout, 1, H, W, 1, H, W, slope, st);
    default: {
      CASMVS_REQUIRE(H % 2 == 0 && W % 2 == 0, "conv2d_forward(K1_UP): odd dims %dx%d", H, W);
      static const bool no_fpn = trace_env_set("CASMVS_NO_FPN_KERNEL");  // A/B switch (profiling)
      const bool al = ((reinterpret_cast<size_t>(in) | reinterpret_cast<size_t>(out)) & 15) == 0;
      if (!no_fpn && slope == 1.0f && (cin == 8 || cin == 16) && cout <= 64 && W % 4 == 0 && W >= 8 && al && N <= 65535) {
        const size_t lds = (size_t)cout * (cin + 2) * sizeof(float);
        dim3 grid((unsigned)casmvs::ceil_div(H * (W / 4), kThreads), (unsigned)N);
        if (cin == 8)
          hipLaunchKernelGGL(fpn_lateral_kernel<8>, grid, dim3(kThreads), lds, st, in, packed, up, out, cout, H, W, c.units, c.slices);
        else
          hipLaunchKernelGGL(fpn_lateral_kernel<16>, grid, dim3(kThreads), lds, st, in, packed, up, out, cout, H, W, c.units, c.slices);
        return casmvs::check_launch("fpn_lateral_kernel");
      }
      return launch_conv16_v<FMT_CI, 1, 8, 4, 1, 8, 32, 1, 1, 1, 1>(c, packed, in, up, out, N, cin, cout, 1, H, W, 1, H, W, slope, st);
    }
  }
}
}  // namespace

extern "C" int casmvs_conv2d_forward_f32(int kind, const float *packed, const float *in, const float *up,
                                         float *out, int N, int cin, int cout, int H, int W, float slope,
                                         void *stream) {
  casmvs::clear_error();
  return conv2d_forward(kind, packed, in, up, out, nullptr, N, cin, cout, H, W, slope, stream);
}

extern "C" size_t casmvs_featurenet_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0 || H % 4 || W % 4) return 0;
  const size_t hw = (size_t)H * W;
  // full res: conv0.0 8, conv0 8, feat0' 32 | half: conv1.0, conv1.1, conv1 16 each, feat1' 32 | quarter: 3 x 32
  return (size_t)N * (48 * hw + 80 * (hw / 4) + 96 * (hw / 16)) * sizeof(float);
}

namespace {
int featurenet_run(const float *const *packed_layers, const void *fused0_packed, int fused0_arith, const float *fused0_bias9, const void *const *ci_layers,
                   const float *imgs,
                   float *feat0, float *feat1, float *feat2, float *feat0_nhwc, float *feat1_nhwc, float *feat2_nhwc,
                   void *workspace, int N, int H, int W, float slope, void *const *layer_events, void *stream) {
  CASMVS_REQUIRE(packed_layers && imgs && feat2 && workspace, "featurenet_forward: null pointer");
  // feat0 / feat1 == NULL with their pixel-major copies given: the engine's own call - nothing downstream of FeatureNet reads the (N, C, h, w) layout of
  // levels 0 / 1 (the plane sweep gathers pixel-major), so their stores are dropped (0.31 GB per batch-8 step).  feat2 feeds lat1 and is always written.
  CASMVS_REQUIRE((feat0 || feat0_nhwc) && (feat1 || feat1_nhwc), "featurenet_forward: feat0 / feat1 may be NULL only with feat0_nhwc / feat1_nhwc given");
  CASMVS_REQUIRE(N > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0,
                 "featurenet_forward: N=%d H=%d W=%d (H, W must be multiples of 4)", N, H, W);
  for (int i = 0; i < 13; ++i) CASMVS_REQUIRE(packed_layers[i], "featurenet_forward: packed_layers[%d] is null", i);
  CASMVS_REQUIRE((fused0_packed == nullptr) == (fused0_bias9 == nullptr), "featurenet_forward: fused0_packed and fused0_bias9 go together");
  CASMVS_REQUIRE(fused0_arith == 0 || fused0_arith == 1, "featurenet_forward: fused0_arith=%d (0 float32 image, 1 split-f16 image)", fused0_arith);
  const bool fuse0 = fused0_packed && casmvs_fpn_tail0_supported(H, W) && N <= 65535;
  const size_t hw = (size_t)N * H * W;
  float *ws = (float *)workspace;
  float *a0 = ws;  ws += 8 * hw;          // conv0.0
  float *c0 = ws;  ws += 8 * hw;          // conv0
  float *f0 = ws;  ws += 32 * hw;         // up(feat1') + lat0(conv0)   (unused by the fused tail)
  float *a1 = ws;  ws += 16 * (hw / 4);   // conv1.0
  float *b1 = ws;  ws += 16 * (hw / 4);   // conv1.1
  float *c1 = ws;  ws += 16 * (hw / 4);   // conv1
  float *f1 = ws;  ws += 32 * (hw / 4);   // up(feat2) + lat1(conv1)
  float *a2 = ws;  ws += 32 * (hw / 16);  // conv2.0
  float *b2 = ws;  ws += 32 * (hw / 16);  // conv2.1
  float *c2 = ws;                         // conv2
  const float *const *P = packed_layers;
  float *feat0_f32 = feat0 ? feat0 : a0;   // where a float32 layer kernel (which always stores (N, C, h, w)) puts a map the caller did not ask for:
  float *feat1_f32 = feat1 ? feat1 : a1;   // conv0.0's / conv1.0's buffers are free by then
  const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
  int rc, li = 0;
#define CASMVS_EV()                                                                            \
  if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);   \
  ++li
#define CASMVS_L(...)                                                                          \
  CASMVS_EV();                                                                                 \
  rc = conv2d_forward(__VA_ARGS__);                                                            \
  if (rc != CASMVS_OK) return rc
  if (ci_layers && ci_layers[7] && casmvs_fnet_conv0_mm_supported(W) && (reinterpret_cast<size_t>(ci_layers[7]) & 15) == 0) {   // conv0.0 + conv0.1 as one kernel on the f16 matrix cores (fnet_conv0_mm.hip)
    CASMVS_EV();   // the `conv0.0` interval of layer_events times the fused kernel, `conv0.1` is empty
    rc = casmvs_fnet_conv0_mm_f32(ci_layers[7], imgs, c0, N, H, W, slope, stream);
    if (rc != CASMVS_OK) return rc;
    CASMVS_EV();
  } else {
    CASMVS_L(CASMVS_CONV2D_K3, P[0], imgs, nullptr, a0, nullptr, N, 3, 8, H, W, slope, stream);          // conv0.0  mvsnet.py:14
    CASMVS_L(CASMVS_CONV2D_K3, P[1], a0, nullptr, c0, nullptr, N, 8, 8, H, W, slope, stream);            // conv0.1  :15
  }
  if (ci_layers && ci_layers[5] && casmvs_conv2d_k5s2_splitf16_supported(8, 16, H, W) && (reinterpret_cast<size_t>(ci_layers[5]) & 15) == 0) {   // conv1.0 on the f16 matrix cores
    CASMVS_EV();
    rc = casmvs_conv2d_k5s2_splitf16_forward_f32(ci_layers[5], c0, a1, N, 8, 16, H, W, slope, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV2D_K5S2, P[2], c0, nullptr, a1, nullptr, N, 8, 16, H, W, slope, stream);       // conv1.0  :18
  }
  if (ci_layers && ci_layers[0] && casmvs_conv2d_ci_splitf16_supported(16, 16, W2)) {   // conv1.1 on the f16 matrix cores (conv2d_ci_splitf16.hip)
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv2d_ci_splitf16_forward_f32(ci_layers[0], a1, b1, nullptr, N, 16, 16, H2, W2, slope, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV2D_K3, P[3], a1, nullptr, b1, nullptr, N, 16, 16, H2, W2, slope, stream);        // conv1.1  :19
  }
  if (ci_layers && ci_layers[1] && casmvs_conv2d_ci_splitf16_supported(16, 16, W2)) {   // conv1.2 on the f16 matrix cores (conv2d_ci_splitf16.hip)
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv2d_ci_splitf16_forward_f32(ci_layers[1], b1, c1, nullptr, N, 16, 16, H2, W2, slope, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV2D_K3, P[4], b1, nullptr, c1, nullptr, N, 16, 16, H2, W2, slope, stream);        // conv1.2  :20
  }
  if (ci_layers && ci_layers[6] && casmvs_conv2d_k5s2_splitf16_supported(16, 32, H2, W2) && (reinterpret_cast<size_t>(ci_layers[6]) & 15) == 0) {   // conv2.0 on the f16 matrix cores
    CASMVS_EV();
    rc = casmvs_conv2d_k5s2_splitf16_forward_f32(ci_layers[6], c1, a2, N, 16, 32, H2, W2, slope, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV2D_K5S2, P[5], c1, nullptr, a2, nullptr, N, 16, 32, H2, W2, slope, stream);    // conv2.0  :23
  }
  if (ci_layers && ci_layers[2] && casmvs_conv2d_ci_splitf16_supported(32, 32, W4)) {   // conv2.1 on the f16 matrix cores (conv2d_ci_splitf16.hip)
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv2d_ci_splitf16_forward_f32(ci_layers[2], a2, b2, nullptr, N, 32, 32, H4, W4, slope, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV2D_K3, P[6], a2, nullptr, b2, nullptr, N, 32, 32, H4, W4, slope, stream);        // conv2.1  :24
  }
  if (ci_layers && ci_layers[3] && casmvs_conv2d_ci_splitf16_supported(32, 32, W4)) {   // conv2.2 on the f16 matrix cores (conv2d_ci_splitf16.hip)
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv2d_ci_splitf16_forward_f32(ci_layers[3], b2, c2, nullptr, N, 32, 32, H4, W4, slope, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV2D_K3, P[7], b2, nullptr, c2, nullptr, N, 32, 32, H4, W4, slope, stream);        // conv2.2  :25
  }
  CASMVS_L(CASMVS_CONV2D_K1, P[8], c2, nullptr, feat2, feat2_nhwc, N, 32, 32, H4, W4, 1.0f, stream);      // toplayer :48
  CASMVS_L(CASMVS_CONV2D_K1_UP, P[9], c1, feat2, f1, nullptr, N, 16, 32, H2, W2, 1.0f, stream);        // lat1 + up :49
  if (fuse0) {   // lat0 + up + smooth0 in one kernel (fpn_fused.hip); the `lat0` interval of layer_events times it, `smooth0` is empty
    CASMVS_EV();
    rc = fused0_arith == 1 ? casmvs_fpn_tail0_splitf16_f32(fused0_packed, fused0_bias9, c0, f1, feat0, feat0_nhwc, N, H, W, stream)
                           : casmvs_fpn_tail0_f32(reinterpret_cast<const float *>(fused0_packed), fused0_bias9, c0, f1, feat0_f32, feat0_nhwc, N, H, W, stream);   // :50-51,54
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV2D_K1_UP, P[10], c0, f1, f0, nullptr, N, 8, 32, H, W, 1.0f, stream);           // lat0 + up :50
  }
  if (ci_layers && ci_layers[4] && casmvs_conv2d_ci_splitf16_supported(32, 16, W2) && (reinterpret_cast<size_t>(feat1_nhwc) & 15) == 0) {   // smooth1 on the f16 matrix cores
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv2d_ci_splitf16_forward_f32(ci_layers[4], f1, feat1, feat1_nhwc, N, 32, 16, H2, W2, 1.0f, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV2D_K3, P[11], f1, nullptr, feat1_f32, feat1_nhwc, N, 32, 16, H2, W2, 1.0f, stream); // smooth1  :53
  }
  if (fuse0) {
    CASMVS_EV();
  } else {
    CASMVS_L(CASMVS_CONV2D_K3, P[12], f0, nullptr, feat0_f32, feat0_nhwc, N, 32, 8, H, W, 1.0f, stream);  // smooth0  :54
  }
#undef CASMVS_L
#undef CASMVS_EV
  if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[13], (hipStream_t)stream);
  return CASMVS_OK;
}
}  // namespace

extern "C" int casmvs_featurenet_forward_f32(const float *const *packed_layers, const float *imgs,
                                             float *feat0, float *feat1, float *feat2, float *feat0_nhwc,
                                             float *feat1_nhwc, float *feat2_nhwc, void *workspace, int N,
                                             int H, int W, float slope, void *const *layer_events,
                                             void *stream) {
  casmvs::clear_error();
  return featurenet_run(packed_layers, nullptr, 0, nullptr, nullptr, imgs, feat0, feat1, feat2, feat0_nhwc, feat1_nhwc, feat2_nhwc, workspace, N, H, W,
                        slope, layer_events, stream);
}

extern "C" int casmvs_featurenet_forward_fused_f32(const float *const *packed_layers, const void *fused0_packed, int fused0_arith,
                                                   const float *fused0_bias9, const void *const *ci_layers, const float *imgs, float *feat0, float *feat1,
                                                   float *feat2, float *feat0_nhwc, float *feat1_nhwc, float *feat2_nhwc,
                                                   void *workspace, int N, int H, int W, float slope,
                                                   void *const *layer_events, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(fused0_packed && fused0_bias9, "featurenet_forward_fused: null pointer");
  return featurenet_run(packed_layers, fused0_packed, fused0_arith, fused0_bias9, ci_layers, imgs, feat0, feat1, feat2, feat0_nhwc, feat1_nhwc, feat2_nhwc, workspace,
                        N, H, W, slope, layer_events, stream);
}

extern "C" size_t casmvs_costreg_workspace_bytes(int B, int D, int h, int w) {
  if (B <= 0 || D <= 0 || h <= 0 || w <= 0 || D % 8 || h % 8 || w % 8) return 0;
  const size_t n = (size_t)D * h * w;  // full-resolution voxels
  // conv0 8n | conv1, conv2 16n/8 each | conv3, conv4 32n/64 each | conv5, conv6 64n/512 each |
  // up7 32n/64 | up9 16n/8 | up11 8n
  const size_t floats = 8 * n + 2 * (2 * n) + 2 * (n / 2) + 2 * (n / 8) + n / 2 + 2 * n + 8 * n;
  return (size_t)B * floats * sizeof(float);
}

// The middle z slice (kz = 1) of a 3D channel-inner image as the image of the 2D layer with the same (ky, kx) taps: [slice][quad][27 taps][64] ->
// [slice][quad][9 taps][64], then the scale / shift / zero tail unchanged.  A volume of ONE plane (conv6 at cascade level 0: D / 8 = 1) multiplies the taps
// kz = 0 and kz = 2 with zero padding only: the layer IS the 2D convolution with this image, at a third of the matrix instructions.
namespace {
__global__ __launch_bounds__(kThreads) void extract_kz1_image_kernel(const float *__restrict__ src, float *__restrict__ dst, int blocks, int tail) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < blocks * 576) {
    const int blk = i / 576, r = i - blk * 576;
    dst[i] = src[(size_t)blk * 27 * 64 + 9 * 64 + r];
  } else if (i < blocks * 576 + tail) {
    dst[i] = src[(size_t)blocks * 27 * 64 + (i - blocks * 576)];
  }
}
}  // namespace

#ifndef CASMVS_CONV6_2D
#define CASMVS_CONV6_2D 1   // conv6 of a one-plane volume as a 2D layer (A/B builds: 0 = the 3D kernel over its zero padding, rounds 1-5)
#endif
#ifndef CASMVS_ZFUSED_MIN_TILES
#define CASMVS_ZFUSED_MIN_TILES 180   // conv11 + prob fused from this many 16 x 60 pixel tiles on (A/B builds: a huge value = never)
#endif
#ifndef CASMVS_ZM_CIN32
#define CASMVS_ZM_CIN32 1   // conv0 at cin = 32 (cascade level 2) on conv0_zmarch.hip's warp-specialised kernel (0: the tiled kernel, A/B builds)
#endif
namespace {
// conv0 .. conv11 (+ skips) into the workspace, then the `prob` head: on its own (depth == nullptr), or fused with the
// softmax / regression / confidence that consumes it (casmvs_prob_regress_f32).
// split_layers (casmvs_costreg_regress_f32): the images of the layers that have a form on the f16 matrix cores - conv0, conv2, conv4, conv6 (stride 1),
// conv9, conv11 (transposed), conv1, conv3 (stride 2) - each or nullptr (the float32 MFMA kernel).  With conv0's split-f16 image every cin runs a z-marching kernel
// (conv0_zmarch.hip): cin = 8 / 16 on 8 x 64 patches with two workgroups per CU (1.26x / 1.5x the tiled kernel at batch 8 on the MI355X,
// profiles/r04_conv0_zm_wide_ab.txt), cin = 32 the warp-specialised form (1.15x, profiles/r04_conv0_zw_ab.txt).
int costreg_run(const char *who, const float *const *packed_layers, const void *const *split_layers, int conv0_arith, const float *vol, const float *depth_values,
                float *cost, float *depth, float *confidence, int32_t *index, void *workspace, int B, int cin, int D,
                int h, int w, float slope, void *const *layer_events, void *stream) {
  CASMVS_REQUIRE(packed_layers && vol && cost && workspace, "%s: null pointer", who);
  CASMVS_REQUIRE(B > 0 && cin > 0 && D > 0 && h > 0 && w > 0 && D % 8 == 0 && h % 8 == 0 && w % 8 == 0,
                 "%s: B=%d cin=%d D=%d h=%d w=%d (D, h, w must be multiples of 8)", who, B, cin, D, h, w);
  for (int i = 0; i < 11; ++i) CASMVS_REQUIRE(packed_layers[i], "%s: packed_layers[%d] is null", who, i);
  // every layer form - float32 MFMA and split-f16 alike - addresses one sample's tensor with 32-bit byte offsets: say so here, once, instead of from
  // whichever layer a volume past the limit reaches first (there is no form to fall back to; the batch is not limited)
  CASMVS_REQUIRE((size_t)(cin > 8 ? cin : 8) * D * h * w < ((size_t)1 << 29),
                 "%s: one sample's tensors must hold < 2^29 floats each (cin=%d D=%d h=%d w=%d: %zu); split the volume along D or the image into tiles", who, cin, D,
                 h, w, (size_t)(cin > 8 ? cin : 8) * D * h * w);
  const size_t n = (size_t)B * D * h * w;
  float *ws = (float *)workspace;
  float *c0 = ws;            ws += 8 * n;
  float *c1 = ws;            ws += 2 * n;
  float *c2 = ws;            ws += 2 * n;
  float *c3 = ws;            ws += n / 2;
  float *c4 = ws;            ws += n / 2;
  float *c5 = ws;            ws += n / 8;
  float *c6 = ws;            ws += n / 8;
  float *u7 = ws;            ws += n / 2;
  float *u9 = ws;            ws += 2 * n;
  float *u11 = ws;
  const float sl = slope;  // ABN leaky_relu slope (activation_param, 0.01 in the reference)
  const float *const *P = packed_layers;
  int rc;
  int li = 0;
#define CASMVS_L(...)                                                                          \
  if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);   \
  ++li;                                                                                        \
  rc = casmvs_conv3d_forward_f32(__VA_ARGS__);                                                 \
  if (rc != CASMVS_OK) return rc
  CASMVS_REQUIRE(conv0_arith == CASMVS_CONV0_F32 || conv0_arith == CASMVS_CONV0_SPLIT_BF16 || conv0_arith == CASMVS_CONV0_SPLIT_F16,
                 "%s: conv0_arith=%d", who, conv0_arith);
  const void *conv0_split = split_layers ? split_layers[0] : nullptr;
  const void *conv2_split = split_layers ? split_layers[1] : nullptr, *conv4_split = split_layers ? split_layers[2] : nullptr;
  const void *conv6_split = split_layers ? split_layers[3] : nullptr;
  const void *conv9_split = split_layers ? split_layers[4] : nullptr, *conv11_split = split_layers ? split_layers[5] : nullptr;
  const void *conv1_split = split_layers ? split_layers[6] : nullptr, *conv3_split = split_layers ? split_layers[7] : nullptr;
  CASMVS_REQUIRE(conv0_arith == CASMVS_CONV0_F32 || conv0_split, "%s: conv0_arith=%d needs the split image of conv0", who, conv0_arith);
  const bool split_ok = (reinterpret_cast<size_t>(vol) & 15) == 0;
  if (conv0_arith == CASMVS_CONV0_SPLIT_BF16 && split_ok && casmvs_conv0_splitbf16_supported(cin, w)) {
    // conv0 on the bf16 matrix cores, float32 operands as three exact bf16 slices (conv0_splitbf16.hip)
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv0_splitbf16_forward_f32(conv0_split, vol, c0, B, cin, D, h, w, sl, 0, stream);
    if (rc != CASMVS_OK) return rc;
  } else if (conv0_arith == CASMVS_CONV0_SPLIT_F16 && (cin == 16 || cin == 8 || (CASMVS_ZM_CIN32 && cin == 32)) && split_ok && casmvs_conv0_zmarch_supported(cin, w)) {
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv0_zmarch_forward_f32(conv0_split, vol, c0, B, cin, D, h, w, sl, stream);   // the same arithmetic, input-stationary along z
    if (rc != CASMVS_OK) return rc;
  } else if (conv0_arith == CASMVS_CONV0_SPLIT_F16 && split_ok && casmvs_conv0_splitf16_supported(cin, w)) {
    // conv0 on the f16 matrix cores, float32 operands as two scaled float16 slices (conv0_splitf16.hip)
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv0_splitf16_forward_f32(conv0_split, vol, c0, B, cin, D, h, w, sl, 0, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV_S1, P[0], vol, nullptr, c0, B, cin, 8, D, h, w, sl, stream);               // conv0
  }
  if (conv1_split && casmvs_conv_s2_splitf16_supported(8, 16, w) && (reinterpret_cast<size_t>(conv1_split) & 15) == 0) {   // conv1 on the f16 matrix cores
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv_s2_splitf16_forward_f32(conv1_split, c0, c1, B, 8, 16, D, h, w, sl, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV_S2, P[1], c0, nullptr, c1, B, 8, 16, D, h, w, sl, stream);                 // conv1
  }
  if (conv2_split && casmvs_conv_ci_splitf16_supported(16, 16, w / 2)) {                            // conv2 on the f16 matrix cores
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv_ci_splitf16_forward_f32(conv2_split, c1, c2, B, 16, 16, D / 2, h / 2, w / 2, sl, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV_S1, P[2], c1, nullptr, c2, B, 16, 16, D / 2, h / 2, w / 2, sl, stream);    // conv2
  }
  // conv3 on the f16 matrix cores where its 6 x 32 output patches fill the chip (one workgroup per CU: measured 1.13x / 1.43x the float32 kernel at
  // 264 / 880 patches, 0.97x at 96, 0.8-0.9x at 12-33: profiles/r04_conv_s2_depth_ab.txt)
  const long conv3_patches = (long)B * casmvs::ceil_div(h / 4, 6) * casmvs::ceil_div(w / 4, 32);
  if (conv3_split && casmvs_conv_s2_splitf16_supported(16, 32, w / 2) && (reinterpret_cast<size_t>(conv3_split) & 15) == 0 && conv3_patches >= 100) {
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv_s2_splitf16_forward_f32(conv3_split, c2, c3, B, 16, 32, D / 2, h / 2, w / 2, sl, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV_S2, P[3], c2, nullptr, c3, B, 16, 32, D / 2, h / 2, w / 2, sl, stream);    // conv3
  }
  // conv4 on the f16 matrix cores where there are enough of its 256-voxel tiles (4 x 4 x 16, or 2 x 8 x 16 for a 2-plane volume) to
  // fill the chip (measured: 60 tiles are faster on the float32 kernel's deep variant, 120 and more on the f16 one)
  const int tz4 = D / 4 <= 2 ? 2 : 4;
  const long conv4_tiles = (long)B * casmvs::ceil_div(D / 4, tz4) * casmvs::ceil_div(h / 4, 16 / tz4) * casmvs::ceil_div(w / 4, 16);
  if (conv4_split && casmvs_conv_ci_splitf16_supported(32, 32, w / 4) && (D / 4 == 2 || D / 4 >= 3) && conv4_tiles >= 100) {
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv_ci_splitf16_forward_f32(conv4_split, c3, c4, B, 32, 32, D / 4, h / 4, w / 4, sl, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV_S1, P[4], c3, nullptr, c4, B, 32, 32, D / 4, h / 4, w / 4, sl, stream);    // conv4
  }
  CASMVS_L(CASMVS_CONV_S2, P[5], c4, nullptr, c5, B, 32, 64, D / 4, h / 4, w / 4, sl, stream);      // conv5
  // conv6 on the f16 matrix cores where the volume has >= 3 planes and enough 4 x 4 x 16 tiles (as conv4)
  const long conv6_tiles = (long)B * casmvs::ceil_div(D / 8, 4) * casmvs::ceil_div(h / 8, 4) * casmvs::ceil_div(w / 8, 16);
  if (conv6_split && casmvs_conv_ci_splitf16_supported(64, 64, w / 8) && D / 8 >= 3 && conv6_tiles >= 100) {
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_conv_ci_splitf16_forward_f32(conv6_split, c5, c6, B, 64, 64, D / 8, h / 8, w / 8, sl, stream);
    if (rc != CASMVS_OK) return rc;
  } else if (CASMVS_CONV6_2D && D / 8 == 1 && packed_floats(CASMVS_CONV2D_K3, 64, 64) <= 8 * n) {
    // one plane (cascade level 0, D = 8): two of the three z taps multiply zero padding - the layer as the 2D convolution of its middle z slice.  The 2D
    // image is cut out of the 3D one on the device, into conv11's output buffer (free until conv11; never written at all on the fused-tail path): no new
    // operand in the ABI.  Same taps in the same order on the same float32 MFMA: the 3D kernel's result bit for bit.
    LayerCfg c3;
    (void)layer_cfg(CASMVS_CONV_S1, 64, 64, c3);
    const int blocks = c3.slices * c3.units, tail = 2 * c3.slices * c3.coutb + 64, total = blocks * 576 + tail;
    hipLaunchKernelGGL(extract_kz1_image_kernel, dim3((unsigned)casmvs::ceil_div(total, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, P[6], u11, blocks, tail);
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = conv2d_forward(CASMVS_CONV2D_K3, u11, c5, nullptr, c6, nullptr, B, 64, 64, h / 8, w / 8, sl, stream);
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV_S1, P[6], c5, nullptr, c6, B, 64, 64, D / 8, h / 8, w / 8, sl, stream);      // conv6
  }
  CASMVS_L(CASMVS_CONV_T2, P[7], c6, c4, u7, B, 64, 32, D / 8, h / 8, w / 8, sl, stream);           // conv4 + conv7
  if (conv9_split && casmvs_deconv9_splitf16_supported(w / 4) && (reinterpret_cast<size_t>(conv9_split) & 15) == 0) {
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_deconv9_splitf16_forward_f32(conv9_split, u7, c2, u9, B, D / 4, h / 4, w / 4, sl, stream);   // conv2 + conv9 on the f16 matrix cores
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV_T2, P[8], u7, c2, u9, B, 32, 16, D / 4, h / 4, w / 4, sl, stream);         // conv2 + conv9
  }
  // conv11 + `prob` + regression as ONE depth-walking kernel (conv11_prob_zfused.hip: the 8-channel tensor between them never reaches memory) where its
  // 16 x 60 pixel tiles fill the chip: measured 1.26x / 1.27x / 1.12x the two kernels at 768 / 2816 / 192 tiles (cascade levels 1 / 0 / 2, batch 8), 1.05x at 352,
  // 0.7x at 96 (profiles/r04_conv11_prob_zfused_first_run.txt).  The `conv11` interval of layer_events is then empty, `prob` times the fused kernel.
  const long zf_tiles = (long)B * casmvs::ceil_div(h, 16) * casmvs::ceil_div(w, 60);
  constexpr long zf_min_tiles = CASMVS_ZFUSED_MIN_TILES;
  if (conv11_split && depth != nullptr && (reinterpret_cast<size_t>(conv11_split) & 15) == 0 && zf_tiles >= zf_min_tiles &&
      casmvs_conv11_prob_zfused_supported(D / 2, h / 2, w / 2)) {
    if (layer_events) {
      (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);       // conv11
      (void)hipEventRecord((hipEvent_t)layer_events[li + 1], (hipStream_t)stream);   // prob
    }
    rc = casmvs_conv11_prob_zfused_f32(conv11_split, P[10], u9, c0, depth_values, cost, depth, confidence, index, B, D / 2, h / 2, w / 2, sl, 1.0f, stream);
    if (rc != CASMVS_OK) return rc;
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[11], (hipStream_t)stream);
    return CASMVS_OK;
  }
  if (conv11_split && casmvs_deconv11_splitf16_supported(w / 2) && (reinterpret_cast<size_t>(conv11_split) & 15) == 0) {
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    ++li;
    rc = casmvs_deconv11_splitf16_forward_f32(conv11_split, u9, c0, u11, B, D / 2, h / 2, w / 2, sl, stream);   // conv0 + conv11 on the f16 matrix cores
    if (rc != CASMVS_OK) return rc;
  } else {
    CASMVS_L(CASMVS_CONV_T2, P[9], u9, c0, u11, B, 16, 8, D / 2, h / 2, w / 2, sl, stream);         // conv0 + conv11
  }
  if (depth == nullptr) {
    CASMVS_L(CASMVS_CONV_S1, P[10], u11, nullptr, cost, B, 8, 1, D, h, w, 1.0f, stream);            // prob
  } else {
    if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[li], (hipStream_t)stream);
    rc = casmvs_prob_regress_f32(P[10], u11, depth_values, cost, depth, confidence, index, B, 8, D, h, w, 1.0f,
                                 trace_env_int("CASMVS_PROB_ZCHUNK", 0), stream);                   // prob + softmax regression
    if (rc != CASMVS_OK) return rc;
  }
#undef CASMVS_L
  if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[11], (hipStream_t)stream);
  return CASMVS_OK;
}
}  // namespace

extern "C" int casmvs_costreg_forward_f32(const float *const *packed_layers, const float *vol,
                                          float *cost, void *workspace, int B, int cin, int D,
                                          int h, int w, float slope, void *const *layer_events,
                                          void *stream) {
  casmvs::clear_error();
  return costreg_run("costreg_forward", packed_layers, nullptr, CASMVS_CONV0_F32, vol, nullptr, cost, nullptr, nullptr, nullptr, workspace, B, cin, D, h, w,
                     slope, layer_events, stream);
}

extern "C" int casmvs_costreg_regress_f32(const float *const *packed_layers, const void *const *split_layers, int conv0_arith, const float *vol,
                                          const float *depth_values, float *cost, float *depth, float *confidence, int32_t *index,
                                          void *workspace, int B, int cin, int D, int h, int w, float slope,
                                          void *const *layer_events, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(depth_values && depth && confidence, "costreg_regress: null pointer");
  return costreg_run("costreg_regress", packed_layers, split_layers, conv0_arith, vol, depth_values, cost, depth, confidence, index, workspace, B,
                     cin, D, h, w, slope, layer_events, stream);
}

extern "C" int casmvs_selftest_mfma_rate(int shape, int blocks, int iters, float *tflops) {
  casmvs::clear_error();
  CASMVS_REQUIRE(shape >= 0 && shape <= 7 && blocks > 0 && iters > 0 && tflops, "selftest_mfma_rate: bad arguments");
  float *d = nullptr;
  hipEvent_t e0, e1;
  if (hipMalloc(&d, 64) != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "selftest_mfma_rate: hipMalloc failed");
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto launch = [&](int n) {
    switch (shape) {
      case 0: hipLaunchKernelGGL(mfma_rate_kernel<0>, dim3(blocks), dim3(kThreads), 0, 0, d, n); break;
      case 1: hipLaunchKernelGGL(mfma_rate_kernel<1>, dim3(blocks), dim3(kThreads), 0, 0, d, n); break;
      case 2: hipLaunchKernelGGL(mfma_rate_kernel<2>, dim3(blocks), dim3(kThreads), 0, 0, d, n); break;
      case 3: hipLaunchKernelGGL(mfma_rate_kernel<3>, dim3(blocks), dim3(kThreads), 0, 0, d, n); break;
      case 4: hipLaunchKernelGGL(mfma_lds_rate_kernel, dim3(blocks), dim3(kThreads), 0, 0, d, n); break;
      case 5: hipLaunchKernelGGL(mfma_lds_rate2_kernel<5>, dim3(blocks), dim3(kThreads), 8192 * 4, 0, d, n / 2); break;
      case 6: hipLaunchKernelGGL(mfma_lds_rate2_kernel<6>, dim3(blocks), dim3(kThreads), 10496 * 4, 0, d, n / 2); break;
      default: hipLaunchKernelGGL(mfma_lds_rate2_kernel<7>, dim3(blocks), dim3(kThreads), 10496 * 4, 0, d, n / 2); break;
    }
  };
  launch(16);  // warm-up
  (void)hipEventRecord(e0, 0);
  launch(iters);
  (void)hipEventRecord(e1, 0);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  if (e != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "selftest_mfma_rate: %s", hipGetErrorString(e));
  const double flop_per_mfma[8] = {512.0, 2048.0, 4096.0, 2048.0, 2048.0, 2048.0, 2048.0, 2048.0};  // 4x4x1_16b, 16x16x4, 32x32x2, 16x16x1_4b, 16x16x4 + ds_read
  const double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 * flop_per_mfma[shape];
  *tflops = (float)(flops / (ms * 1e-3) / 1e12);
  return CASMVS_OK;
}

extern "C" int casmvs_selftest_mfma(float *dump) {
  casmvs::clear_error();
  float *d = nullptr;
  if (hipMalloc(&d, 16 * 64 * sizeof(float)) != hipSuccess)
    return casmvs::fail(CASMVS_ERR_HIP, "selftest: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, 0, d);
  float h[16 * 64];
  hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "selftest: %s", hipGetErrorString(e));
  if (dump)
    for (int i = 0; i < 16 * 64; ++i) dump[i] = h[i];
  // 16x16x4: lane l reg r holds D[row = 4*(l>>4) + r][col = l&15], D = A * B
  for (int r = 0; r < 4; ++r)
    for (int l = 0; l < 64; ++l) {
      const int row = 4 * (l >> 4) + r, col = l & 15;
      float want = 0.f;
      for (int k = 0; k < 4; ++k) want += (float)(1 + row + 16 * k) * (float)((1 + k) * (3 + col));
      const float got = h[(0 * 4 + r) * 64 + l];
      if (got != want)
        return casmvs::fail(CASMVS_ERR_HIP, "selftest: mfma_16x16x4 reg=%d lane=%d: got %g want %g", r, l, got, want);
    }
  const int abids[3] = {0, 5, 15};
  for (int k = 0; k < 3; ++k)
    for (int r = 0; r < 4; ++r)
      for (int l = 0; l < 64; ++l) {
        const float want = (float)(4 * abids[k] + r + 1) * (float)(100 + l);
        const float got = h[((k + 1) * 4 + r) * 64 + l];
        if (got != want)
          return casmvs::fail(CASMVS_ERR_HIP, "selftest: mfma_4x4x1 cbsz=4 abid=%d reg=%d lane=%d: got %g want %g", abids[k], r, l, got, want);
      }
  return CASMVS_OK;
}

#ifdef CASMVS_TRACE
extern "C" int casmvs_trace_read(unsigned long long *host, int clear) {
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * 64 * 128) != hipSuccess) return -3;
  if (clear) {
    static unsigned long long z[64 * 128];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace), z, sizeof(z)) != hipSuccess) return -3;
  }
  return 0;
}
#endif
