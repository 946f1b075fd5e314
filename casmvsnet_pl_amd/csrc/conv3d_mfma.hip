// CostRegNet 3D convolutions on the gfx950 matrix cores, fp32 in / fp32 accumulate.
//
// Reference semantics: models/mvsnet.py:60-104 (CostRegNet), models/modules.py:21-31
// (ConvBnReLU3D): Conv3d / ConvTranspose3d (k3, p1) -> eval-mode ABN -> (+ skip).
//
// Formulation.  The channel counts are tiny (Cout in {1, 8, 16, 32, 64}), so the usual
// "voxels x Cout" 16x16 / 32x32 MFMA tiles would waste half of the matrix pipe on the
// full-resolution layers (Cout = 8) that hold most of the FLOPs.  Instead every layer runs on
//     v_mfma_f32_4x4x1_16b_f32  with  CBSZ = 4 (broadcast ONE A block to all 16 blocks):
//         D[co 0..3][voxel = lane] += A[co 0..3] * B[voxel = lane]          (K = 1)
// i.e. each LANE owns one output voxel, the 4 accumulator registers are 4 output channels, the B
// operand is the (tap-shifted) input value of the lane's voxel and the A operand is a 4-float
// weight column that ABID picks out of a 64-lane VGPR holding 16 such columns.  512 FLOP per
// 8-cycle instruction = the full fp32 MFMA rate with zero padding waste for any Cout % 4 == 0,
// coalesced NCDHW loads/stores (a wavefront = 64 consecutive voxels of a tile), and the
// epilogue (folded ABN, leaky-relu, skip add) is a plain per-lane FMA.
//
// Data flow per workgroup (256 threads = 4 wavefronts, G groups of 64 voxels per wavefront):
//   for each chunk of CK input channels:  stage the zero-padded halo tile of the chunk in LDS
//     for each of the 27 taps:  A images (prefetched one tap ahead, straight from L1/L2 - every
//       workgroup streams the same few KB), G*CK conflict-free ds_read_b32 B operands,
//       G*CK*Q MFMAs (Q = Cout-per-block / 4).
// Cout > 16 is split into 16-channel slices over blockIdx.z.
//
// Packed parameter image (built on the host by casmvs_conv3d_pack_f32):
//   [slice][stage][tap 0..26][j 0..NV-1][64 lanes]  then  scale[slices*COUTB], shift[slices*COUTB],
//   then 64 zero floats (target of out-of-range staging loads)
//   image (stage, tap, j), lane l: n = 16*j + l/4, cil = n / Q, q = n % Q, i = l % 4
//   holds w[co = slice*COUTB + 4q + i][ci = stage*CK + cil][tap]  (0 outside cin/cout).
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;

template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// One K = 1 step for 4 output channels x 64 voxels: acc[r] (lane) += a[4*ABID + r] * b[lane].
template <int ABID>
__device__ __forceinline__ f32x4 mfma_bcast(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, /*cbsz=*/4, /*abid=*/ABID, /*blgp=*/0);
}

// Same with a loop-variable ABID: every call site sits in a fully unrolled loop, so the switch
// folds to the single matching instruction.
__device__ __forceinline__ f32x4 mfma_sel(int abid, float a, float b, f32x4 c) {
  switch (abid) {
    case 0: return mfma_bcast<0>(a, b, c);
    case 1: return mfma_bcast<1>(a, b, c);
    case 2: return mfma_bcast<2>(a, b, c);
    case 3: return mfma_bcast<3>(a, b, c);
    case 4: return mfma_bcast<4>(a, b, c);
    case 5: return mfma_bcast<5>(a, b, c);
    case 6: return mfma_bcast<6>(a, b, c);
    case 7: return mfma_bcast<7>(a, b, c);
    case 8: return mfma_bcast<8>(a, b, c);
    case 9: return mfma_bcast<9>(a, b, c);
    case 10: return mfma_bcast<10>(a, b, c);
    case 11: return mfma_bcast<11>(a, b, c);
    case 12: return mfma_bcast<12>(a, b, c);
    case 13: return mfma_bcast<13>(a, b, c);
    case 14: return mfma_bcast<14>(a, b, c);
    default: return mfma_bcast<15>(a, b, c);
  }
}

struct LayerCfg {
  int coutb;   // output channels per block (multiple of 4)
  int ck;      // input channels per LDS stage
  int slices;  // ceil(cout / coutb)
  int nv;      // A images (64-lane VGPRs) per (stage, tap)
  int nstages;
};

inline bool layer_cfg(int kind, int cin, int cout, LayerCfg &c) {
  if (cin < 1 || cout < 1) return false;
  if (kind == CASMVS_CONV_S1) {
    if (cout == 1) c.coutb = 4;
    else if (cout == 8) c.coutb = 8;
    else if (cout % 16 == 0) c.coutb = 16;
    else return false;
    c.ck = 8;
  } else if (kind == CASMVS_CONV_S2) {
    if (cout % 16 != 0) return false;
    c.coutb = 16;
    c.ck = 4;
  } else if (kind == CASMVS_CONV_T2) {
    if (cout == 8) c.coutb = 8;
    else if (cout % 16 == 0) c.coutb = 16;
    else return false;
    c.ck = 8;
  } else {
    return false;
  }
  c.slices = (cout + c.coutb - 1) / c.coutb;
  c.nv = (c.ck * (c.coutb / 4) + 15) / 16;
  c.nstages = (cin + c.ck - 1) / c.ck;
  return true;
}


// ---- staging: global -> registers -> LDS, software-pipelined one chunk ahead -------------------
// A chunk = CK input channels of the zero-padded halo tile (CK*IZ planes of IY*IX floats) plus
// the chunk's 27*NV weight images.  A thread copies the same NPASS in-plane positions of every
// plane, so the (iy, ix) decode, the bounds tests and the in-plane global offset are computed
// ONCE per kernel (StagePlan); a plane then costs one add + load per position.  The loads of
// chunk s+1 are issued right after the barrier that publishes chunk s and land in registers
// while the MFMA loop of chunk s runs (5-6 us of cover for ~2 us of L2/HBM latency); they are
// written to LDS after the next barrier.  Weights go through LDS too, so that the MFMA loop
// contains no vector-memory instruction (an in-loop global load would make the compiler's
// in-order vmcnt wait drain the whole prefetch at the first tap).
template <int IY, int IX>
struct StagePlan {
  static constexpr int PLANE = IY * IX;
  static constexpr int NPASS = (PLANE + kThreads - 1) / kThreads;
  int goff[NPASS];   // gy * Wi + gx of this thread's position (valid only if inb)
  bool inb[NPASS];   // position inside the image in y and x
  __device__ __forceinline__ void init(int iy0, int ix0, int Hi, int Wi) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int pe = threadIdx.x + p * kThreads;
      const int iy = pe / IX, ix = pe - iy * IX;
      const int gy = iy0 + iy, gx = ix0 + ix;
      inb[p] = pe < PLANE && gy >= 0 && gy < Hi && gx >= 0 && gx < Wi;
      goff[p] = gy * Wi + gx;
    }
  }
};

template <int CK, int IZ, int IY, int IX, int NW>
struct StageRegs {
  static constexpr int NPL = CK * IZ, NPASS = StagePlan<IY, IX>::NPASS;
  static constexpr int NWR = (NW + kThreads - 1) / kThreads;
  float v[NPL][NPASS];
  float w[NWR];

  // issue every load of chunk `ci0 / CK`; nothing here waits
  __device__ __forceinline__ void load(const StagePlan<IY, IX> &plan, const float *__restrict__ inb,
                                       size_t in_cs, int cin, int ci0, int iz0, int Di, int HiWi,
                                       const float *__restrict__ wchunk, const float *__restrict__ zero) {
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
      // branch-free and mask-free: out-of-range positions load from a zero word that the packed
      // parameter image carries at its end, so no predicate has to survive until the data lands.
      // 32-bit element offsets (one sample's input is < 2^31 floats).
      const int poff = ci * (int)in_cs + gz * HiWi;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const bool ok = plane_ok && plan.inb[p];
        const float *ptr = ok ? inb + (poff + plan.goff[p]) : zero;
        v[pl][p] = *ptr;
      }
    }
  }

  __device__ __forceinline__ void store(float *tile, float *wts) const {
    constexpr int PLANE = IY * IX;
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int e = threadIdx.x + i * kThreads;
      if (e < NW) wts[e] = w[i];
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int pe = threadIdx.x + p * kThreads;
        if (pe < PLANE) tile[pl * PLANE + pe] = v[pl][p];
      }
  }
};

// ---- Conv3d k3 p1, stride 1 or 2 -------------------------------------------------------------
template <int STRIDE, int COUTB, int CK, int G, int TZ, int TY, int TX>
struct ConvCfg {
  static constexpr int Q = COUTB / 4;
  static constexpr int NV = (CK * Q + 15) / 16;
  static constexpr int IZ = STRIDE * (TZ - 1) + 3, IY = STRIDE * (TY - 1) + 3, IX = STRIDE * (TX - 1) + 3;
  static constexpr int SY = IX, SZ = IY * SY, SC = IZ * SZ;
  static constexpr int NW = 27 * NV * 64;                       // weight floats per chunk
  static constexpr size_t LDS_BYTES = (size_t)(CK * SC + NW) * sizeof(float);
};

template <int STRIDE, int COUTB, int CK, int G, int TZ, int TY, int TX>
__global__ __launch_bounds__(kThreads) void conv3d_kernel(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ skip,
    float *__restrict__ out, int cin, int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
    int nstages, int tiles_x, int tiles_y, float slope) {
  static_assert(TZ * TY * TX == 4 * G * 64, "tile must hold 4 waves x G groups x 64 voxels");
  using Cfg = ConvCfg<STRIDE, COUTB, CK, G, TZ, TY, TX>;
  constexpr int Q = Cfg::Q, NV = Cfg::NV, IZ = Cfg::IZ, IY = Cfg::IY, IX = Cfg::IX;
  constexpr int SY = Cfg::SY, SZ = Cfg::SZ, SC = Cfg::SC, NW = Cfg::NW;
  extern __shared__ float smem[];
  float *tile = smem;            // [CK][IZ][IY][IX]
  float *wts = smem + CK * SC;   // [27][NV][64]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bid = blockIdx.x;
  const int tx0 = (bid % tiles_x) * TX;
  const int ty0 = ((bid / tiles_x) % tiles_y) * TY;
  const int tz0 = (bid / (tiles_x * tiles_y)) * TZ;
  const int b = blockIdx.y, slice = blockIdx.z;
  const int slices = gridDim.z;

  int base[G], vx[G], vy[G], vz[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int v = (wave * G + g) * 64 + lane;
    vx[g] = v % TX;
    vy[g] = (v / TX) % TY;
    vz[g] = v / (TX * TY);
    base[g] = vz[g] * STRIDE * SZ + vy[g] * STRIDE * SY + vx[g] * STRIDE;
  }
  f32x4 acc[G][Q];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[g][q] = f32x4{0.f, 0.f, 0.f, 0.f};

  const size_t in_cs = (size_t)Di * Hi * Wi;  // input channel stride
  const float *inb = in + (size_t)b * cin * in_cs;
  const int iz0 = tz0 * STRIDE - 1, iy0 = ty0 * STRIDE - 1, ix0 = tx0 * STRIDE - 1;
  const int T = nstages * 27;
  const float *wslice = wpk + (size_t)slice * T * NV * 64;
  const float *zero = wpk + (size_t)slices * T * NV * 64 + 2 * slices * COUTB;  // 64 zero floats

  // MFMA loop of one chunk, written out in issue order and pinned with sched_barrier: within a
  // tap, item i = (c, g) needs one B operand (ds_read_b32, immediate offset c * SC) and feeds Q
  // MFMAs.  The read of item i + P is issued right before the MFMAs of item i (wrapping into the
  // next tap), so P items (>= 128 MFMA cycles) of LDS latency are always covered and the waitcnt
  // pass emits counted lgkmcnt waits instead of draining after every read.  The A images of the
  // next tap are read from LDS a whole tap ahead.
  constexpr int NI = CK * G;                // items per tap
  constexpr int P = NI < 8 ? NI : 8;        // read-ahead distance (items)
  auto tap_off = [&](int tap) -> int { return (tap / 9) * SZ + ((tap / 3) % 3) * SY + (tap % 3); };

  StagePlan<IY, IX> plan;
  plan.init(iy0, ix0, Hi, Wi);
  StageRegs<CK, IZ, IY, IX, NW> regs;
  regs.load(plan, inb, in_cs, cin, 0, iz0, Di, Hi * Wi, wslice, zero);
  for (int s = 0; s < nstages; ++s) {
    __syncthreads();  // every wave is done reading the previous chunk
    regs.store(tile, wts);
    __syncthreads();
    if (s + 1 < nstages)  // prefetch the next chunk; consumed after the next barrier
      regs.load(plan, inb, in_cs, cin, (s + 1) * CK, iz0, Di, Hi * Wi, wslice + (size_t)(s + 1) * NW, zero);

    float a_cur[NV], a_nxt[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) a_cur[j] = wts[j * 64 + lane];
    int ad_c[G], ad_n[G];  // per-group LDS word address of the current / next tap
#pragma unroll
    for (int g = 0; g < G; ++g) ad_c[g] = base[g];  // tap 0: offset 0
    float ring[P];
#pragma unroll
    for (int i = 0; i < P; ++i) ring[i] = tile[ad_c[i % G] + (i / G) * SC];
    for (int tap = 0; tap < 27; ++tap) {
      const int tapn = tap < 26 ? tap + 1 : 26;  // last tap: harmless re-read
#pragma unroll
      for (int j = 0; j < NV; ++j) a_nxt[j] = wts[(tapn * NV + j) * 64 + lane];
      const int toff_n = tap_off(tapn);
#pragma unroll
      for (int g = 0; g < G; ++g) ad_n[g] = base[g] + toff_n;
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int c = i / G, g = i % G;
        const float bcur = ring[i % P];
        const int ii = i + P;
        if (ii < NI) ring[i % P] = tile[ad_c[ii % G] + (ii / G) * SC];
        else ring[i % P] = tile[ad_n[(ii - NI) % G] + ((ii - NI) / G) * SC];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int n = c * Q + q;
          acc[g][q] = mfma_sel(n % 16, a_cur[n / 16], bcur, acc[g][q]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) a_cur[j] = a_nxt[j];
#pragma unroll
      for (int g = 0; g < G; ++g) ad_c[g] = ad_n[g];
    }
  }

  // epilogue: y = lrelu(acc * scale + shift) (+ skip)
  const float *scale = wpk + (size_t)slices * T * NV * 64 + slice * COUTB;
  const float *shift = scale + slices * COUTB;
  const size_t out_cs = (size_t)Do * Ho * Wo;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int oz = tz0 + vz[g], oy = ty0 + vy[g], ox = tx0 + vx[g];
    if (oz >= Do || oy >= Ho || ox >= Wo) continue;
    const size_t vo = ((size_t)oz * Ho + oy) * Wo + ox;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 4 * q + r, co = slice * COUTB + col;
        if (co < cout) {
          float v = fmaf(acc[g][q][r], scale[col], shift[col]);
          v = v > 0.0f ? v : v * slope;
          const size_t o = ((size_t)b * cout + co) * out_cs + vo;
          if (skip) v += skip[o];
          out[o] = v;
        }
      }
  }
}

// ---- ConvTranspose3d k3 s2 p1 op1 --------------------------------------------------------------
// out[2i - 1 + k] += in[i] * w[k] per axis: even outputs o = 2m take (k = 1, i = m); odd outputs
// o = 2m + 1 take (k = 2, i = m) and (k = 0, i = m + 1).  Each lane owns one input cell m and
// produces the two x-parities of output row (2mz + pz, 2my + py); (pz, py) comes from blockIdx.
template <int COUTB, int CK, int TZ, int TY, int TX>
struct DeconvCfg {
  static constexpr int Q = COUTB / 4;
  static constexpr int NV = (CK * Q + 15) / 16;
  static constexpr int IZ = TZ + 1, IY = TY + 1, IX = TX + 1;
  static constexpr int SY = IX, SZ = IY * SY, SC = IZ * SZ;
  static constexpr int NW = 27 * NV * 64;
  static constexpr size_t LDS_BYTES = (size_t)(CK * SC + NW) * sizeof(float);
};

template <int COUTB, int CK, int TZ, int TY, int TX>
__global__ __launch_bounds__(kThreads) void deconv3d_kernel(
    const float *__restrict__ in, const float *__restrict__ wpk, const float *__restrict__ skip,
    float *__restrict__ out, int cin, int cout, int Di, int Hi, int Wi, int nstages, int tiles_x,
    int tiles_y, int ntiles, float slope) {
  static_assert(TZ * TY * TX == 256, "tile must hold 4 waves x 64 cells");
  using Cfg = DeconvCfg<COUTB, CK, TZ, TY, TX>;
  constexpr int Q = Cfg::Q, NV = Cfg::NV, IZ = Cfg::IZ, IY = Cfg::IY, IX = Cfg::IX;
  constexpr int SY = Cfg::SY, SZ = Cfg::SZ, SC = Cfg::SC, NW = Cfg::NW;
  extern __shared__ float smem[];
  float *tile = smem;
  float *wts = smem + CK * SC;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pass = blockIdx.x / ntiles, bid = blockIdx.x - pass * ntiles;
  const int pz = pass >> 1, py = pass & 1;
  const int tx0 = (bid % tiles_x) * TX;
  const int ty0 = ((bid / tiles_x) % tiles_y) * TY;
  const int tz0 = (bid / (tiles_x * tiles_y)) * TZ;
  const int b = blockIdx.y, slice = blockIdx.z;
  const int slices = gridDim.z;

  const int v = wave * 64 + lane;
  const int vx = v % TX, vy = (v / TX) % TY, vz = v / (TX * TY);
  const int base = vz * SZ + vy * SY + vx;

  f32x4 acc0[Q], acc1[Q];  // x parity 0 / 1
#pragma unroll
  for (int q = 0; q < Q; ++q) acc0[q] = acc1[q] = f32x4{0.f, 0.f, 0.f, 0.f};

  const size_t in_cs = (size_t)Di * Hi * Wi;
  const float *inb = in + (size_t)b * cin * in_cs;
  const int T = nstages * 27;
  const float *wslice = wpk + (size_t)slice * T * NV * 64;
  const float *zero = wpk + (size_t)slices * T * NV * 64 + 2 * slices * COUTB;  // 64 zero floats
  const int nzt = pz ? 2 : 1, nyt = py ? 2 : 1;
  StagePlan<IY, IX> plan;
  plan.init(ty0, tx0, Hi, Wi);
  StageRegs<CK, IZ, IY, IX, NW> regs;
  regs.load(plan, inb, in_cs, cin, 0, tz0, Di, Hi * Wi, wslice, zero);

  for (int s = 0; s < nstages; ++s) {
    __syncthreads();
    regs.store(tile, wts);
    __syncthreads();
    if (s + 1 < nstages)
      regs.load(plan, inb, in_cs, cin, (s + 1) * CK, tz0, Di, Hi * Wi, wslice + (size_t)(s + 1) * NW, zero);
    for (int zt = 0; zt < nzt; ++zt) {
      const int kz = pz ? (zt == 0 ? 2 : 0) : 1, dz = (pz && zt == 1) ? 1 : 0;
      for (int yt = 0; yt < nyt; ++yt) {
        const int ky = py ? (yt == 0 ? 2 : 0) : 1, dy = (py && yt == 1) ? 1 : 0;
        const int tap0 = (kz * 3 + ky) * 3;  // kx = 0, 1, 2 follow
        float a0[NV], a1[NV], a2[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          a0[j] = wts[((tap0 + 0) * NV + j) * 64 + lane];
          a1[j] = wts[((tap0 + 1) * NV + j) * 64 + lane];
          a2[j] = wts[((tap0 + 2) * NV + j) * 64 + lane];
        }
        const int toff = dz * SZ + dy * SY;
        float b0[CK], b1[CK];
#pragma unroll
        for (int c = 0; c < CK; ++c) {
          b0[c] = tile[c * SC + base + toff];
          b1[c] = tile[c * SC + base + toff + 1];
        }
        static_for<CK>([&](auto c_) {
          constexpr int c = decltype(c_)::value;
          static_for<Q>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            constexpr int n = c * Q + q;
            acc0[q] = mfma_bcast<n % 16>(a1[n / 16], b0[c], acc0[q]);  // px = 0: k = 1, i = m
            acc1[q] = mfma_bcast<n % 16>(a2[n / 16], b0[c], acc1[q]);  // px = 1: k = 2, i = m
            acc1[q] = mfma_bcast<n % 16>(a0[n / 16], b1[c], acc1[q]);  // px = 1: k = 0, i = m + 1
          });
        });
      }
    }
  }

  const float *scale = wpk + (size_t)slices * T * NV * 64 + slice * COUTB;
  const float *shift = scale + slices * COUTB;
  const int mz = tz0 + vz, my = ty0 + vy, mx = tx0 + vx;
  if (mz >= Di || my >= Hi || mx >= Wi) return;
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const size_t out_cs = (size_t)Do * Ho * Wo;
  const size_t vo = ((size_t)(2 * mz + pz) * Ho + (2 * my + py)) * Wo + 2 * mx;
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = 4 * q + r, co = slice * COUTB + col;
      if (co < cout) {
        float v0 = fmaf(acc0[q][r], scale[col], shift[col]);
        float v1 = fmaf(acc1[q][r], scale[col], shift[col]);
        v0 = v0 > 0.0f ? v0 : v0 * slope;
        v1 = v1 > 0.0f ? v1 : v1 * slope;
        const size_t o = ((size_t)b * cout + co) * out_cs + vo;
        if (skip) {
          const f32x2 sk = *reinterpret_cast<const f32x2 *>(skip + o);
          v0 += sk[0];
          v1 += sk[1];
        }
        *reinterpret_cast<f32x2 *>(out + o) = f32x2{v0, v1};
      }
    }
}

// ---- MFMA lane-mapping probe ---------------------------------------------------------------------
__global__ void mfma_probe_kernel(float *out) {
  const int lane = threadIdx.x;
  const float a = (float)(lane + 1), b = (float)(100 + lane);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 d0 = mfma_bcast<0>(a, b, z), d5 = mfma_bcast<5>(a, b, z), d15 = mfma_bcast<15>(a, b, z);
  f32x4 dn = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) {
    out[(0 * 4 + r) * 64 + lane] = d0[r];
    out[(1 * 4 + r) * 64 + lane] = d5[r];
    out[(2 * 4 + r) * 64 + lane] = d15[r];
    out[(3 * 4 + r) * 64 + lane] = dn[r];
  }
}

// Kernels that need more than the default 64 KiB of LDS must opt in once per process.
template <class K>
int ensure_lds(K kernel, size_t bytes, const char *what) {
  static bool done = false;  // one instance per kernel instantiation (template)
  if (!done && bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "%s: hipFuncSetAttribute(%zu B LDS): %s", what, bytes, hipGetErrorString(e));
  }
  done = true;
  return CASMVS_OK;
}

template <int STRIDE, int COUTB, int CK, int G, int TZ, int TY, int TX>
int launch_conv(const LayerCfg &c, const float *packed, const float *in, const float *skip,
                float *out, int B, int cin, int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                float slope, hipStream_t st) {
  using Cfg = ConvCfg<STRIDE, COUTB, CK, G, TZ, TY, TX>;
  auto kernel = conv3d_kernel<STRIDE, COUTB, CK, G, TZ, TY, TX>;
  if (int rc = ensure_lds(kernel, Cfg::LDS_BYTES, "conv3d_kernel")) return rc;
  const int tiles_x = casmvs::ceil_div(Wo, TX), tiles_y = casmvs::ceil_div(Ho, TY),
            tiles_z = casmvs::ceil_div(Do, TZ);
  dim3 grid((unsigned)(tiles_x * tiles_y * tiles_z), (unsigned)B, (unsigned)c.slices);
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), Cfg::LDS_BYTES, st, in, packed, skip, out, cin, cout,
                     Di, Hi, Wi, Do, Ho, Wo, c.nstages, tiles_x, tiles_y, slope);
  return casmvs::check_launch("conv3d_kernel");
}

template <int COUTB, int CK, int TZ, int TY, int TX>
int launch_deconv(const LayerCfg &c, const float *packed, const float *in, const float *skip,
                  float *out, int B, int cin, int cout, int Di, int Hi, int Wi, float slope,
                  hipStream_t st) {
  using Cfg = DeconvCfg<COUTB, CK, TZ, TY, TX>;
  auto kernel = deconv3d_kernel<COUTB, CK, TZ, TY, TX>;
  if (int rc = ensure_lds(kernel, Cfg::LDS_BYTES, "deconv3d_kernel")) return rc;
  const int tiles_x = casmvs::ceil_div(Wi, TX), tiles_y = casmvs::ceil_div(Hi, TY),
            tiles_z = casmvs::ceil_div(Di, TZ);
  const int ntiles = tiles_x * tiles_y * tiles_z;
  dim3 grid((unsigned)(4 * ntiles), (unsigned)B, (unsigned)c.slices);
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), Cfg::LDS_BYTES, st, in, packed, skip, out, cin, cout,
                     Di, Hi, Wi, c.nstages, tiles_x, tiles_y, ntiles, slope);
  return casmvs::check_launch("deconv3d_kernel");
}

}  // namespace

extern "C" size_t casmvs_conv3d_packed_floats(int kind, int cin, int cout) {
  LayerCfg c;
  if (!layer_cfg(kind, cin, cout, c)) return 0;
  return (size_t)c.slices * c.nstages * 27 * c.nv * 64 + 2 * (size_t)c.slices * c.coutb + 64;
}

extern "C" int casmvs_conv3d_pack_f32(int kind, int cin, int cout, const float *weight,
                                      const float *scale, const float *shift, float *packed) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && packed, "conv3d_pack: null pointer");
  LayerCfg c;
  if (!layer_cfg(kind, cin, cout, c))
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "conv3d_pack: kind=%d cin=%d cout=%d", kind, cin, cout);
  const int Q = c.coutb / 4;
  float *p = packed;
  for (int sl = 0; sl < c.slices; ++sl)
    for (int s = 0; s < c.nstages; ++s)
      for (int tap = 0; tap < 27; ++tap)
        for (int j = 0; j < c.nv; ++j)
          for (int l = 0; l < 64; ++l) {
            const int n = 16 * j + l / 4, i = l % 4;
            const int cil = n / Q, q = n % Q;
            const int ci = s * c.ck + cil, co = sl * c.coutb + 4 * q + i;
            float w = 0.0f;
            if (cil < c.ck && ci < cin && co < cout) {
              w = (kind == CASMVS_CONV_T2) ? weight[((size_t)ci * cout + co) * 27 + tap]
                                           : weight[((size_t)co * cin + ci) * 27 + tap];
            }
            *p++ = w;
          }
  const int cp = c.slices * c.coutb;
  for (int co = 0; co < cp; ++co) p[co] = (co < cout) ? (scale ? scale[co] : 1.0f) : 0.0f;
  for (int co = 0; co < cp; ++co) p[cp + co] = (co < cout) ? (shift ? shift[co] : 0.0f) : 0.0f;
  for (int i = 0; i < 64; ++i) p[2 * cp + i] = 0.0f;  // zero words the staging loads point out-of-range lanes at
  return CASMVS_OK;
}

extern "C" int casmvs_conv3d_forward_f32(int kind, const float *packed, const float *in,
                                         const float *skip, float *out, int B, int cin, int cout,
                                         int D, int H, int W, float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed && in && out, "conv3d_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && B <= 65535 && D > 0 && H > 0 && W > 0, "conv3d_forward: bad shape B=%d D=%d H=%d W=%d", B, D, H, W);
  LayerCfg c;
  if (!layer_cfg(kind, cin, cout, c))
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "conv3d_forward: kind=%d cin=%d cout=%d", kind, cin, cout);
  hipStream_t st = (hipStream_t)stream;
  if (kind == CASMVS_CONV_S1) {
    if (c.coutb == 4) return launch_conv<1, 4, 8, 4, 8, 4, 32>(c, packed, in, skip, out, B, cin, cout, D, H, W, D, H, W, slope, st);
    if (c.coutb == 8) return launch_conv<1, 8, 8, 4, 8, 4, 32>(c, packed, in, skip, out, B, cin, cout, D, H, W, D, H, W, slope, st);
    // coutb == 16: large volumes use 512-voxel tiles, small (deep) volumes 256-voxel tiles
    const long big_blocks = (long)casmvs::ceil_div(W, 16) * casmvs::ceil_div(H, 8) * casmvs::ceil_div(D, 4) * c.slices * B;
    if (big_blocks >= 1024) return launch_conv<1, 16, 8, 2, 4, 8, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, D, H, W, slope, st);
    return launch_conv<1, 16, 8, 1, 1, 16, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, D, H, W, slope, st);
  }
  if (kind == CASMVS_CONV_S2) {
    CASMVS_REQUIRE(D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "conv3d_forward(S2): odd input dims %dx%dx%d", D, H, W);
    return launch_conv<2, 16, 4, 1, 2, 8, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, D / 2, H / 2, W / 2, slope, st);
  }
  if (c.coutb == 8) return launch_deconv<8, 8, 2, 8, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
  return launch_deconv<16, 8, 2, 8, 16>(c, packed, in, skip, out, B, cin, cout, D, H, W, slope, st);
}

extern "C" size_t casmvs_costreg_workspace_bytes(int B, int D, int h, int w) {
  if (B <= 0 || D <= 0 || h <= 0 || w <= 0 || D % 8 || h % 8 || w % 8) return 0;
  const size_t n = (size_t)D * h * w;  // full-resolution voxels
  // conv0 8n | conv1, conv2 16n/8 each | conv3, conv4 32n/64 each | conv5, conv6 64n/512 each |
  // up7 32n/64 | up9 16n/8 | up11 8n
  const size_t floats = 8 * n + 2 * (2 * n) + 2 * (n / 2) + 2 * (n / 8) + n / 2 + 2 * n + 8 * n;
  return (size_t)B * floats * sizeof(float);
}

extern "C" int casmvs_costreg_forward_f32(const float *const *packed_layers, const float *vol,
                                          float *cost, void *workspace, int B, int cin, int D,
                                          int h, int w, float slope, void *const *layer_events,
                                          void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(packed_layers && vol && cost && workspace, "costreg_forward: null pointer");
  CASMVS_REQUIRE(B > 0 && cin > 0 && D > 0 && h > 0 && w > 0 && D % 8 == 0 && h % 8 == 0 && w % 8 == 0,
                 "costreg_forward: B=%d cin=%d D=%d h=%d w=%d (D, h, w must be multiples of 8)", B, cin, D, h, w);
  for (int i = 0; i < 11; ++i) CASMVS_REQUIRE(packed_layers[i], "costreg_forward: packed_layers[%d] is null", i);
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
  CASMVS_L(CASMVS_CONV_S1, P[0], vol, nullptr, c0, B, cin, 8, D, h, w, sl, stream);                 // conv0
  CASMVS_L(CASMVS_CONV_S2, P[1], c0, nullptr, c1, B, 8, 16, D, h, w, sl, stream);                   // conv1
  CASMVS_L(CASMVS_CONV_S1, P[2], c1, nullptr, c2, B, 16, 16, D / 2, h / 2, w / 2, sl, stream);      // conv2
  CASMVS_L(CASMVS_CONV_S2, P[3], c2, nullptr, c3, B, 16, 32, D / 2, h / 2, w / 2, sl, stream);      // conv3
  CASMVS_L(CASMVS_CONV_S1, P[4], c3, nullptr, c4, B, 32, 32, D / 4, h / 4, w / 4, sl, stream);      // conv4
  CASMVS_L(CASMVS_CONV_S2, P[5], c4, nullptr, c5, B, 32, 64, D / 4, h / 4, w / 4, sl, stream);      // conv5
  CASMVS_L(CASMVS_CONV_S1, P[6], c5, nullptr, c6, B, 64, 64, D / 8, h / 8, w / 8, sl, stream);      // conv6
  CASMVS_L(CASMVS_CONV_T2, P[7], c6, c4, u7, B, 64, 32, D / 8, h / 8, w / 8, sl, stream);           // conv4 + conv7
  CASMVS_L(CASMVS_CONV_T2, P[8], u7, c2, u9, B, 32, 16, D / 4, h / 4, w / 4, sl, stream);           // conv2 + conv9
  CASMVS_L(CASMVS_CONV_T2, P[9], u9, c0, u11, B, 16, 8, D / 2, h / 2, w / 2, sl, stream);           // conv0 + conv11
  CASMVS_L(CASMVS_CONV_S1, P[10], u11, nullptr, cost, B, 8, 1, D, h, w, 1.0f, stream);              // prob
#undef CASMVS_L
  if (layer_events) (void)hipEventRecord((hipEvent_t)layer_events[11], (hipStream_t)stream);
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
  const int abids[3] = {0, 5, 15};
  for (int k = 0; k < 3; ++k)
    for (int r = 0; r < 4; ++r)
      for (int l = 0; l < 64; ++l) {
        const float want = (float)(4 * abids[k] + r + 1) * (float)(100 + l);
        const float got = h[(k * 4 + r) * 64 + l];
        if (got != want)
          return casmvs::fail(CASMVS_ERR_HIP, "selftest: mfma_4x4x1 cbsz=4 abid=%d reg=%d lane=%d: got %g want %g", abids[k], r, l, got, want);
      }
  return CASMVS_OK;
}
