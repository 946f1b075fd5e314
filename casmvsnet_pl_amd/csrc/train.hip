// Training-side kernels: what `train.py` needs beyond the inference engine.
//
//   conv_wgrad_kernel            weight gradient of every convolution kind of the model (Conv2d k3 / k5s2 / k1, Conv3d k3
//                                s1 / s2, ConvTranspose3d k3 s2) on the matrix cores: torch's conv backward w.r.t. `weight`
//                                (models/modules.py:13,26, models/mvsnet.py:36-38,74-89)
//   conv_dgrad_direct_kernel     input gradient of the layer shapes the forward MFMA kernels cannot express as another
//                                forward layer (Conv2d k5s2, the 8-channel 1x1 lateral); every other input gradient IS a
//                                forward launch with adjoint weights (casmvsnet_pl_amd/training.py)
//   channel_sums / abn_*         train-mode ABN = BatchNorm with batch statistics + leaky ReLU (inplace_abn.ABN /
//                                InPlaceABN as used by modules.py:14,27 and mvsnet.py:77,82,87), forward and backward
//   upsample2x_add_*             the FPN top-down step F.interpolate(x2, bilinear, align_corners=True) + lateral
//                                (mvsnet.py:36-38), forward and backward
//   costvol_var_bwd_kernel       gradient of the variance cost volume (mvsnet.py:137-167) w.r.t. the feature maps
//
// Everything is fp32; reductions that decide parameters (weight gradients, batch statistics) are accumulated per
// workgroup and summed in a fixed order (double for the statistics): results are run-to-run reproducible.
#include "common.h"
#include "plane_sweep.h"
#include "split_f16.h"
#include "fixed_accum.h"

namespace {

using namespace casmvs_dev;

constexpr int kThreads = 256;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---- weight gradient -------------------------------------------------------------------------------------------------
// Every kind is G[cs][cb][t] = sum over (b, o) of small[b][cs][o] * big[b][cb][S o - P + t]   (per axis; t = kernel tap):
//   Conv (stride S, pad P):        small = grad_out (cs = cout), big = input (cb = cin)          -> (cout, cin, taps)
//   ConvTranspose3d (k3 s2 p1 op1): small = input (cs = cin),   big = grad_out (cb = cout), S = 2 -> (cin, cout, taps)
// i.e. a GEMM with the positions o as the contraction: D[16 cs x 16 cb] += A[16 cs x 4 o] B[4 o x 16 cb] per tap.
// A workgroup owns one (16 cs, 16 cb) pair and walks tiles of 4 x 16 positions of the small grid; its four waves take four
// consecutive k-steps (4 positions each) of a tile row and keep one accumulator per tap.  No atomics: every wave
// writes its partial sums, wgrad_reduce_kernel adds them in a fixed order.
// (Rounds 3 / 4: operand tiles with channel strides = 2 mod 32 make the matrix phase's ds_read_b32 conflict-free on the CPU bank model - the odd strides
// are 2-way conflicted - and give the same bits, but the whole training step did not move on the MI355X: 14.10 against 14.17 ms, inside the
// run-to-run noise (profiles/r04_bench_train_variants.txt): the kernel is not bound by those reads.  Removed rather than kept as an option.)
template <int S, int KZ, int KS>
struct WgradCfg {
  static constexpr int T = KZ * KS * KS;
  // tile of the small grid: TZ x 4 x 16 positions.  The stride-1 3D layers (the bulk of the FLOPs) take four planes per
  // tile: the kz halo is then 6 planes per 4 instead of 3 per 1 (staging bytes per position: 2.5x instead of 5x)
  static constexpr int TZ = (KZ == 3 && S == 1) ? 4 : 1, TY = 4, TX = 16;
  static constexpr int ROWS = TZ * TY, NPOS = ROWS * TX;   // (z, y) rows of 16 positions; a wave owns ROWS / 4 of them
  static constexpr int PZ = KZ / 2, P = KS / 2;
  static constexpr int IZ = (TZ - 1) * S + KZ, IY = (TY - 1) * S + KS, IX = (TX - 1) * S + KS;
  static constexpr int SY = IX, SZ = IY * IX;
  static constexpr int SC = (IZ * SZ) | 1;   // odd channel stride: the 16 cb lanes of a B operand hit 16 different banks
  static constexpr int SS = NPOS + 1;             // row stride of the small tile
  static constexpr int TILE_FLOATS = 16 * SC + 16 * SS, RED_FLOATS = T * 256;   // the end-of-kernel reduction reuses the buffer
  static constexpr size_t LDS_BYTES = (size_t)(TILE_FLOATS > RED_FLOATS ? TILE_FLOATS : RED_FLOATS) * sizeof(float);
};

template <int S, int KZ, int KS, bool VEC>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_kernel(const float *__restrict__ small, const float *__restrict__ big,
                                                             float *__restrict__ partial, int B, int Cs, int Cb, int Zs,
                                                             int Ys, int Xs, int cb_groups, int tiles_z, int tiles_y, int tiles_x) {
  using Cfg = WgradCfg<S, KZ, KS>;
  constexpr int T = Cfg::T, IZ = Cfg::IZ, IY = Cfg::IY, IX = Cfg::IX, SY = Cfg::SY, SZ = Cfg::SZ, SC = Cfg::SC, SS = Cfg::SS;
  constexpr int TZ = Cfg::TZ, TY = Cfg::TY, NPOS = Cfg::NPOS, RPW = Cfg::ROWS / 4;
  CASMVS_DYNAMIC_LDS(float, smem);
  float *bigT = smem;               // [16 cb][IZ][IY][IX]
  float *smallT = smem + 16 * SC;   // [16 cs][NPOS positions]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, kq = lane >> 4;
  const int rt = blockIdx.y / cb_groups, cg = blockIdx.y - rt * cb_groups;
  const int Zb = KZ == 1 ? 1 : Zs * S, Yb = Ys * S, Xb = Xs * S;
  const size_t small_cs = (size_t)Zs * Ys * Xs, big_cs = (size_t)Zb * Yb * Xb;
  const int tiles = B * tiles_z * tiles_y * tiles_x;
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    int r = tile;
    const int tx = r % tiles_x; r /= tiles_x;
    const int ty = r % tiles_y; r /= tiles_y;
    const int tz = r % tiles_z, b = r / tiles_z;
    const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * Cfg::TX;
    __syncthreads();   // the previous tile's operands are no longer read
    const int bz0 = (KZ == 1 ? 0 : oz0 * S) - Cfg::PZ, by0 = oy0 * S - Cfg::P, bx0 = ox0 * S - Cfg::P;
    if constexpr (VEC) {
      // Staging in 16-byte units (grid widths % 4 == 0, 16-byte aligned tensors of at most 2^31 bytes): a tile row of the
      // small grid is 4 aligned vectors; a row of the big tile is P halo floats, 4 S aligned vectors, KS - P - S halo floats.
      // Buffer loads: a unit outside the grid / beyond the channel count carries the offset kOOB and the hardware returns
      // zeros (the padding).  All loads of a tile are in flight before the first LDS write: one memory round trip, and ~6x
      // fewer index computations than element-wise staging (which cost as many issue cycles as the tile's MFMAs).
      constexpr int kOOB = (int)0x80000000u;
      constexpr int ROWS = Cfg::ROWS, P = Cfg::P;
      constexpr int SU = 16 * ROWS * 4, NSU = (SU + kThreads - 1) / kThreads;
      constexpr int RB = 16 * IZ * IY, VPR = 4 * S, BU = RB * VPR, NBU = (BU + kThreads - 1) / kThreads;
      constexpr int NH = KS - S, HU = RB * NH, NHU = (HU + kThreads - 1) / kThreads;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(small), 0, (int)((size_t)B * Cs * small_cs * 4), 0x00020000);
      const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(big), 0, (int)((size_t)B * Cb * big_cs * 4), 0x00020000);
      f32x4 sv[NSU], bv[NBU];
      float hv[NHU > 0 ? NHU : 1];
#pragma unroll
      for (int i = 0; i < NSU; ++i) {
        const int u = (int)threadIdx.x + i * kThreads;
        const int c = u / (ROWS * 4), r = (u >> 2) % ROWS, k = u & 3;
        const int oz = oz0 + r / TY, oy = oy0 + r % TY, ox = ox0 + 4 * k, ch = rt * 16 + c;
        const bool ok = u < SU && ch < Cs && oz < Zs && oy < Ys && ox < Xs;
        sv[i] = buf_load4(rs, ok ? ((((b * Cs + ch) * Zs + oz) * Ys + oy) * Xs + ox) * 4 : kOOB, 0);
      }
      if constexpr (NH > 0) {
#pragma unroll
        for (int i = 0; i < NHU; ++i) {
          const int u = (int)threadIdx.x + i * kThreads;
          const int row = u / NH, h = u - row * NH, c = row / (IZ * IY), rem = row - c * (IZ * IY), iz = rem / IY, iy = rem - iz * IY;
          const int ix = h < P ? h : 16 * S + h;   // left halo 0 .. P-1, right halo P + 16 S ...
          const int gz = bz0 + iz, gy = by0 + iy, gx = bx0 + ix, ch = cg * 16 + c;
          const bool ok = u < HU && ch < Cb && gz >= 0 && gz < Zb && gy >= 0 && gy < Yb && gx >= 0 && gx < Xb;
          hv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, ok ? ((((b * Cb + ch) * Zb + gz) * Yb + gy) * Xb + gx) * 4 : kOOB, 0, 0));
        }
      }
#pragma unroll
      for (int i = 0; i < NSU; ++i) {
        const int u = (int)threadIdx.x + i * kThreads;
        if (u < SU) {
          float *dst = smallT + (u / (ROWS * 4)) * SS + ((u >> 2) % ROWS) * 16 + 4 * (u & 3);
          dst[0] = sv[i][0]; dst[1] = sv[i][1]; dst[2] = sv[i][2]; dst[3] = sv[i][3];
        }
      }
      // the big tile's vectors in batches of at most 10 units per thread (40 registers in flight; 2 x 7 for the 14 of the stride-2 3D form)
      constexpr int UBV = NBU <= 10 ? 10 : 7;
#pragma unroll
      for (int i0 = 0; i0 < NBU; i0 += UBV) {
#pragma unroll
        for (int i = i0; i < (i0 + UBV < NBU ? i0 + UBV : NBU); ++i) {
          const int u = (int)threadIdx.x + i * kThreads;
          const int row = u / VPR, k = u - row * VPR, c = row / (IZ * IY), rem = row - c * (IZ * IY), iz = rem / IY, iy = rem - iz * IY;
          const int gz = bz0 + iz, gy = by0 + iy, gx = bx0 + P + 4 * k, ch = cg * 16 + c;
          const bool ok = u < BU && ch < Cb && gz >= 0 && gz < Zb && gy >= 0 && gy < Yb && gx < Xb;
          bv[i] = buf_load4(rb, ok ? ((((b * Cb + ch) * Zb + gz) * Yb + gy) * Xb + gx) * 4 : kOOB, 0);
        }
#pragma unroll
        for (int i = i0; i < (i0 + UBV < NBU ? i0 + UBV : NBU); ++i) {
          const int u = (int)threadIdx.x + i * kThreads;
          if (u < BU) {
            const int row = u / VPR, k = u - row * VPR, c = row / (IZ * IY), rem = row - c * (IZ * IY), iz = rem / IY, iy = rem - iz * IY;
            float *dst = bigT + c * SC + iz * SZ + iy * SY + P + 4 * k;
            dst[0] = bv[i][0]; dst[1] = bv[i][1]; dst[2] = bv[i][2]; dst[3] = bv[i][3];
          }
        }
      }
      if constexpr (NH > 0) {
#pragma unroll
        for (int i = 0; i < NHU; ++i) {
          const int u = (int)threadIdx.x + i * kThreads;
          if (u < HU) {
            const int row = u / NH, h = u - row * NH, c = row / (IZ * IY), rem = row - c * (IZ * IY), iz = rem / IY, iy = rem - iz * IY;
            bigT[c * SC + iz * SZ + iy * SY + (h < P ? h : 16 * S + h)] = hv[i];
          }
        }
      }
    } else {
      // Scalar staging (any width / alignment) in batches of 8 elements per thread: all 8 loads are issued (from clamped, always
      // valid addresses; the value is zeroed by a select when the element lies outside the grid / beyond the channel count)
      // before the first LDS write.
      constexpr int UB = 8;
      // small tile: 16 channels x NPOS positions (q = (z * TY + y) * 16 + x)
      for (int e0 = 0; e0 < 16 * NPOS; e0 += UB * kThreads) {
        float v[UB];
  #pragma unroll
        for (int i = 0; i < UB; ++i) {
          const int e = min(e0 + threadIdx.x + i * kThreads, 16 * NPOS - 1);
          const int c = e / NPOS, q = e - c * NPOS, oz = oz0 + q / (TY * 16), oy = oy0 + (q >> 4) % TY, ox = ox0 + (q & 15), ch = rt * 16 + c;
          const bool ok = ch < Cs && oz < Zs && oy < Ys && ox < Xs;
          const float l = small[((size_t)b * Cs + min(ch, Cs - 1)) * small_cs + ((size_t)min(oz, Zs - 1) * Ys + min(oy, Ys - 1)) * Xs + min(ox, Xs - 1)];
          v[i] = ok ? l : 0.0f;
        }
  #pragma unroll
        for (int i = 0; i < UB; ++i) {
          const int e = e0 + threadIdx.x + i * kThreads;
          if (e < 16 * NPOS) smallT[(e / NPOS) * SS + (e % NPOS)] = v[i];
        }
      }
      // big tile: 16 channels x the footprint of the tile's taps, zero padding outside the grid / beyond Cb
      constexpr int NBIG = 16 * IZ * IY * IX;
      for (int e0 = 0; e0 < NBIG; e0 += UB * kThreads) {
        float v[UB];
        int lo[UB];
  #pragma unroll
        for (int i = 0; i < UB; ++i) {
          const int e = e0 + threadIdx.x + i * kThreads;
          const int ec = min(e, NBIG - 1);
          const int c = ec / (IZ * IY * IX), rem = ec - c * (IZ * IY * IX);
          const int iz = rem / (IY * IX), rem2 = rem - iz * (IY * IX), iy = rem2 / IX, ix = rem2 - iy * IX;
          const int gz = bz0 + iz, gy = by0 + iy, gx = bx0 + ix, ch = cg * 16 + c;
          const bool ok = ch < Cb && gz >= 0 && gz < Zb && gy >= 0 && gy < Yb && gx >= 0 && gx < Xb;
          const float l = big[((size_t)b * Cb + min(ch, Cb - 1)) * big_cs + ((size_t)min(max(gz, 0), Zb - 1) * Yb + min(max(gy, 0), Yb - 1)) * Xb + min(max(gx, 0), Xb - 1)];
          v[i] = ok ? l : 0.0f;
          lo[i] = e < NBIG ? c * SC + iz * SZ + iy * SY + ix : -1;
        }
  #pragma unroll
        for (int i = 0; i < UB; ++i)
          if (lo[i] >= 0) bigT[lo[i]] = v[i];
      }
    }
    __syncthreads();
    // this wave's rows (z, y) of the tile, four k-steps (positions ox = 4 ks + kq) each
#pragma unroll 1
    for (int rr = 0; rr < RPW; ++rr) {   // rolled: unrolled, the 16 x T operand loads of a wave were all hoisted (368 registers)
      const int row = wave * RPW + rr, rz = row / TY, ry = row % TY;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const float a = smallT[i16 * SS + row * 16 + ks * 4 + kq];
        const float *bp = bigT + i16 * SC + (rz * S) * SZ + (ry * S) * SY + (ks * 4 + kq) * S;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int tz_ = t / (KS * KS), ty_ = (t / KS) % KS, tx_ = t % KS;
          acc[t] = mfma16(a, bp[tz_ * SZ + ty_ * SY + tx_], acc[t]);
        }
      }
    }
  }
  // The four waves' accumulators are added through LDS in wave order (deterministic), then the workgroup writes ONE partial:
  // partial[blockIdx.x * gridDim.y + blockIdx.y][t][cs 16][cb 16]; D row = 4 kq + r, column = i16
  __syncthreads();
  float *red = smem;   // T * 256 floats (LDS_BYTES covers it)
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float *q = red + t * 256 + (4 * kq + r) * 16 + i16;
          *q = wv == 0 ? acc[t][r] : *q + acc[t][r];
        }
    }
    __syncthreads();
  }
  float *pp = partial + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * (size_t)(T * 256);
  for (int e = threadIdx.x; e < T * 256; e += kThreads) pp[e] = red[e];
}

// grad_weight[cs][cb][t] = sum over the workgroups' partials in a fixed order.  A block owns 32 consecutive elements of the
// partial layout (128-byte coalesced reads) and splits the up to 512 partials into 8 slices, one per 32 threads: a thread
// walks its slice with eight loads in flight (eight running sums, combined in a fixed order), thread 0..31 then add the
// slices in slice order - run-to-run reproducible.  (One thread per element walking all 512 partials took 72 us per layer,
// 3.3 ms of a 35 ms training step, for 7 MB of reads.)
constexpr int kRedElems = 32, kRedSlices = kThreads / kRedElems;
__global__ __launch_bounds__(kThreads) void wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ gw, int Cs, int Cb,
                                                               int T, int cb_groups, int gy, int gx) {
  __shared__ float red[kRedSlices][kRedElems];
  const int le = threadIdx.x & (kRedElems - 1), sl = threadIdx.x / kRedElems;
  const int e = blockIdx.x * kRedElems + le;   // (y, t, i, j); gy * T * 256 is a multiple of 32: no ragged block
  const size_t stride = (size_t)gy * (T * 256);
  const float *p = partial + e;
  const int per = (gx + kRedSlices - 1) / kRedSlices, lo = sl * per, hi = min(lo + per, gx);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int x = lo;
  for (; x + 8 <= hi; x += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(x + k) * stride];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
  for (; x < hi; ++x) acc[(x - lo) & 7] += p[(size_t)x * stride];
  red[sl][le] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (sl != 0) return;
  float total = red[0][le];
#pragma unroll
  for (int q = 1; q < kRedSlices; ++q) total += red[q][le];
  const int j = e & 15, i = (e >> 4) & 15, t = (e >> 8) % T, y = (e >> 8) / T;
  const int cs = (y / cb_groups) * 16 + i, cb = (y % cb_groups) * 16 + j;
  if (cs < Cs && cb < Cb) gw[((size_t)cs * Cb + cb) * T + t] = total;
}

// ---- direct input gradient (layer shapes without an adjoint forward kernel) ------------------------------------------
// grad_in[b][ci][p] = sum over (co, t) with S o - P + t = p of grad_out[b][co][o] * w[co][ci][t]   (Conv2d / Conv3d weights)
__global__ __launch_bounds__(kThreads) void conv_dgrad_direct_kernel(const float *__restrict__ gout, const float *__restrict__ w,
                                                                    float *__restrict__ gin, int cin, int cout, int Zi, int Yi,
                                                                    int Xi, int S, int KZ, int KS) {
  const int b = blockIdx.z, ci = blockIdx.y;
  const size_t in_cs = (size_t)Zi * Yi * Xi;
  const size_t p = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (p >= in_cs) return;
  const int x = (int)(p % Xi), y = (int)((p / Xi) % Yi), z = (int)(p / ((size_t)Xi * Yi));
  const int Sz = KZ == 1 ? 1 : S;   // 2D layers: one plane
  const int Zo = Zi / Sz, Yo = Yi / S, Xo = Xi / S, PZ = KZ / 2, P = KS / 2, T = KZ * KS * KS;
  const size_t out_cs = (size_t)Zo * Yo * Xo;
  float acc = 0.0f;
  for (int tz = 0; tz < KZ; ++tz) {
    const int nz = z + PZ - tz;
    if (nz < 0 || nz % Sz || nz / Sz >= Zo) continue;
    for (int ty = 0; ty < KS; ++ty) {
      const int ny = y + P - ty;
      if (ny < 0 || ny % S || ny / S >= Yo) continue;
      for (int tx = 0; tx < KS; ++tx) {
        const int nx = x + P - tx;
        if (nx < 0 || nx % S || nx / S >= Xo) continue;
        const size_t o = ((size_t)(nz / Sz) * Yo + ny / S) * Xo + nx / S;
        const int t = (tz * KS + ty) * KS + tx;
        for (int co = 0; co < cout; ++co)
          acc = fmaf(gout[((size_t)b * cout + co) * out_cs + o], w[((size_t)co * cin + ci) * T + t], acc);
      }
    }
  }
  gin[((size_t)b * cin + ci) * in_cs + p] = acc;
}

// The same sum for ALL CIN input channels of a position in one thread: a grad_out value is loaded once and used CIN times (the
// per-channel form above re-read it per channel and spent its time in the tap loop's divergent `continue`s: 370 us per layer).
// The taps that reach an input position are enumerated directly (t = t0, t0 + S, ...: t = (p + P) mod S is the first with
// S o = p + P - t), in the same ascending (tz, ty, tx, co) order: bit-identical results.
template <int CIN>
__global__ __launch_bounds__(kThreads) void conv_dgrad_direct_all_kernel(const float *__restrict__ gout, const float *__restrict__ w,
                                                                        float *__restrict__ gin, int cout, int Zi, int Yi, int Xi, int S,
                                                                        int KZ, int KS) {
  const int b = blockIdx.z;
  const size_t in_cs = (size_t)Zi * Yi * Xi;
  const size_t p = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (p >= in_cs) return;
  const int x = (int)(p % Xi), y = (int)((p / Xi) % Yi), z = (int)(p / ((size_t)Xi * Yi));
  const int Sz = KZ == 1 ? 1 : S;   // 2D layers: one plane
  const int Zo = Zi / Sz, Yo = Yi / S, Xo = Xi / S, PZ = KZ / 2, P = KS / 2, T = KZ * KS * KS;
  const size_t out_cs = (size_t)Zo * Yo * Xo;
  float acc[CIN];
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.0f;
  for (int tz = (z + PZ) % Sz; tz < KZ; tz += Sz) {
    const int oz = (z + PZ - tz) / Sz;
    if (z + PZ - tz < 0 || oz >= Zo) continue;
    for (int ty = (y + P) % S; ty < KS; ty += S) {
      const int oy = (y + P - ty) / S;
      if (y + P - ty < 0 || oy >= Yo) continue;
      for (int tx = (x + P) % S; tx < KS; tx += S) {
        const int ox = (x + P - tx) / S;
        if (x + P - tx < 0 || ox >= Xo) continue;
        const size_t o = ((size_t)oz * Yo + oy) * Xo + ox;
        const int t = (tz * KS + ty) * KS + tx;
        for (int co = 0; co < cout; ++co) {
          const float g = gout[((size_t)b * cout + co) * out_cs + o];
          const float *wr = w + (size_t)co * CIN * T + t;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) acc[ci] = fmaf(g, wr[ci * T], acc[ci]);
        }
      }
    }
  }
#pragma unroll
  for (int ci = 0; ci < CIN; ++ci) gin[((size_t)b * CIN + ci) * in_cs + p] = acc[ci];
}

// ---- per-channel sums (batch statistics, bias gradients, ABN backward reductions) ------------------------------------
// x, y: (N, C, n) ; out[c][blk][2] (double) = partial sums of channel c over this block's slice:
//   MODE 0: sum x, sum x^2                     (ABN forward statistics; bias gradient = the first)
//   MODE 1: g = x * (y > 0 ? 1 : slope), xhat = (y2 - mean[c]) * rstd[c]: sum g, sum g * xhat   (ABN backward)
template <int MODE>
__global__ __launch_bounds__(kThreads) void channel_sums_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                               const float *__restrict__ y2, const float *__restrict__ mean,
                                                               const float *__restrict__ rstd, double *__restrict__ out, int N,
                                                               int C, size_t n, float slope) {
  const int c = blockIdx.y, nblk = gridDim.x;
  float s0 = 0.0f, s1 = 0.0f;
  double d0 = 0.0, d1 = 0.0;
  const float mu = MODE == 1 ? mean[c] : 0.0f, rs = MODE == 1 ? rstd[c] : 0.0f;
  int cnt = 0;
  // block `blockIdx.x` owns the slice [lo, hi) of every image's plane of channel c: contiguous, coalesced, no division per
  // element; four elements per thread and step are loaded before they are used
  const size_t per = (n + nblk - 1) / nblk, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  auto term = [&](size_t off, float &a, float &bq) {
    if (MODE == 0) {
      a = x[off];
      bq = a * a;
    } else {
      const float g = x[off] * (y[off] > 0.0f ? 1.0f : slope);
      a = g;
      bq = g * ((y2[off] - mu) * rs);
    }
  };
  for (int img = 0; img < N; ++img) {
    const size_t base = ((size_t)img * C + c) * n;
    for (size_t e = lo + threadIdx.x; e < hi; e += 4 * kThreads) {
      float a[4], bq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const size_t ee = e + (size_t)i * kThreads;
        a[i] = 0.0f; bq[i] = 0.0f;
        term(base + (ee < hi ? ee : e), a[i], bq[i]);
        if (ee >= hi) { a[i] = 0.0f; bq[i] = 0.0f; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) { s0 += a[i]; s1 += bq[i]; }
      cnt += 4;
      if (cnt >= 64) {   // fp32 runs of 64 values, then double
        d0 += s0; d1 += s1; s0 = s1 = 0.0f; cnt = 0;
      }
    }
  }
  d0 += s0; d1 += s1;
  __shared__ double red[2][kThreads];
  red[0][threadIdx.x] = d0; red[1][threadIdx.x] = d1;
  __syncthreads();
  for (int st = kThreads / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[((size_t)c * nblk + blockIdx.x) * 2] = red[0][0];
    out[((size_t)c * nblk + blockIdx.x) * 2 + 1] = red[1][0];
  }
}

// y = lrelu(x * a[c] + b[c])   (a = gamma * rstd, b = beta - mean * a)
__global__ __launch_bounds__(kThreads) void abn_apply_kernel(const float *__restrict__ x, const float *__restrict__ a, const float *__restrict__ bq,
                                                            float *__restrict__ y, int C, size_t n, float slope) {
  const int nc = blockIdx.y;
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= n) return;
  const int c = nc % C;
  const float v = fmaf(x[(size_t)nc * n + e], a[c], bq[c]);
  y[(size_t)nc * n + e] = v > 0.0f ? v : v * slope;
}

// gx = a[c] * (g - m1[c] - xhat * m2[c]),  g = gy * (y > 0 ? 1 : slope), xhat = (x - mean[c]) * rstd[c]
// (m1 = sum g / M, m2 = sum g xhat / M, a = gamma * rstd)
__global__ __launch_bounds__(kThreads) void abn_bwd_apply_kernel(const float *__restrict__ gy, const float *__restrict__ y, const float *__restrict__ x,
                                                                const float *__restrict__ a, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                const float *__restrict__ m1, const float *__restrict__ m2, float *__restrict__ gx,
                                                                int C, size_t n, float slope) {
  const int nc = blockIdx.y;
  const size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (e >= n) return;
  const int c = nc % C;
  const size_t off = (size_t)nc * n + e;
  const float g = gy[off] * (y[off] > 0.0f ? 1.0f : slope);
  const float xhat = (x[off] - mean[c]) * rstd[c];
  gx[off] = a[c] * (g - m1[c] - xhat * m2[c]);
}

// ---- device-side packing of a layer image (training: the weights change every step) ---------------------------------------
// out[i] = source[index[i]] over the virtual source vector [weight (n_w), bias (n_b), 0.0, 1.0]: the host packer's layout as
// a fixed gather (training.py derives `index` once per layer shape), one launch instead of a cat + index_select (+ the
// transpose / flip copies of an adjoint layer, which are folded into the index).
__global__ __launch_bounds__(kThreads) void pack_gather_kernel(const float *__restrict__ w, const float *__restrict__ bias,
                                                              const int *__restrict__ index, float *__restrict__ out, int n_w, int n_b,
                                                              int n_out) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= n_out) return;
  const int k = index[i];
  out[i] = k < n_w ? w[k] : (k < n_w + n_b ? bias[k - n_w] : (k == n_w + n_b ? 0.0f : 1.0f));
}

// The same gather for EVERY layer image of a training step in one launch (training.py's pack plan: 88 launches of ~4.7 us each were 0.41 ms of an
// 11.3 ms step): workgroup b works on the segment whose first workgroup is the last one <= b (binary search over <= a few hundred segments).
struct PackSegment {   // mirrors include/casmvs.h: casmvs_pack_segment
  const float *w, *bias;
  const int *index;
  float *out;
  int n_w, n_b, n_out, first_block;
};
__global__ __launch_bounds__(kThreads) void pack_gather_batch_kernel(const PackSegment *__restrict__ segs, int n_seg) {
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackSegment sg = segs[lo];
  const int i = ((int)blockIdx.x - sg.first_block) * kThreads + threadIdx.x;
  if (i >= sg.n_out) return;
  const int k = sg.index[i];
  sg.out[i] = k < sg.n_w ? sg.w[k] : (k < sg.n_w + sg.n_b ? sg.bias[k - sg.n_w] : (k == sg.n_w + sg.n_b ? 0.0f : 1.0f));
}

// ---- per-channel epilogues of the batch statistics ---------------------------------------------------------------------
// One thread per channel turns the workgroups' partial sums (C, blocks, 2) into everything the layer needs, in double like
// F.batch_norm's accumulation: forward -> mean, biased variance, rstd, scale = gamma * rstd, shift = beta - mean * scale and
// the running statistics (momentum update with the UNBIASED variance); backward -> grad_gamma, grad_beta, m1, m2.
// abs_eps >= 0: InPlaceABN's gamma = |weight| + abs_eps (inplace_abn.py), its gradient goes back through the abs.
// (As ~40 tensor-sized-1 torch operations per layer these epilogues were 1500 of the 1900 kernels of a training step.)
__global__ __launch_bounds__(64) void abn_train_finish_kernel(const double *__restrict__ sums, int blocks, int C, double M,
                                                             const float *__restrict__ weight, const float *__restrict__ bias, float abs_eps,
                                                             float eps, float momentum, float *__restrict__ rmean, float *__restrict__ rvar,
                                                             float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean,
                                                             float *__restrict__ rstd) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  for (int b = 0; b < blocks; ++b) {
    s0 += sums[((size_t)c * blocks + b) * 2];
    s1 += sums[((size_t)c * blocks + b) * 2 + 1];
  }
  const double mu = s0 / M;
  double var = s1 / M - mu * mu;   // biased
  var = var > 0.0 ? var : 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const double g = abs_eps >= 0.0f ? (double)(fabsf(weight[c]) + abs_eps) : (double)weight[c];
  scale[c] = (float)(g * rs);
  shift[c] = (float)((double)bias[c] - mu * g * rs);
  mean[c] = (float)mu;
  rstd[c] = (float)rs;
  if (rmean) {   // F.batch_norm: running = (1 - momentum) * running + momentum * batch (running_var: unbiased)
    const float keep = (float)(1.0 - (double)momentum);
    rmean[c] = rmean[c] * keep + momentum * (float)mu;
    rvar[c] = rvar[c] * keep + momentum * (float)(var * (M / (M > 1.0 ? M - 1.0 : 1.0)));
  }
}

__global__ __launch_bounds__(64) void abn_bwd_finish_kernel(const double *__restrict__ sums, int blocks, int C, double M,
                                                           const float *__restrict__ weight, float abs_eps, float *__restrict__ gweight,
                                                           float *__restrict__ gbias, float *__restrict__ m1, float *__restrict__ m2) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s0 = 0.0, s1 = 0.0;
  for (int b = 0; b < blocks; ++b) {
    s0 += sums[((size_t)c * blocks + b) * 2];
    s1 += sums[((size_t)c * blocks + b) * 2 + 1];
  }
  const float w = weight[c];
  const float sgn = abs_eps >= 0.0f ? (w > 0.0f ? 1.0f : (w < 0.0f ? -1.0f : 0.0f)) : 1.0f;   // d|w| / dw
  gweight[c] = (float)s1 * sgn;
  gbias[c] = (float)s0;
  m1[c] = (float)(s0 / M);
  m2[c] = (float)(s1 / M);
}

// ---- the per-channel epilogue INSIDE the elementwise kernel ----------------------------------------------------------------
// The two epilogue kernels above are one-workgroup launches between the statistics pass and the elementwise pass of every layer and direction: 76 of
// them per training step, ~6.3 us each plus the kernel boundary.  These forms let every workgroup of the elementwise pass reduce the channel's <= 256
// partial sums itself (one per thread, a fixed-order tree in LDS: every workgroup of a channel gets the same bits) and derive the channel's constants;
// the first workgroup of the channel's first image also writes them out (the backward pass needs them) and updates the running statistics.
constexpr int kAbnChunk = 2048;   // elements per workgroup: two 16-byte accesses per thread when the plane size allows
typedef float vf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void channel_totals(const double *__restrict__ sums, int blocks, int c, double &t0, double &t1) {
  __shared__ double red[2][kThreads];
  double s0 = 0.0, s1 = 0.0;
  for (int b = threadIdx.x; b < blocks; b += kThreads) {
    s0 += sums[((size_t)c * blocks + b) * 2];
    s1 += sums[((size_t)c * blocks + b) * 2 + 1];
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int st = kThreads / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      red[0][threadIdx.x] += red[0][threadIdx.x + st];
      red[1][threadIdx.x] += red[1][threadIdx.x + st];
    }
    __syncthreads();
  }
  t0 = red[0][0]; t1 = red[1][0];
}

// y = lrelu(x * a + b) with a, b from the batch statistics of casmvs_channel_sums_f64 (abn_train_finish_kernel's arithmetic)
__global__ __launch_bounds__(kThreads) void abn_train_apply_kernel(const float *__restrict__ x, const double *__restrict__ sums, int blocks, double M,
                                                                  const float *__restrict__ weight, const float *__restrict__ bias, float abs_eps,
                                                                  float eps, float momentum, float *__restrict__ rmean, float *__restrict__ rvar,
                                                                  float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean,
                                                                  float *__restrict__ rstd, float *__restrict__ y, int C, size_t n, float slope) {
  const int nc = blockIdx.y, c = nc % C;
  double s0, s1;
  channel_totals(sums, blocks, c, s0, s1);
  const double mu = s0 / M;
  double var = s1 / M - mu * mu;   // biased
  var = var > 0.0 ? var : 0.0;
  const double rs = 1.0 / sqrt(var + (double)eps);
  const double g = abs_eps >= 0.0f ? (double)(fabsf(weight[c]) + abs_eps) : (double)weight[c];
  const float a = (float)(g * rs), bq = (float)((double)bias[c] - mu * g * rs);
  if (blockIdx.x == 0 && nc == c && threadIdx.x == 0) {
    scale[c] = a;
    shift[c] = bq;
    mean[c] = (float)mu;
    rstd[c] = (float)rs;
    if (rmean) {   // F.batch_norm: running = (1 - momentum) * running + momentum * batch (running_var: unbiased)
      const float keep = (float)(1.0 - (double)momentum);
      rmean[c] = rmean[c] * keep + momentum * (float)mu;
      rvar[c] = rvar[c] * keep + momentum * (float)(var * (M / (M > 1.0 ? M - 1.0 : 1.0)));
    }
  }
  const size_t lo = (size_t)blockIdx.x * kAbnChunk, hi = lo + kAbnChunk < n ? lo + kAbnChunk : n;
  const float *xp = x + (size_t)nc * n;
  float *yp = y + (size_t)nc * n;
  if ((n & 3) == 0) {   // planes start on 16-byte boundaries
    for (size_t e = lo + 4 * (size_t)threadIdx.x; e < hi; e += 4 * kThreads) {
      vf4 v = *reinterpret_cast<const vf4 *>(xp + e);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t = fmaf(v[i], a, bq);
        v[i] = t > 0.0f ? t : t * slope;
      }
      *reinterpret_cast<vf4 *>(yp + e) = v;
    }
  } else {
    for (size_t e = lo + threadIdx.x; e < hi; e += kThreads) {
      const float v = fmaf(xp[e], a, bq);
      yp[e] = v > 0.0f ? v : v * slope;
    }
  }
}

// gx = a * (g - m1 - xhat * m2) with m1, m2 from the sums of casmvs_abn_backward_sums_f64 (abn_bwd_finish_kernel's arithmetic)
__global__ __launch_bounds__(kThreads) void abn_bwd_apply_stats_kernel(const float *__restrict__ gy, const float *__restrict__ y, const float *__restrict__ x,
                                                                      const double *__restrict__ sums, int blocks, double M,
                                                                      const float *__restrict__ weight, float abs_eps, const float *__restrict__ a,
                                                                      const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                      float *__restrict__ gweight, float *__restrict__ gbias, float *__restrict__ gx,
                                                                      int C, size_t n, float slope) {
  const int nc = blockIdx.y, c = nc % C;
  double s0, s1;
  channel_totals(sums, blocks, c, s0, s1);
  const float m1 = (float)(s0 / M), m2 = (float)(s1 / M);
  if (blockIdx.x == 0 && nc == c && threadIdx.x == 0) {
    const float w = weight[c];
    const float sgn = abs_eps >= 0.0f ? (w > 0.0f ? 1.0f : (w < 0.0f ? -1.0f : 0.0f)) : 1.0f;   // d|w| / dw
    gweight[c] = (float)s1 * sgn;
    gbias[c] = (float)s0;
  }
  const float ac = a[c], mu = mean[c], rs = rstd[c];
  const size_t lo = (size_t)blockIdx.x * kAbnChunk, hi = lo + kAbnChunk < n ? lo + kAbnChunk : n;
  const size_t base = (size_t)nc * n;
  auto one = [&](float gyv, float yv, float xv) {
    const float g = gyv * (yv > 0.0f ? 1.0f : slope);
    const float xhat = (xv - mu) * rs;
    return ac * (g - m1 - xhat * m2);
  };
  if ((n & 3) == 0) {
    for (size_t e = lo + 4 * (size_t)threadIdx.x; e < hi; e += 4 * kThreads) {
      const vf4 gv = *reinterpret_cast<const vf4 *>(gy + base + e), yv = *reinterpret_cast<const vf4 *>(y + base + e);
      const vf4 xv = *reinterpret_cast<const vf4 *>(x + base + e);
      vf4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = one(gv[i], yv[i], xv[i]);
      *reinterpret_cast<vf4 *>(gx + base + e) = o;
    }
  } else {
    for (size_t e = lo + threadIdx.x; e < hi; e += kThreads) gx[base + e] = one(gy[base + e], y[base + e], x[base + e]);
  }
}

// ---- FPN top-down step ----------------------------------------------------------------------------------------------------
// ATen upsample_bilinear2d, align_corners = True: src = o * (n_in - 1) / (n_out - 1), i0 = floor(src), i1 = i0 + (i0 < n_in - 1),
// l1 = src - i0, l0 = 1 - l1; value = l0y * (l0x v00 + l1x v01) + l1y * (l0x v10 + l1x v11)
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp lerp_of(int o, int n_in, int n_out) {
  const float scale = n_out > 1 ? (float)(n_in - 1) / (float)(n_out - 1) : 0.0f;
  const float src = scale * (float)o;
  Lerp r;
  r.i0 = (int)src;
  r.i1 = r.i0 + (r.i0 < n_in - 1 ? 1 : 0);
  r.l1 = src - (float)r.i0;
  r.l0 = 1.0f - r.l1;
  return r;
}

__global__ __launch_bounds__(kThreads) void upsample2x_add_kernel(const float *__restrict__ lat, const float *__restrict__ up, float *__restrict__ out,
                                                                 int H, int W) {
  const int nc = blockIdx.y, h = H / 2, w = W / 2;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= H * W) return;
  const int y = p / W, x = p - y * W;
  const Lerp ly = lerp_of(y, h, H), lx = lerp_of(x, w, W);
  const float *u = up + (size_t)nc * h * w;
  const float v = ly.l0 * (lx.l0 * u[ly.i0 * w + lx.i0] + lx.l1 * u[ly.i0 * w + lx.i1]) +
                  ly.l1 * (lx.l0 * u[ly.i1 * w + lx.i0] + lx.l1 * u[ly.i1 * w + lx.i1]);
  out[(size_t)nc * H * W + p] = v + lat[(size_t)nc * H * W + p];
}

// grad_up[iy][ix] = sum over the fine pixels whose stencil contains (iy, ix): a gather over the <= 6 x 6 candidates, no atomics
__global__ __launch_bounds__(kThreads) void upsample2x_bwd_kernel(const float *__restrict__ gout, float *__restrict__ gup, int H, int W) {
  const int nc = blockIdx.y, h = H / 2, w = W / 2;
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= h * w) return;
  const int iy = p / w, ix = p - iy * w;
  const float *g = gout + (size_t)nc * H * W;
  const int y_lo = max(0, 2 * iy - 3), y_hi = min(H - 1, 2 * iy + 4), x_lo = max(0, 2 * ix - 3), x_hi = min(W - 1, 2 * ix + 4);
  float acc = 0.0f;
  for (int y = y_lo; y <= y_hi; ++y) {
    const Lerp ly = lerp_of(y, h, H);
    const float wy = (ly.i0 == iy ? ly.l0 : 0.0f) + (ly.i1 == iy ? ly.l1 : 0.0f);
    if (wy == 0.0f && ly.i0 != iy && ly.i1 != iy) continue;
    for (int x = x_lo; x <= x_hi; ++x) {
      const Lerp lx = lerp_of(x, w, W);
      const float wx = (lx.i0 == ix ? lx.l0 : 0.0f) + (lx.i1 == ix ? lx.l1 : 0.0f);
      acc += g[y * W + x] * (wy * wx);
    }
  }
  gup[(size_t)nc * h * w + p] = acc;
}

// ---- variance cost volume backward ----------------------------------------------------------------------------------
// var = Q / V - (S / V)^2 with S = sum of the V views' values, Q = sum of their squares (mvsnet.py:140-167), so
// d var / d x_v = 2 x_v / V - 2 S / V^2 for the reference (x_0 = ref feature, every plane) and for each warped view.
// The source-view gradient is the transpose of the bilinear gather (modules.py:87-89): a scatter.  One workgroup owns
// (a 32 x TH tile of reference pixels, 8 planes, CG channels, ONE source view): everything it scatters falls into the
// bounding box of its taps in that view (the epipolar band of the tile: ~44 x 26 pixels at TH = 16), so the contributions are
// accumulated in an LDS image of that box and the box is added to the gradient map once, one atomic per touched element
// and channel - ~2 atomics per pixel and channel instead of the ~16 of a per-tap scatter.
//
// ORDER-INDEPENDENT ACCUMULATION (round 5; train.py:99-127 must be reproducible run to run): every sum of this kernel - the LDS image AND the gradient map
// the images are flushed into - is a 64-bit INTEGER in fixed point, and integer addition is associative: the result does not depend on the order in which
// lanes, waves or workgroups add.  (Round 4 kept the LDS image in fixed point with a scale per workgroup and flushed it with float atomics: 18 runs of the same
// 12 SGD steps gave 18 loss trajectories.)  One scale per (sample, channel), known before the kernel starts:
//   volume_absmax_kernel   G[b][c] = largest finite |upstream gradient| of the channel's (group's) volume, F[b][c] = largest finite |feature| over all views
//                          (a tree of maxima over the bit patterns, no atomics: fixed_accum.h)
//   bound[b][c] = 16 G F / V (variance: a contribution is g (2 x_v / V - 2 S / V^2) w with |x_v| <= F, |S| <= V F, w <= 1: at most 4 G F / V, two of them
//                          merged by the lane exchange below: 8 G F / V) or 4 k G F (correlation: g k ref w <= k G F), a STRICT bound with 2x slack for the
//                          float32 roundings: 2^be >= bound, unit 2^(be - U)
//   U = min(44, 62 - ceil log2(D h w)): a cell receives at most one contribution per (pixel, plane), each below 2^(U - 1) units: no overflow by construction
//                          (44: the conversion below is exact for |x| < 2^51 and the reference view's per-chunk sum is < 2^7 bounds)
// so a contribution 2^-15 of the bound still carries 24 bits at the LARGEST volumes, and the final value is the exact integer sum rounded ONCE to float32
// (costvol_fixed_finish_kernel) - closer to the float64 derivative than a float32 accumulation.  Non-finite contributions (a NaN / infinite upstream gradient
// or feature; the maxima skip them) cannot be integers: they are added to the float32 output map itself with float atomics - a sum of non-finite values is
// non-finite in any order - and the finish adds the fixed-point sum to it: exactly the elements a float accumulation would poison are poisoned.
// G F = 0: every finite contribution is exactly zero; the scale is 0 and so is the gradient.
// Phase 1: the tile's taps in the view -> box (wave-uniform after an LDS min / max).  Phase 2: per (pixel, plane) the warped
// values of ALL views are re-gathered (S needs them), the own view's gradient goes into the box; the workgroups of the
// first source view also accumulate the reference view's gradient in registers.  Phase 3: box -> global.
// wave shift by one lane (gfx9 DPP wave_shr:1 / wave_shl:1): lane i reads lane i - 1 / i + 1; the first / last lane reads 0
__device__ __forceinline__ int lane_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int lane_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ float lane_prev(float v) { return __builtin_bit_cast(float, lane_prev(__builtin_bit_cast(int, v))); }

// grad_feats (zero, or the non-finite contributions) += the fixed-point sums in float32, one rounding per element
template <bool GWC>
__global__ __launch_bounds__(kThreads) void costvol_fixed_finish_kernel(const unsigned long long *__restrict__ acc, const unsigned *__restrict__ gmax,
                                                                        const unsigned *__restrict__ fmax, float *__restrict__ gfeats, int V, int C, int G, int hw,
                                                                        int U) {
  const int row = blockIdx.y;   // (b, v, c)
  const int b = row / (V * C), c = row % C;
  const int cpg = GWC ? C / G : 1;
  int be;
  const bool ok = fixed_exponent(gmax[GWC ? b * G + c / cpg : b * C + c], fmax[b * C + c], GWC ? 4.0 / ((double)cpg * (double)(V - 1)) : 16.0 / (double)V, be);
  const double from_fixed = ok ? pow2_double(be - U) : 0.0;
  const size_t base = (size_t)row * hw;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < hw; p += gridDim.x * kThreads) {
    const long long val = (long long)acc[base + p];
    gfeats[base + p] += (float)((double)val * from_fixed);
  }
}

// VS: the number of source views when the instantiation fixes it (their gathers are then issued together), 0 = V - 1 at run time.
// GWC: the group-wise correlation volume (mvsnet.py:142-144,157-162,169-172) instead of the variance: vol[g] = sum_v mean_{c in g} ref[c] warped_v[c] / (V - 1),
// gvol (B, G, D, h, w); d / d warped_v[c] = gvol[group of c] ref[c] k and d / d ref[c] = gvol[group of c] k sum_v warped_v[c], k = 1 / ((C / G) (V - 1)):
// the same scatter with another contribution - only the workgroups of the first source view gather (they own the reference view's gradient).
template <int CG, int TH, int VS, bool GWC>
__global__ __launch_bounds__(kThreads, 4) void costvol_var_bwd_kernel(const float *__restrict__ feats, const float *__restrict__ proj,
                                                                  const float *__restrict__ depth, const float *__restrict__ gvol,
                                                                  float *__restrict__ gfeats, unsigned long long *__restrict__ acc,
                                                                  const unsigned *__restrict__ gmax, const unsigned *__restrict__ fmax, int V, int C, int H, int W,
                                                                  int D, int tiles_x, int DCH, int CAP, int G, int U) {
  constexpr int TS = 32, RPT = TS * TH / kThreads, RSTEP = kThreads / TS;   // tile: 32 columns x TH rows, a thread's pixels RSTEP rows apart
  CASMVS_DYNAMIC_LDS(unsigned long long, box);   // [CG][bh][bw], CAP cells per channel: fixed point, the gradient map's own units
  __shared__ int ext[(RPT + 1) * 4];   // tap boxes of the tile's RPT bands of 8 rows, then their union
  const int tid = threadIdx.x, b = blockIdx.z, hw = H * W;
  int r = blockIdx.y;
  const int v = 1 + r % (V - 1); r /= (V - 1);
  const int groups = C / CG, c0 = (r % groups) * CG, d_begin = (r / groups) * DCH, d_end = min(d_begin + DCH, D);
  const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
  const int x = txi * TS + (tid & (TS - 1)), yb = tyi * TH + tid / TS;
  const float *fb = feats + (size_t)b * V * C * hw;
  float *gb = gfeats + (size_t)b * V * C * hw;
  unsigned long long *ab = acc + (size_t)b * V * C * hw;
  const float *Pb = proj + (size_t)b * (V - 1) * 12;
  const float *Pv = Pb + (v - 1) * 12;
  const float *db = depth + (size_t)b * D * hw;
  const float fV = (float)V;
  const int cpg = GWC ? C / G : 1;
  const float kg = GWC ? 1.0f / ((float)cpg * (float)(V - 1)) : 0.0f;
  size_t goff[CG];   // the upstream gradient's plane 0 of the channel (variance) / of the channel's group (correlation)
  double to_fixed[CG];   // 2^(U - be) of the channel, 0 without a scale (then every finite contribution is 0)
  bool checked = false;
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    goff[c] = GWC ? ((size_t)b * G + (c0 + c) / cpg) * D * hw : ((size_t)b * C + c0 + c) * D * hw;
    int be;
    const unsigned gw = gmax[GWC ? b * G + (c0 + c) / cpg : b * C + c0 + c], fw = fmax[b * C + c0 + c];
    const bool ok = fixed_exponent(gw, fw, GWC ? 4.0 / ((double)cpg * (double)(V - 1)) : 16.0 / (double)V, be);
    to_fixed[c] = ok ? pow2_double(U - be) : 0.0;
    checked = checked || needs_finite_check(gw, fw, be);
  }
  // workgroup-uniform (scalar loads): a non-finite value somewhere in these channels' operands - never in a healthy run - selects the loop that checks
  // every contribution; the other loop carries no trace of the check (a per-step test inside ONE loop cost 14 % of the kernel: profiles/r05_varbwd_ab.txt)
  checked = __builtin_amdgcn_readfirstlane((unsigned)checked) != 0u;
  auto fx = [](float val, double scale) { return to_fixed_point(val, scale); };

  // ---- 1. bounding boxes of this view's live taps: the tile's, and (published only when the tile's does not fit the LDS
  // image) one per band of RSTEP rows (a thread's j-th pixel)
  if (tid < (RPT + 1) * 4) ext[tid] = (tid & 1) ? INT_MIN : INT_MAX;
  __syncthreads();
  int bxmn[RPT], bxmx[RPT], bymn[RPT], bymx[RPT];
  int xmn = INT_MAX, xmx = INT_MIN, ymn = INT_MAX, ymx = INT_MIN;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    bxmn[j] = INT_MAX; bxmx[j] = INT_MIN; bymn[j] = INT_MAX; bymx[j] = INT_MIN;
    // clamped, unconditional loads, eight planes at a time: every load of a pixel's planes is in flight before the first tap is computed
    const int yr = yb + j * RSTEP;
    const bool valid = x < W && yr < H;
    const int p = min(yr, H - 1) * W + min(x, W - 1);
    for (int d0 = d_begin; d0 < d_end; d0 += 8) {
      float dvs[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) dvs[i] = db[(size_t)min(d0 + i, d_end - 1) * hw + p];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const Taps t = plane_sweep_taps(Pv, (float)x, (float)yr, dvs[i], W, H);
        if (valid && taps_live(t)) {
          bxmn[j] = min(bxmn[j], t.xl); bxmx[j] = max(bxmx[j], t.xl + 1);
          bymn[j] = min(bymn[j], t.yn); bymx[j] = max(bymx[j], t.ys);
        }
      }
    }
    xmn = min(xmn, bxmn[j]); xmx = max(xmx, bxmx[j]);
    ymn = min(ymn, bymn[j]); ymx = max(ymx, bymx[j]);
  }
  if (xmn <= xmx) {
    atomicMin(&ext[RPT * 4 + 0], xmn); atomicMax(&ext[RPT * 4 + 1], xmx);
    atomicMin(&ext[RPT * 4 + 2], ymn); atomicMax(&ext[RPT * 4 + 3], ymx);
  }
  __syncthreads();
  // The whole tile's box in one LDS image when it fits; otherwise band by band (noisy depth maps spread a tile's taps over
  // far more than its own extent), and a band whose box still does not fit adds straight to the gradient map.
  const bool whole = ext[RPT * 4] > ext[RPT * 4 + 1] ||
                     (ext[RPT * 4 + 1] - ext[RPT * 4] + 1) * (ext[RPT * 4 + 3] - ext[RPT * 4 + 2] + 1) <= CAP;
  if (!whole) {
#pragma unroll
    for (int j = 0; j < RPT; ++j)
      if (bxmn[j] <= bxmx[j]) {
        atomicMin(&ext[j * 4 + 0], bxmn[j]); atomicMax(&ext[j * 4 + 1], bxmx[j]);
        atomicMin(&ext[j * 4 + 2], bymn[j]); atomicMax(&ext[j * 4 + 3], bymx[j]);
      }
    __syncthreads();
  }
  float *gsv = gb + ((size_t)v * C + c0) * hw;
  unsigned long long *asv = ab + ((size_t)v * C + c0) * hw;
  const int lane = tid & 63;
  for (int seg = 0; seg < (whole ? 1 : RPT); ++seg) {
  const int *eb = ext + (whole ? RPT : seg) * 4;
  const int bx0 = eb[0], by0 = eb[2];
  const bool any = eb[0] <= eb[1];
  const int bw = any ? eb[1] - eb[0] + 1 : 0, bh = any ? eb[3] - eb[2] + 1 : 0;
  const int cells = bw * bh;
  const bool in_lds = cells <= CAP;
  if (in_lds)
    for (int e = tid; e < CG * cells; e += kThreads) box[e] = 0ull;
  __syncthreads();

  // ---- 2. per (pixel, plane): S over the views, own view's gradient into the box
  // every lane runs the loops (the lanes exchange tap gradients with their neighbours by DPP below); a lane outside the
  // image works on a clamped pixel with a zero upstream gradient and never adds anything
  auto scatter = [&](auto checked_c) {
  constexpr bool CHECKED = decltype(checked_c)::value;
  for (int j = whole ? 0 : seg; j < (whole ? RPT : seg + 1); ++j) {
    const int yr = yb + j * RSTEP;
    const bool valid = x < W && yr < H;
    const int xc = min(x, W - 1), y = min(yr, H - 1), p = y * W + xc;
    float ref[CG], gref[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) {
      ref[c] = fb[(size_t)(c0 + c) * hw + p];
      gref[c] = 0.0f;
    }
    // the plane's hypothesis and upstream gradients are loaded one plane ahead (clamped, unconditional: nothing in the loop waits on a load it has just
    // issued except the gathers - seven serialised round trips per step before, two now, one with VS)
    const float *gp = gvol + p;
    float dv_next = db[(size_t)d_begin * hw + p], g_next[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) g_next[c] = gp[goff[c] + (size_t)d_begin * hw];
    const bool gathers = !GWC || v == 1;   // correlation: the warped values only enter the reference view's gradient
    const int nv = V;
    for (int d = d_begin; d < d_end; ++d) {
      const float dv = dv_next;
      float S[CG], xv[CG], gd[CG];
      Taps tv;
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        S[c] = GWC ? 0.0f : ref[c];
        xv[c] = 0.0f;
        gd[c] = valid ? g_next[c] : 0.0f;
      }
      const int dn = min(d + 1, d_end - 1);
      dv_next = db[(size_t)dn * hw + p];
#pragma unroll
      for (int c = 0; c < CG; ++c) g_next[c] = gp[goff[c] + (size_t)dn * hw];
      if (GWC && !gathers) {
        tv = plane_sweep_taps(Pv, (float)xc, (float)y, dv, W, H);
      } else if constexpr (VS > 0) {   // every view's taps, then every gather, then the arithmetic: one round trip for all views
        Taps ts[VS];
        float raw[VS][CG][4];
#pragma unroll
        for (int u = 0; u < VS; ++u) ts[u] = plane_sweep_taps(Pb + u * 12, (float)xc, (float)y, dv, W, H);
#pragma unroll
        for (int u = 0; u < VS; ++u) {
          const float *fu = fb + ((size_t)(u + 1) * C + c0) * hw;
          const int on = ts[u].yn * W + ts[u].xl, os = ts[u].ys * W + ts[u].xl;
#pragma unroll
          for (int c = 0; c < CG; ++c) {
            const float *fc = fu + (size_t)c * hw;
            raw[u][c][0] = fc[on]; raw[u][c][1] = fc[on + 1]; raw[u][c][2] = fc[os]; raw[u][c][3] = fc[os + 1];
          }
        }
        asm volatile("" ::: "memory");   // keeps the loads above the first use
#pragma unroll
        for (int u = 0; u < VS; ++u) {
          const bool own = u + 1 == v;
          if (own) tv = ts[u];
#pragma unroll
          for (int c = 0; c < CG; ++c) {
            const float val = fmaf(raw[u][c][3], ts[u].w_sr, fmaf(raw[u][c][2], ts[u].w_sl, fmaf(raw[u][c][1], ts[u].w_nr, raw[u][c][0] * ts[u].w_nl)));
            S[c] += val;
            if (own) xv[c] = val;
          }
        }
      } else {
        for (int u = 1; u < nv; ++u) {   // dead voxels: all four weights are 0 and the (clamped) addresses are valid
          const Taps t = plane_sweep_taps(Pb + (u - 1) * 12, (float)xc, (float)y, dv, W, H);
          const float *fu = fb + ((size_t)u * C + c0) * hw;
          const int on = t.yn * W + t.xl, os = t.ys * W + t.xl;
          const bool own = u == v;
          if (own) tv = t;
#pragma unroll
          for (int c = 0; c < CG; ++c) {
            const float *fc = fu + (size_t)c * hw;
            const float val = fmaf(fc[os + 1], t.w_sr, fmaf(fc[os], t.w_sl, fmaf(fc[on + 1], t.w_nr, fc[on] * t.w_nl)));
            S[c] += val;
            if (own) xv[c] = val;
          }
        }
      }
      const bool live = valid && taps_live(tv);
      // Neighbouring lanes are neighbouring pixels: lane i's RIGHT tap column is usually lane i + 1's LEFT one.  The right
      // contribution travels one lane up (DPP) and is added to the neighbour's left one: ~2 adds per channel and row pair
      // instead of 4.  A lane keeps its right tap only when the next lane does not continue the run; dead lanes (key -100) never match.
      const int kxl = live ? tv.xl : -100;
      const int pxl = lane_prev(kxl), pyn = lane_prev(tv.yn), pys = lane_prev(tv.ys);
      const int nxl = lane_next(kxl), nyn = lane_next(tv.yn), nys = lane_next(tv.ys);
      const bool mp_n = live && lane > 0 && pyn == tv.yn && pxl + 1 == tv.xl, mp_s = live && lane > 0 && pys == tv.ys && pxl + 1 == tv.xl;
      const bool ab_n = lane < 63 && nyn == tv.yn && nxl == tv.xl + 1, ab_s = lane < 63 && nys == tv.ys && nxl == tv.xl + 1;
      const int lo_n = (tv.yn - by0) * bw + (tv.xl - bx0), lo_s = (tv.ys - by0) * bw + (tv.xl - bx0);
      const int go_n = tv.yn * W + tv.xl, go_s = tv.ys * W + tv.xl;
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        const float g = gd[c];
        float gx;
        if (GWC) {
          gref[c] += g * kg * S[c];
          gx = g * kg * ref[c];
        } else {
          const float common = 2.0f * S[c] / (fV * fV);
          gref[c] += g * (2.0f * ref[c] / fV - common);
          gx = g * (2.0f * xv[c] / fV - common);
        }
        const float rn = gx * tv.w_nr, rs = gx * tv.w_sr;
        const float prn = lane_prev(rn), prs = lane_prev(rs);
        const float an = gx * tv.w_nl + (mp_n ? prn : 0.0f), as = gx * tv.w_sl + (mp_s ? prs : 0.0f);
        if (live) {
          // (separate LDS / global code: ONE pointer that may be either compiles to flat atomics - no ds_add_u64 at all, 132 VGPRs)
          float *qf = gsv + (size_t)c * hw;
          auto add = [&](unsigned long long *q, int o, int go, float val) {
            if (!CHECKED || is_finite(val)) atomicAdd(q + o, fx(val, to_fixed[c]));
            else unsafeAtomicAdd(qf + go, val);
          };
          if (in_lds) {
            unsigned long long *q = box + c * cells;
            add(q, lo_n, go_n, an);
            if (!ab_n) add(q, lo_n + 1, go_n + 1, rn);
            add(q, lo_s, go_s, as);
            if (!ab_s) add(q, lo_s + 1, go_s + 1, rs);
          } else {
            unsigned long long *q = asv + (size_t)c * hw;
            add(q, go_n, go_n, an);
            if (!ab_n) add(q, go_n + 1, go_n + 1, rn);
            add(q, go_s, go_s, as);
            if (!ab_s) add(q, go_s + 1, go_s + 1, rs);
          }
        }
      }
    }
    if (v == 1 && valid) {   // view 0 (the reference features): one add per plane chunk
#pragma unroll
      for (int c = 0; c < CG; ++c) {
        if (!CHECKED || is_finite(gref[c])) atomicAdd(ab + (size_t)(c0 + c) * hw + p, fx(gref[c], to_fixed[c]));
        else unsafeAtomicAdd(gb + (size_t)(c0 + c) * hw + p, gref[c]);
      }
    }
  }
  };
  if (checked) scatter(std::true_type{});
  else scatter(std::false_type{});
  __syncthreads();

  // ---- 3. box -> gradient map (lanes = consecutive columns of a box row of one channel plane)
  if (in_lds)
    for (int e = tid; e < CG * cells; e += kThreads) {
      const unsigned long long val = box[e];
      if (val != 0ull) {
        const int c = e / cells, cell = e - c * cells, row = cell / bw, col = cell - row * bw;
        atomicAdd(asv + (size_t)c * hw + (by0 + row) * W + bx0 + col, val);
      }
    }
  __syncthreads();   // the image is zeroed again for the next band
  }
}

struct WgradGeom {
  int S, KZ, KS, transposed;
};
bool wgrad_geom(int kind, WgradGeom &g) {
  switch (kind) {
    case CASMVS_CONV_S1: g = {1, 3, 3, 0}; return true;
    case CASMVS_CONV_S2: g = {2, 3, 3, 0}; return true;
    case CASMVS_CONV_T2: g = {2, 3, 3, 1}; return true;
    case CASMVS_CONV2D_K3: g = {1, 1, 3, 0}; return true;
    case CASMVS_CONV2D_K5S2: g = {2, 1, 5, 0}; return true;
    case CASMVS_CONV2D_K1:
    case CASMVS_CONV2D_K1_UP: g = {1, 1, 1, 0}; return true;
    default: return false;
  }
}

struct WgradLaunch {
  int Cs, Cb, Zs, Ys, Xs, row_tiles, cb_groups, tiles_z, tiles_y, tiles_x, gx, gy, T;
};
// D, H, W: the layer's INPUT dims (as in the forward call)
bool wgrad_launch(int kind, int B, int cin, int cout, int D, int H, int W, WgradLaunch &l) {
  WgradGeom g;
  if (!wgrad_geom(kind, g) || B < 1 || cin < 1 || cout < 1 || D < 1 || H < 1 || W < 1) return false;
  if (g.KZ == 1 && D != 1) return false;
  if (g.S == 2 && !g.transposed && ((g.KZ == 3 && D % 2) || H % 2 || W % 2)) return false;
  if (g.transposed) {   // small = input grid
    l.Cs = cin; l.Cb = cout; l.Zs = D; l.Ys = H; l.Xs = W;
  } else {              // small = output grid
    l.Cs = cout; l.Cb = cin;
    l.Zs = g.KZ == 3 ? D / g.S : 1; l.Ys = H / g.S; l.Xs = W / g.S;
  }
  l.T = g.KZ * g.KS * g.KS;
  l.row_tiles = casmvs::ceil_div(l.Cs, 16);
  l.cb_groups = casmvs::ceil_div(l.Cb, 16);
  l.tiles_z = casmvs::ceil_div(l.Zs, (g.KZ == 3 && g.S == 1) ? 4 : 1);   // WgradCfg::TZ
  l.tiles_y = casmvs::ceil_div(l.Ys, 4);
  l.tiles_x = casmvs::ceil_div(l.Xs, 16);
  l.gy = l.row_tiles * l.cb_groups;
  const long tiles = (long)B * l.tiles_z * l.tiles_y * l.tiles_x;
  long gx = 512 / l.gy;
  if (gx < 1) gx = 1;
  if (gx > tiles) gx = tiles;
  l.gx = (int)gx;
  return true;
}

template <int S, int KZ, int KS, bool VEC>
int launch_wgrad_v(const WgradLaunch &l, const float *small, const float *big, float *partial, int B, hipStream_t st) {
  auto kernel = conv_wgrad_kernel<S, KZ, KS, VEC>;
  const size_t lds = WgradCfg<S, KZ, KS>::LDS_BYTES;
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), lds, "conv_wgrad_kernel")) return rc;
  hipLaunchKernelGGL(kernel, dim3((unsigned)l.gx, (unsigned)l.gy), dim3(kThreads), lds, st, small, big, partial, B, l.Cs, l.Cb, l.Zs,
                     l.Ys, l.Xs, l.cb_groups, l.tiles_z, l.tiles_y, l.tiles_x);
  return casmvs::check_launch("conv_wgrad_kernel");
}

template <int S, int KZ, int KS>
int launch_wgrad(const WgradLaunch &l, const float *small, const float *big, float *partial, int B, hipStream_t st) {
  // 16-byte staging: grid rows that start 16-byte aligned and tensors a 32-bit buffer offset can address
  const size_t small_bytes = (size_t)B * l.Cs * l.Zs * l.Ys * l.Xs * 4;
  const size_t big_bytes = (size_t)B * l.Cb * (KZ == 1 ? 1 : l.Zs * S) * (l.Ys * S) * (l.Xs * S) * 4;
  const bool vec = l.Xs % 4 == 0 && small_bytes <= (1ull << 31) && big_bytes <= (1ull << 31) &&
                   (reinterpret_cast<size_t>(small) & 15) == 0 && (reinterpret_cast<size_t>(big) & 15) == 0;
  return vec ? launch_wgrad_v<S, KZ, KS, true>(l, small, big, partial, B, st) : launch_wgrad_v<S, KZ, KS, false>(l, small, big, partial, B, st);
}

}  // namespace

extern "C" size_t casmvs_conv_wgrad_workspace_bytes(int kind, int B, int cin, int cout, int D, int H, int W) {
  WgradLaunch l;
  if (!wgrad_launch(kind, B, cin, cout, D, H, W, l)) return 0;
  return (size_t)l.gx * l.gy * l.T * 256 * sizeof(float);
}

namespace {
int conv_wgrad_run(int kind, const float *in, const float *grad_out, float *grad_weight, void *workspace, int B, int cin, int cout, int D, int H, int W,
                   void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(in && grad_out && grad_weight && workspace, "conv_wgrad: null pointer");
  WgradLaunch l;
  if (!wgrad_launch(kind, B, cin, cout, D, H, W, l))
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "conv_wgrad: kind=%d B=%d cin=%d cout=%d input %dx%dx%d", kind, B, cin, cout, D, H, W);
  WgradGeom g;
  wgrad_geom(kind, g);
  const float *small = g.transposed ? in : grad_out, *big = g.transposed ? grad_out : in;
  float *partial = static_cast<float *>(workspace);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (g.KZ == 3 && g.S == 1) rc = launch_wgrad<1, 3, 3>(l, small, big, partial, B, st);
  else if (g.KZ == 3) rc = launch_wgrad<2, 3, 3>(l, small, big, partial, B, st);
  else if (g.KS == 3) rc = launch_wgrad<1, 1, 3>(l, small, big, partial, B, st);
  else if (g.KS == 5) rc = launch_wgrad<2, 1, 5>(l, small, big, partial, B, st);
  else rc = launch_wgrad<1, 1, 1>(l, small, big, partial, B, st);
  if (rc) return rc;
  const int n = l.gy * l.T * 256;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(n / kRedElems)), dim3(kThreads), 0, st, partial, grad_weight, l.Cs,
                     l.Cb, l.T, l.cb_groups, l.gy, l.gx);
  return casmvs::check_launch("wgrad_reduce_kernel");
}
}  // namespace

extern "C" int casmvs_conv_wgrad_f32(int kind, const float *in, const float *grad_out, float *grad_weight, void *workspace, int B,
                                     int cin, int cout, int D, int H, int W, void *stream) {
  return conv_wgrad_run(kind, in, grad_out, grad_weight, workspace, B, cin, cout, D, H, W, stream);
}


extern "C" int casmvs_conv_dgrad_direct_f32(int kind, const float *weight, const float *grad_out, float *grad_in, int B, int cin,
                                            int cout, int D, int H, int W, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && grad_out && grad_in, "conv_dgrad_direct: null pointer");
  WgradGeom g;
  if (!wgrad_geom(kind, g) || g.transposed)
    return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "conv_dgrad_direct: kind=%d (Conv2d / Conv3d kinds only)", kind);
  CASMVS_REQUIRE(B > 0 && B <= 65535 && cin > 0 && cin <= 65535 && cout > 0 && D > 0 && H > 0 && W > 0 && (g.KZ == 3 || D == 1),
                 "conv_dgrad_direct: bad shape B=%d cin=%d cout=%d %dx%dx%d", B, cin, cout, D, H, W);
  const size_t in_cs = (size_t)D * H * W;
  if (cin == 8 || cin == 16) {
    dim3 grid_all((unsigned)((in_cs + kThreads - 1) / kThreads), 1u, (unsigned)B);
    if (cin == 8)
      hipLaunchKernelGGL(conv_dgrad_direct_all_kernel<8>, grid_all, dim3(kThreads), 0, (hipStream_t)stream, grad_out, weight, grad_in, cout, D,
                         H, W, g.S, g.KZ, g.KS);
    else
      hipLaunchKernelGGL(conv_dgrad_direct_all_kernel<16>, grid_all, dim3(kThreads), 0, (hipStream_t)stream, grad_out, weight, grad_in, cout, D,
                         H, W, g.S, g.KZ, g.KS);
    return casmvs::check_launch("conv_dgrad_direct_all_kernel");
  }
  dim3 grid((unsigned)((in_cs + kThreads - 1) / kThreads), (unsigned)cin, (unsigned)B);
  hipLaunchKernelGGL(conv_dgrad_direct_kernel, grid, dim3(kThreads), 0, (hipStream_t)stream, grad_out, weight, grad_in, cin, cout, D, H, W,
                     g.S, g.KZ, g.KS);
  return casmvs::check_launch("conv_dgrad_direct_kernel");
}

namespace {
int sums_blocks(int N, size_t n) {
  const size_t total = (size_t)N * n;
  size_t b = (total + (size_t)kThreads * 64 - 1) / ((size_t)kThreads * 64);
  if (b < 1) b = 1;
  if (b > 256) b = 256;
  return (int)b;
}
}  // namespace

extern "C" int casmvs_channel_sums_blocks(int N, size_t n) { return sums_blocks(N, n); }

// out: (C, blocks, 2) doubles with blocks = casmvs_channel_sums_blocks(N, n): sum x, sum x^2 over this block's slice
extern "C" int casmvs_channel_sums_f64(const float *x, double *out, int N, int C, size_t n, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(x && out && N > 0 && C > 0 && C <= 65535 && n > 0, "channel_sums: bad arguments");
  hipLaunchKernelGGL(channel_sums_kernel<0>, dim3((unsigned)sums_blocks(N, n), (unsigned)C), dim3(kThreads), 0, (hipStream_t)stream, x,
                     nullptr, nullptr, nullptr, nullptr, out, N, C, n, 0.0f);
  return casmvs::check_launch("channel_sums_kernel");
}

extern "C" int casmvs_abn_apply_f32(const float *x, const float *scale, const float *shift, float *y, int N, int C, size_t n, float slope,
                                    void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(x && scale && shift && y && N > 0 && C > 0 && (size_t)N * C <= 65535 && n > 0, "abn_apply: bad arguments");
  hipLaunchKernelGGL(abn_apply_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads), (unsigned)(N * C)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, scale, shift, y, C, n, slope);
  return casmvs::check_launch("abn_apply_kernel");
}

// sums: (C, blocks, 2) doubles: sum g, sum g * xhat  (g = grad_y * lrelu'(y), xhat = (x - mean) * rstd)
extern "C" int casmvs_abn_backward_sums_f64(const float *grad_y, const float *y, const float *x, const float *mean, const float *rstd,
                                            double *sums, int N, int C, size_t n, float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(grad_y && y && x && mean && rstd && sums && N > 0 && C > 0 && C <= 65535 && n > 0, "abn_backward_sums: bad arguments");
  hipLaunchKernelGGL(channel_sums_kernel<1>, dim3((unsigned)sums_blocks(N, n), (unsigned)C), dim3(kThreads), 0, (hipStream_t)stream, grad_y, y,
                     x, mean, rstd, sums, N, C, n, slope);
  return casmvs::check_launch("channel_sums_kernel<1>");
}

extern "C" int casmvs_abn_backward_apply_f32(const float *grad_y, const float *y, const float *x, const float *scale, const float *mean,
                                             const float *rstd, const float *m1, const float *m2, float *grad_x, int N, int C, size_t n,
                                             float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(grad_y && y && x && scale && mean && rstd && m1 && m2 && grad_x && N > 0 && C > 0 && (size_t)N * C <= 65535 && n > 0,
                 "abn_backward_apply: bad arguments");
  hipLaunchKernelGGL(abn_bwd_apply_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads), (unsigned)(N * C)), dim3(kThreads), 0,
                     (hipStream_t)stream, grad_y, y, x, scale, mean, rstd, m1, m2, grad_x, C, n, slope);
  return casmvs::check_launch("abn_bwd_apply_kernel");
}

extern "C" int casmvs_pack_gather_f32(const float *weight, const float *bias, const int *index, float *out, int n_weight, int n_bias,
                                      int n_out, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(weight && index && out && n_weight > 0 && n_bias >= 0 && (n_bias == 0 || bias) && n_out > 0,
                 "pack_gather: bad arguments");
  hipLaunchKernelGGL(pack_gather_kernel, dim3((unsigned)casmvs::ceil_div(n_out, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, weight,
                     bias, index, out, n_weight, n_bias, n_out);
  return casmvs::check_launch("pack_gather_kernel");
}

static_assert(sizeof(PackSegment) == sizeof(casmvs_pack_segment), "PackSegment mirrors casmvs_pack_segment");
extern "C" int casmvs_pack_gather_batch_f32(const casmvs_pack_segment *segments, int n_segments, int n_blocks, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(segments && n_segments > 0 && n_blocks > 0, "pack_gather_batch: bad arguments");
  hipLaunchKernelGGL(pack_gather_batch_kernel, dim3((unsigned)n_blocks), dim3(kThreads), 0, (hipStream_t)stream,
                     reinterpret_cast<const PackSegment *>(segments), n_segments);
  return casmvs::check_launch("pack_gather_batch_kernel");
}

// sums: (C, blocks, 2) doubles from casmvs_channel_sums_f64; every output a device vector of C floats
extern "C" int casmvs_abn_train_finish_f32(const double *sums, int blocks, int C, double count, const float *weight, const float *bias,
                                           float abs_eps, float eps, float momentum, float *running_mean, float *running_var,
                                           float *scale, float *shift, float *mean, float *rstd, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(sums && weight && bias && scale && shift && mean && rstd && blocks > 0 && C > 0 && count > 0.0 &&
                     (running_mean != nullptr) == (running_var != nullptr),
                 "abn_train_finish: bad arguments");
  hipLaunchKernelGGL(abn_train_finish_kernel, dim3((unsigned)casmvs::ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, blocks, C, count,
                     weight, bias, abs_eps, eps, momentum, running_mean, running_var, scale, shift, mean, rstd);
  return casmvs::check_launch("abn_train_finish_kernel");
}

// casmvs_abn_train_finish_f32 + casmvs_abn_apply_f32 as one launch
extern "C" int casmvs_abn_train_apply_f32(const float *x, const double *sums, int blocks, double count, const float *weight, const float *bias,
                                          float abs_eps, float eps, float momentum, float *running_mean, float *running_var, float *scale,
                                          float *shift, float *mean, float *rstd, float *y, int N, int C, size_t n, float slope, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(x && y && sums && weight && bias && scale && shift && mean && rstd && blocks > 0 && count > 0.0 && N > 0 && C > 0 &&
                     (size_t)N * C <= 65535 && n > 0 && (running_mean != nullptr) == (running_var != nullptr),
                 "abn_train_apply: bad arguments");
  hipLaunchKernelGGL(abn_train_apply_kernel, dim3((unsigned)((n + kAbnChunk - 1) / kAbnChunk), (unsigned)(N * C)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, sums, blocks, count, weight, bias, abs_eps, eps, momentum, running_mean, running_var, scale, shift, mean,
                     rstd, y, C, n, slope);
  return casmvs::check_launch("abn_train_apply_kernel");
}

// casmvs_abn_backward_finish_f32 + casmvs_abn_backward_apply_f32 as one launch (m1 / m2 stay inside the kernel)
extern "C" int casmvs_abn_backward_apply_fused_f32(const float *grad_y, const float *y, const float *x, const double *sums, int blocks, double count,
                                                   const float *weight, float abs_eps, const float *scale, const float *mean, const float *rstd,
                                                   float *grad_weight, float *grad_bias, float *grad_x, int N, int C, size_t n, float slope,
                                                   void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(grad_y && y && x && sums && weight && scale && mean && rstd && grad_weight && grad_bias && grad_x && blocks > 0 && count > 0.0 &&
                     N > 0 && C > 0 && (size_t)N * C <= 65535 && n > 0,
                 "abn_backward_apply_fused: bad arguments");
  hipLaunchKernelGGL(abn_bwd_apply_stats_kernel, dim3((unsigned)((n + kAbnChunk - 1) / kAbnChunk), (unsigned)(N * C)), dim3(kThreads), 0,
                     (hipStream_t)stream, grad_y, y, x, sums, blocks, count, weight, abs_eps, scale, mean, rstd, grad_weight, grad_bias, grad_x, C, n,
                     slope);
  return casmvs::check_launch("abn_bwd_apply_stats_kernel");
}

// sums: (C, blocks, 2) doubles from casmvs_abn_backward_sums_f64
extern "C" int casmvs_abn_backward_finish_f32(const double *sums, int blocks, int C, double count, const float *weight, float abs_eps,
                                              float *grad_weight, float *grad_bias, float *m1, float *m2, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(sums && weight && grad_weight && grad_bias && m1 && m2 && blocks > 0 && C > 0 && count > 0.0,
                 "abn_backward_finish: bad arguments");
  hipLaunchKernelGGL(abn_bwd_finish_kernel, dim3((unsigned)casmvs::ceil_div(C, 64)), dim3(64), 0, (hipStream_t)stream, sums, blocks, C, count,
                     weight, abs_eps, grad_weight, grad_bias, m1, m2);
  return casmvs::check_launch("abn_bwd_finish_kernel");
}

extern "C" int casmvs_upsample2x_add_f32(const float *lat, const float *up, float *out, int N, int C, int H, int W, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(lat && up && out && N > 0 && C > 0 && (size_t)N * C <= 65535 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
                 "upsample2x_add: bad arguments");
  hipLaunchKernelGGL(upsample2x_add_kernel, dim3((unsigned)casmvs::ceil_div(H * W, kThreads), (unsigned)(N * C)), dim3(kThreads), 0,
                     (hipStream_t)stream, lat, up, out, H, W);
  return casmvs::check_launch("upsample2x_add_kernel");
}

extern "C" int casmvs_upsample2x_backward_f32(const float *grad_out, float *grad_up, int N, int C, int H, int W, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(grad_out && grad_up && N > 0 && C > 0 && (size_t)N * C <= 65535 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0,
                 "upsample2x_backward: bad arguments");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3((unsigned)casmvs::ceil_div((H / 2) * (W / 2), kThreads), (unsigned)(N * C)), dim3(kThreads), 0,
                     (hipStream_t)stream, grad_out, grad_up, H, W);
  return casmvs::check_launch("upsample2x_bwd_kernel");
}

namespace {
// workspace of the volume backward: [acc: B V C h w 64-bit fixed-point sums][gmax: B (G | C) uint32][fmax: B C uint32][partial maxima of volume_absmax_kernel]
struct VolBwdWs {
  size_t acc_bytes, gmax_off, fmax_off, partial_off, total;
};
VolBwdWs volume_backward_ws(int B, int V, int C, int G, int D, int h, int w) {
  VolBwdWs l;
  auto pad = [](size_t n) { return (n + 15) & ~(size_t)15; };
  l.acc_bytes = (size_t)B * V * C * h * w * sizeof(unsigned long long);
  l.gmax_off = l.acc_bytes;
  l.fmax_off = l.gmax_off + pad((size_t)B * (G > 0 ? G : C) * sizeof(unsigned));
  l.partial_off = l.fmax_off + pad((size_t)B * C * sizeof(unsigned));
  l.total = l.partial_off + pad(((size_t)B * (G > 0 ? G : C) * absmax_chunks((size_t)D * h * w) + (size_t)B * V * C * absmax_chunks((size_t)h * w)) * sizeof(unsigned));
  return l;
}
// the variance volume's (G = 0) or the correlation volume's (G > 0) gradient w.r.t. the feature maps: zeroing, the channels' largest magnitudes, the
// scatter into the fixed-point map, the map -> float32
int volume_backward(const float *feats, const float *proj, const float *depth, const float *grad_vol, float *grad_feats, void *workspace, int B, int V,
                    int C, int G, int h, int w, int D, void *stream, const char *what) {
  CASMVS_REQUIRE(feats && proj && depth && grad_vol && grad_feats && workspace, "%s: null pointer", what);
  CASMVS_REQUIRE(B > 0 && B <= 65535 && V >= 2 && V <= 64 && h > 1 && w > 1 && D > 0, "%s: bad shape B=%d V=%d h=%d w=%d D=%d", what, B, V, h, w, D);
  CASMVS_REQUIRE(C % 4 == 0 && C >= 4 && C <= 64, "%s: C=%d (a multiple of 4 up to 64)", what, C);
  CASMVS_REQUIRE(G == 0 || (G > 0 && C % G == 0), "%s: G=%d does not divide C=%d", what, G, C);
  CASMVS_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "%s: workspace must be 16-byte aligned", what);
  CASMVS_REQUIRE((size_t)B * V * C <= 65535, "%s: B V C = %zu rows", what, (size_t)B * V * C);
  hipStream_t st = (hipStream_t)stream;
  const VolBwdWs ws = volume_backward_ws(B, V, C, G, D, h, w);
  hipError_t e = hipMemsetAsync(grad_feats, 0, (size_t)B * V * C * h * w * sizeof(float), st);
  if (e == hipSuccess) e = hipMemsetAsync(workspace, 0, ws.acc_bytes, st);
  if (e != hipSuccess) return casmvs::fail(CASMVS_ERR_HIP, "%s: hipMemsetAsync: %s", what, hipGetErrorString(e));
  unsigned long long *acc = static_cast<unsigned long long *>(workspace);
  unsigned *gmax = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + ws.gmax_off);
  unsigned *fmax = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + ws.fmax_off);
  const int hw = h * w, U = casmvs::fixed_point_bits(D, h, w);
  unsigned *partial = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + ws.partial_off);
  if (int rc = launch_volume_absmax(grad_vol, feats, partial, gmax, fmax, B * (G > 0 ? G : C), (size_t)D * hw, B, V, C, (size_t)hw, st)) return rc;
  // 8 planes per workgroup; the LDS image holds 1152 box pixels (a 32 x 16 tile whose taps spread over ~44 x 26) of 4 channels
  // in 64-bit fixed point = 36 KiB: four workgroups per CU (the kernel waits on its gathers: 0.92 -> 0.43 ms at level 1 from two to four)
  constexpr int dch = 8, cap = 1152, th = 16, CG = 4;
  const int tiles_x = casmvs::ceil_div(w, 32), tiles_y = casmvs::ceil_div(h, th), chunks = casmvs::ceil_div(D, dch);
  const long gy = (long)chunks * (C / CG) * (V - 1);
  CASMVS_REQUIRE(gy <= 65535, "%s: D=%d C=%d V=%d: too many (plane chunk, channel group, view) items", what, D, C, V);
  auto kernel = G > 0 ? (V == 3 ? costvol_var_bwd_kernel<CG, th, 2, true> : costvol_var_bwd_kernel<CG, th, 0, true>)
                      : (V == 3 ? costvol_var_bwd_kernel<CG, th, 2, false> : costvol_var_bwd_kernel<CG, th, 0, false>);
  const size_t lds = (size_t)CG * cap * sizeof(unsigned long long);
  if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), lds, "costvol_var_bwd_kernel")) return rc;
  dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)gy, (unsigned)B);
  hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, st, feats, proj, depth, grad_vol, grad_feats, acc, gmax, fmax, V, C, h, w, D, tiles_x, dch, cap, G, U);
  if (int rc = casmvs::check_launch("costvol_var_bwd_kernel")) return rc;
  dim3 fgrid((unsigned)std::min(casmvs::ceil_div(hw, kThreads), 64), (unsigned)(B * V * C));
  if (G > 0) hipLaunchKernelGGL(costvol_fixed_finish_kernel<true>, fgrid, dim3(kThreads), 0, st, acc, gmax, fmax, grad_feats, V, C, G, hw, U);
  else hipLaunchKernelGGL(costvol_fixed_finish_kernel<false>, fgrid, dim3(kThreads), 0, st, acc, gmax, fmax, grad_feats, V, C, G, hw, U);
  return casmvs::check_launch("costvol_fixed_finish_kernel");
}
}  // namespace

extern "C" size_t casmvs_costvol_backward_workspace_bytes(int B, int V, int C, int G, int D, int h, int w) {
  if (B < 1 || V < 2 || C < 1 || G < 0 || D < 1 || h < 1 || w < 1) return 0;
  return volume_backward_ws(B, V, C, G, D, h, w).total;
}

extern "C" int casmvs_costvol_var_backward_f32(const float *feats, const float *proj, const float *depth, const float *grad_vol,
                                               float *grad_feats, void *workspace, int B, int V, int C, int h, int w, int D, void *stream) {
  casmvs::clear_error();
  return volume_backward(feats, proj, depth, grad_vol, grad_feats, workspace, B, V, C, 0, h, w, D, stream, "costvol_var_backward");
}

extern "C" int casmvs_costvol_gwc_backward_f32(const float *feats, const float *proj, const float *depth, const float *grad_vol,
                                               float *grad_feats, void *workspace, int B, int V, int C, int G, int h, int w, int D, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(G > 0, "costvol_gwc_backward: G=%d", G);
  return volume_backward(feats, proj, depth, grad_vol, grad_feats, workspace, B, V, C, G, h, w, D, stream, "costvol_gwc_backward");
}

