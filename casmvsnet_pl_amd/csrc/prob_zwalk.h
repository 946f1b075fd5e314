// The multiply phase of the `prob` head (Conv3d 8 -> 1, k3 p1: models/mvsnet.py:89,104) walking the depth axis: shared by prob_regress.hip (input planes
// staged from memory) and conv11_prob_zfused.hip (input planes produced in LDS by the transposed convolution in front of it).
#pragma once
#include <type_traits>

#include "buffer_ops.h"

#ifndef CASMVS_PZ_ABL
#define CASMVS_PZ_ABL 0   // profiling builds only (WRONG results): 1 no FMAs, 2 no LDS tap reads, 4 no global loads, 8 no LDS staging
#endif                    // writes, 16 no barriers, 32 no scalar weight loads (constants)

namespace casmvs {
namespace pz {

using namespace casmvs::buf;

// Contribution of the staged plane to the accumulators: A[2 - kz] += sum_{pair, ky, kx} in * w[kz][ky][kx] for the
// kz in KZM (bit mask).  `rows`: this lane's first row / position inside the slot; wpk: the P1 weight image
// ([pair][tap (27 + 5 zeros)][channel of the pair], conv3d_mfma.hip pack_weight) read as wave-uniform scalars.
// SP / RS: floats per channel pair of a plane slot / per staged row (ProbZCfg; the slot layout [pair][row][position][channel of the pair]).
// WLDS: `wpk` is an LDS image [step (pair, ky)][20 floats: (kz, kx) -> (even, odd channel) at 2 (3 kz + kx), 2 unused] read with wave-uniform
// 16-byte LDS loads into vector registers instead of scalar loads from memory: LDS returns in order, so the step's wait is a COUNTED wait that leaves the
// next step's reads in flight (a scalar load's wait is always a full drain), and the 36 scalar registers of the two weight buffers are free (conv11_prob_zfused.hip).
template <int KZM, int SP, int RS, bool WLDS = false>
__device__ __forceinline__ void zwalk_plane(const float *rows, const float *__restrict__ wpk, f32x2 (&A)[3][2]) {
  constexpr int NSTEP = 4 * 3;  // step i = (pair i / 3, ky = i % 3); 8 input channels = 4 pairs
  // Software pipeline, pinned with sched_barrier: the two LDS rows AND the scalar weight loads of step i + 1 are issued
  // before the FMAs of step i, so that the one wait a step needs (lgkmcnt(0): scalar loads return out of order, any wait
  // on them is a full drain) finds everything landed.  (First version: the compiler issued each step's s_load / ds_read
  // right in front of its own wait - 12 exposed scalar-cache + LDS latencies per plane with 72 FMA cycles between them:
  // the kernel ran at a third of its VALU time whatever the chunking or the prefetch depth.)
  f32x4v lo[2], hi[2];
  f32x2 W[2][3][3];  // [buffer][kz][kx]: (even, odd channel) weights of the step
  auto fetch = [&](auto buf_, int i) {
    constexpr int BUF = decltype(buf_)::value;
    const float *row = rows + (i / 3) * SP + (i % 3) * RS;
    if (CASMVS_PZ_ABL & 2) {
      lo[BUF] = f32x4v{1.f, 2.f, 3.f, (float)i};
      hi[BUF] = lo[BUF];
    } else {
      lo[BUF] = *reinterpret_cast<const f32x4v *>(row);
      hi[BUF] = *reinterpret_cast<const f32x4v *>(row + 4);
    }
    if constexpr (WLDS) {
      const f32x4v *wv = reinterpret_cast<const f32x4v *>(wpk + i * 20);
      f32x4v q[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) q[j] = wv[j];
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        if (!((KZM >> kz) & 1)) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int e = 2 * (3 * kz + kx);
          W[BUF][kz][kx] = f32x2{q[e / 4][e % 4], q[e / 4][e % 4 + 1]};
        }
      }
      return;
    }
    const float *wq = wpk + (i / 3) * 64 + (i % 3) * 6;  // taps (kz, ky, kx = 0..2) x (even, odd channel) at [kz * 18 + 2 kx + c]
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      if (!((KZM >> kz) & 1)) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        W[BUF][kz][kx] = (CASMVS_PZ_ABL & 32) ? f32x2{0.5f + kz, 0.25f * kx} : f32x2{wq[kz * 18 + 2 * kx], wq[kz * 18 + 2 * kx + 1]};
    }
  };
  auto fmas = [&](auto buf_) {
    constexpr int BUF = decltype(buf_)::value;
    const f32x2 P[4] = {f32x2{lo[BUF][0], lo[BUF][1]}, f32x2{lo[BUF][2], lo[BUF][3]}, f32x2{hi[BUF][0], hi[BUF][1]}, f32x2{hi[BUF][2], hi[BUF][3]}};
    if (CASMVS_PZ_ABL & 1) {   // keep the operands live without the 18 FMAs
      A[1][0] = A[1][0] + P[0] + P[3];
      if (KZM & 1) A[2][0] = A[2][0] + W[BUF][0][0];
      if (KZM & 4) A[0][0] = A[0][0] + W[BUF][2][2];
      return;
    }
    // tap by tap over the (up to) six independent accumulators (kz, pixel): no two consecutive FMAs depend on each other
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        if (!((KZM >> kz) & 1)) continue;
        A[2 - kz][0] = __builtin_elementwise_fma(P[kx], W[BUF][kz][kx], A[2 - kz][0]);
        A[2 - kz][1] = __builtin_elementwise_fma(P[kx + 1], W[BUF][kz][kx], A[2 - kz][1]);
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

}  // namespace pz
}  // namespace casmvs
