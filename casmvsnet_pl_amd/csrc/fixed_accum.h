// Order-independent accumulation for the scatter kernels of the training path (train.hip: the cost volume's backward; backward.hip: homo_warp's backward).
// Float atomics make a sum depend on the order the hardware happens to retire them in - 18 runs of the same 12 SGD steps gave 18 loss trajectories in round 4.
// Here every scattered sum is a 64-bit INTEGER in fixed point (integer addition is associative), with one scale per (sample, channel) that is known before
// the scatter starts: 2^be >= a strict bound of the channel's largest contribution (from the largest finite magnitudes of its operands, found by
// volume_absmax_kernel + volume_absmax_reduce_kernel: a tree of maxima, no atomics), unit 2^(be - U).  U leaves room for every contribution that
// can reach one cell: no overflow by construction, no range check, no fallback.  Non-finite contributions (which cannot be integers) are added with float
// atomics to the float32 OUTPUT map - a sum of non-finite values is non-finite in any order - and the finishing pass adds the fixed-point sum, rounded once,
// to it: exactly the elements a float accumulation would make non-finite are non-finite.
#pragma once
#include "common.h"
#include "plane_sweep.h"
#include "split_f16.h"

namespace casmvs {
// U of the fixed-point unit 2^(be - U): at most D h w contributions below 2^(U - 1) units reach one cell; 44: the magic-number conversion below is exact
// for |x| < 2^51 and a register-accumulated sum (the reference view's gradient over a chunk of planes) stays below 2^7 bounds
inline int fixed_point_bits(int D, int h, int w) {
  const unsigned long long n = (unsigned long long)D * h * w;
  int lg = 0;
  while ((1ull << lg) < n) ++lg;
  const int u = 62 - lg;
  return u < 44 ? u : 44;
}
}  // namespace casmvs

namespace casmvs_dev {

// |x| as a bit pattern when x is finite, 0 otherwise: unsigned order = order of the magnitudes
__device__ __forceinline__ unsigned finite_abs_bits(float x) {
  const unsigned b = __builtin_bit_cast(unsigned, x) & 0x7fffffffu;
  return b < 0x7f800000u ? b : 0u;
}
__device__ __forceinline__ bool is_finite(float x) { return (__builtin_bit_cast(unsigned, x) & 0x7fffffffu) < 0x7f800000u; }

// gword / fword: maximum words (volume_absmax_kernel).  2^be >= factor G F (bound = m 2^be, m in [0.5, 1)); false when the bound is 0 (no finite non-zero
// contribution exists).  The product of two float32 magnitudes (denormals included) and a small factor is a NORMAL double: no range to leave.
__device__ __forceinline__ bool fixed_exponent(unsigned gword, unsigned fword, double factor, int &be) {
  const double bound = factor * (double)__builtin_bit_cast(float, gword & 0x7fffffffu) * (double)__builtin_bit_cast(float, fword & 0x7fffffffu);
  be = (int)((__builtin_bit_cast(unsigned long long, bound) >> 52) & 0x7ffull) - 1022;
  return bound > 0.0;
}
// Can a contribution of this (sample, channel) be non-finite?  Yes when an operand is (flag bits), or when finite operands could overflow float32 on the way
// (a bound or a feature beyond 2^120: the kernels' intermediates stay within a few powers of two of them).
__device__ __forceinline__ bool needs_finite_check(unsigned gword, unsigned fword, int be) {
  return ((gword | fword) >> 31) != 0u || be > 120 || (fword & 0x7fffffffu) > 0x7b800000u;
}
__device__ __forceinline__ double pow2_double(int e) { return __builtin_bit_cast(double, (unsigned long long)(1023 + e) << 52); }   // |e| < 1000

// Largest finite magnitudes, two launches and NO atomics (a first version let every wave atomicMax the row's word: 1 400 same-address atomics per row
// serialised at the memory side - 160-250 us per call where the loads need 40):
// volume_absmax_kernel: workgroup = 8192 consecutive floats of one row -> partial[workgroup].  rows [0, rows_g): the upstream gradient's (sample,
//   channel | group) volumes of n_g floats; rows [rows_g, rows_g + rows_f): the feature maps' (sample, view, channel) planes of n_f floats.
// volume_absmax_reduce_kernel: one wave per output word - gmax[row] = max of the row's chunks_g partials; fmax[b C + c] = max over the V views' chunks.
// A maximum word: bits 0 .. 30 = the largest FINITE magnitude's bit pattern, bit 31 = the row holds a non-finite value (the scatter kernels then take the
// path that checks every contribution; rows without one - every row of a healthy training run - never pay for the check).
constexpr int kAbsmaxPer = 32, kAbsmaxThreads = 256;   // floats per thread: eight 16-byte loads in flight
__device__ __forceinline__ unsigned absmax_word(unsigned finite_bits, bool non_finite) { return finite_bits | (non_finite ? 0x80000000u : 0u); }
__device__ __forceinline__ unsigned absmax_combine(unsigned finite_bits, bool non_finite) { return absmax_word(finite_bits, non_finite); }
__device__ __forceinline__ unsigned absmax_merge(unsigned a, unsigned b) { return max(a & 0x7fffffffu, b & 0x7fffffffu) | ((a | b) & 0x80000000u); }
static __global__ __launch_bounds__(256) void volume_absmax_kernel(const float *__restrict__ gvol, const float *__restrict__ feats, unsigned *__restrict__ partial,
                                                                   int rows_g, size_t n_g, int chunks_g, size_t n_f, int chunks_f) {
  constexpr int kPer = kAbsmaxPer, kThreads = kAbsmaxThreads;
  __shared__ unsigned wmax[kThreads / 64];
  const int wg = blockIdx.x;
  const bool is_g = wg < rows_g * chunks_g;
  const int r = is_g ? wg : wg - rows_g * chunks_g, per = is_g ? chunks_g : chunks_f;
  const int row = r / per, chunk = r - row * per;
  const size_t n = is_g ? n_g : n_f;
  const float *src = (is_g ? gvol : feats) + (size_t)row * n;
  const size_t e0 = (size_t)chunk * kThreads * kPer;
  unsigned m = 0u;   // the largest |x| bit pattern, non-finite values included (they order above every finite one)
  if ((n & 3) == 0 && (reinterpret_cast<size_t>(src) & 15) == 0 && e0 + (size_t)kThreads * kPer <= n) {   // a whole chunk: every load issued before the first use
    f32x4 v[kPer / 4];
#pragma unroll
    for (int i = 0; i < kPer / 4; ++i) v[i] = *reinterpret_cast<const f32x4 *>(src + e0 + ((size_t)i * kThreads + threadIdx.x) * 4);
    unsigned f = 0u;
#pragma unroll
    for (int i = 0; i < kPer / 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned b = __builtin_bit_cast(unsigned, v[i][j]) & 0x7fffffffu;
        m = max(m, b);                       // all magnitudes ...
        f = max(f, finite_abs_bits(v[i][j]));   // ... and the finite ones
      }
    m = absmax_word(f, m >= 0x7f800000u);
  } else {
    unsigned f = 0u;
    for (int i = 0; i < kPer; ++i) {
      const size_t e = e0 + (size_t)i * kThreads + threadIdx.x;
      if (e < n) {
        m = max(m, __builtin_bit_cast(unsigned, src[e]) & 0x7fffffffu);
        f = max(f, finite_abs_bits(src[e]));
      }
    }
    m = absmax_word(f, m >= 0x7f800000u);
  }
  m = absmax_combine(casmvs::wave_max_bits(m & 0x7fffffffu), __builtin_amdgcn_ballot_w64((m >> 31) != 0u) != 0);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[wg] = absmax_merge(absmax_merge(wmax[0], wmax[1]), absmax_merge(wmax[2], wmax[3]));
}

static __global__ __launch_bounds__(64) void volume_absmax_reduce_kernel(const unsigned *__restrict__ partial, unsigned *__restrict__ gmax, unsigned *__restrict__ fmax,
                                                                         int rows_g, int chunks_g, int V, int C, int chunks_f) {
  const int o = blockIdx.x, lane = threadIdx.x;
  unsigned m = 0u;
  if (o < rows_g) {
    for (int k = lane; k < chunks_g; k += 64) m = absmax_merge(m, partial[(size_t)o * chunks_g + k]);
  } else {
    const int bc = o - rows_g, b = bc / C, c = bc - b * C;
    const unsigned *pf = partial + (size_t)rows_g * chunks_g;
    for (int k = lane; k < V * chunks_f; k += 64) {
      const int v = k / chunks_f, ch = k - v * chunks_f;
      m = absmax_merge(m, pf[((size_t)(b * V + v) * C + c) * chunks_f + ch]);
    }
  }
  m = absmax_combine(casmvs::wave_max_bits(m & 0x7fffffffu), __builtin_amdgcn_ballot_w64((m >> 31) != 0u) != 0);
  if (lane == 0) (o < rows_g ? gmax + o : fmax + (o - rows_g))[0] = m;
}

// host: both launches; partial: (rows_g chunks_g + rows_f chunks_f) words; fmax may be NULL when rows_f == 0 (V = 0)
inline size_t absmax_chunks(size_t n) { return (n + (size_t)kAbsmaxThreads * kAbsmaxPer - 1) / ((size_t)kAbsmaxThreads * kAbsmaxPer); }
inline int launch_volume_absmax(const float *gvol, const float *feats, unsigned *partial, unsigned *gmax, unsigned *fmax, int rows_g, size_t n_g, int B, int V, int C,
                                size_t n_f, hipStream_t st) {
  const size_t chunks_g = absmax_chunks(n_g), chunks_f = V > 0 ? absmax_chunks(n_f) : 0;
  const size_t wgs = (size_t)rows_g * chunks_g + (size_t)B * V * C * chunks_f;
  if (wgs > 0x7fffffffull) return casmvs::fail(CASMVS_ERR_UNSUPPORTED, "volume_absmax: volume too large");
  hipLaunchKernelGGL(volume_absmax_kernel, dim3((unsigned)wgs), dim3(kAbsmaxThreads), 0, st, gvol, feats, partial, rows_g, n_g, (int)chunks_g, n_f, (int)chunks_f);
  if (int rc = casmvs::check_launch("volume_absmax_kernel")) return rc;
  hipLaunchKernelGGL(volume_absmax_reduce_kernel, dim3((unsigned)(rows_g + (V > 0 ? B * C : 0))), dim3(64), 0, st, partial, gmax, fmax, rows_g, (int)chunks_g, V, C,
                     (int)chunks_f);
  return casmvs::check_launch("volume_absmax_reduce_kernel");
}

// round(val * scale) as a two's-complement integer; |val * scale| < 2^51 (1.5 2^52: the low mantissa bits of x + kMagic are round(x))
__device__ __forceinline__ unsigned long long to_fixed_point(float val, double scale) {
  constexpr double kMagic = 6755399441055744.0;
  return (unsigned long long)(__builtin_bit_cast(long long, __builtin_fma((double)val, scale, kMagic)) - __builtin_bit_cast(long long, kMagic));
}

}  // namespace casmvs_dev
