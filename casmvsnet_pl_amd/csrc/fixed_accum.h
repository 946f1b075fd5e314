// Order-independent accumulation for the scatter kernels of the training path (train.hip: the cost volume's backward; backward.hip: homo_warp's backward).
// Float atomics make a sum depend on the order the hardware happens to retire them in - 18 runs of the same 12 SGD steps gave 18 loss trajectories in round 4.
// Here every scattered sum is a 64-bit INTEGER in fixed point (integer addition is associative), with one scale per (sample, channel) that is known before
// the scatter starts: 2^be >= a strict bound of the channel's largest contribution (from the largest finite magnitudes of its operands, found by
// volume_absmax_kernel with integer atomicMax on the bit patterns - order-independent as well), unit 2^(be - U).  U leaves room for every contribution that
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

// 2^be >= factor G F (bound = m 2^be, m in [0.5, 1)); false when the bound is 0 (no finite non-zero contribution exists).  The product of two float32
// magnitudes (denormals included) and a small factor is a NORMAL double: no range to leave.
__device__ __forceinline__ bool fixed_exponent(unsigned gbits, unsigned fbits, double factor, int &be) {
  const double bound = factor * (double)__builtin_bit_cast(float, gbits) * (double)__builtin_bit_cast(float, fbits);
  be = (int)((__builtin_bit_cast(unsigned long long, bound) >> 52) & 0x7ffull) - 1022;
  return bound > 0.0;
}
__device__ __forceinline__ double pow2_double(int e) { return __builtin_bit_cast(double, (unsigned long long)(1023 + e) << 52); }   // |e| < 1000

// rows [0, rows_g): the upstream gradient's (sample, channel | group) volumes of n_g floats -> gmax[row]; rows [rows_g, rows_g + B V C): the feature maps'
// (sample, view, channel) planes of n_f floats -> fmax[b C + c] (the largest over the views).  Both zeroed by the caller.
static __global__ __launch_bounds__(256) void volume_absmax_kernel(const float *__restrict__ gvol, const float *__restrict__ feats, unsigned *__restrict__ gmax,
                                                                 unsigned *__restrict__ fmax, int rows_g, size_t n_g, int chunks_g, int V, int C, size_t n_f,
                                                                 int chunks_f) {
  constexpr int kPer = 32, kThreads = 256;   // floats per thread
  const int wg = blockIdx.x;
  const bool is_g = wg < rows_g * chunks_g;
  const int r = is_g ? wg : wg - rows_g * chunks_g, per = is_g ? chunks_g : chunks_f;
  const int row = r / per, chunk = r - row * per;
  const size_t n = is_g ? n_g : n_f;
  const float *src = (is_g ? gvol : feats) + (size_t)row * n;
  const size_t e0 = (size_t)chunk * kThreads * kPer;
  unsigned m = 0u;
  if ((n & 3) == 0 && (reinterpret_cast<size_t>(src) & 15) == 0) {
#pragma unroll
    for (int i = 0; i < kPer / 4; ++i) {
      const size_t e = e0 + ((size_t)i * kThreads + threadIdx.x) * 4;
      if (e < n) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(src + e);
        m = max(max(m, finite_abs_bits(v[0])), max(finite_abs_bits(v[1]), max(finite_abs_bits(v[2]), finite_abs_bits(v[3]))));
      }
    }
  } else {
#pragma unroll 4
    for (int i = 0; i < kPer; ++i) {
      const size_t e = e0 + (size_t)i * kThreads + threadIdx.x;
      if (e < n) m = max(m, finite_abs_bits(src[e]));
    }
  }
  m = casmvs::wave_max_bits(m);
  if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(is_g ? gmax + row : fmax + (size_t)(row / (V * C)) * C + row % C, m);
}


// round(val * scale) as a two's-complement integer; |val * scale| < 2^51 (1.5 2^52: the low mantissa bits of x + kMagic are round(x))
__device__ __forceinline__ unsigned long long to_fixed_point(float val, double scale) {
  constexpr double kMagic = 6755399441055744.0;
  return (unsigned long long)(__builtin_bit_cast(long long, __builtin_fma((double)val, scale, kMagic)) - __builtin_bit_cast(long long, kMagic));
}

}  // namespace casmvs_dev
