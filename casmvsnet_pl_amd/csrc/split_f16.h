// The two-slice float16 split shared by the split-f16 kernels (conv0_splitf16.hip, conv_ci_splitf16.hip, fpn_fused_sf.hip):
// x' = x * mult (mult an exact power of two), a = f16(x') (round to nearest even), b = f16(x' - a) with x' - a exact.
// Six VALU operations per PAIR of values: two multiplies, v_cvt_pk_f16_f32 for (a0, a1), two v_fma_mix_f32 that read a's halves as
// float16 sources (r = x * mult - a, exact: x * mult is exact and the difference is representable), v_cvt_pk_f16_f32 for (b0, b1).
// (The compiler's own code for the C++ form re-derived each a as a second, scalar conversion and converted it back: 10 per pair.)
#pragma once
#include <hip/hip_runtime.h>

namespace casmvs {

typedef _Float16 split_f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned split_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_pair_f16(float x0, float x1, float mult, unsigned &a_bits, unsigned &b_bits) {
  const float s0 = x0 * mult, s1 = x1 * mult;                     // exact
  const split_f16x2 a = {(_Float16)s0, (_Float16)s1};             // round to nearest even
  const unsigned ab = __builtin_bit_cast(unsigned, a);
  float r0, r1;
#ifdef CASMVS_SPLIT_NOASM   // debug builds: the plain C++ form (10 operations per pair)
  r0 = s0 - (float)a[0];
  r1 = s1 - (float)a[1];
#else
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(mult), "v"(ab));                 // x0 * mult - a.lo
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(mult), "v"(ab));   // x1 * mult - a.hi
#endif
  const split_f16x2 b = {(_Float16)r0, (_Float16)r1};
  a_bits = ab;
  b_bits = __builtin_bit_cast(unsigned, b);
}

// The same split of a pair that is ALREADY scaled (s = x * mult, an exact product): r = s * 1 - a, the same exact difference - for callers whose
// scaling is a packed multiply along another axis than the channel pair (fpn_fused_sf.hip: two x per instruction, no register shuffles).
__device__ __forceinline__ void split_scaled_pair_f16(float s0, float s1, unsigned &a_bits, unsigned &b_bits) {
  const split_f16x2 a = {(_Float16)s0, (_Float16)s1};             // round to nearest even
  const unsigned ab = __builtin_bit_cast(unsigned, a);
  float r0, r1;
#ifdef CASMVS_SPLIT_NOASM
  r0 = s0 - (float)a[0];
  r1 = s1 - (float)a[1];
#else
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(s0), "v"(ab));                 // s0 - a.lo
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(s1), "v"(ab));   // s1 - a.hi
#endif
  const split_f16x2 b = {(_Float16)r0, (_Float16)r1};
  a_bits = ab;
  b_bits = __builtin_bit_cast(unsigned, b);
}

// 8 channels of one voxel -> the two 16-byte float16 vectors (slice a, slice b)
__device__ __forceinline__ void split8_f16(const float (&x)[8], float mult, split_u32x4 (&o)[2]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned a, b;
    split_pair_f16(x[2 * p], x[2 * p + 1], mult, a, b);
    o[0][p] = a;
    o[1][p] = b;
  }
}

// 8 scaled channels of one voxel (s = x * mult) -> the two slices
__device__ __forceinline__ void split8_scaled_f16(const float (&sv)[8], split_u32x4 (&o)[2]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned a, b;
    split_scaled_pair_f16(sv[2 * p], sv[2 * p + 1], a, b);
    o[0][p] = a;
    o[1][p] = b;
  }
}

// max(m, |a|, |b|) as ONE v_max3_f32 with source modifiers (the nested form fmaxf(m, fmaxf(|a|, |b|)) compiles to two instructions: the compiler does not
// re-associate maxima).  fmaxf skips NaNs in either form: the same value.
__device__ __forceinline__ float absmax3(float m, float a, float b) { return fmaxf(fmaxf(m, fabsf(a)), fabsf(b)); }

// maximum over the wave of a non-negative float's bit pattern (DPP inside rows of 16, scalar across the four rows)
__device__ __forceinline__ unsigned wave_max_bits(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror
  const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}

// The staged tile's power-of-two scaling from the four waves' maxima (bit patterns of non-negative floats, 16-byte aligned in LDS):
// mult = 2^kx puts the largest magnitude into [2^14, 2^15), inv = 2^-kx; an all-zero or denormal tile is scaled by 2^126.
// Non-finite inputs: the per-thread maxima are fmaxf chains, which skip NaNs - a NaN voxel splits into NaN slices and reaches exactly the outputs whose
// taps touch it, as in the float32 kernels.  An INFINITE voxel makes the tile's maximum infinite (e = 255): no finite scaling exists for its neighbours,
// so the whole staged unit is poisoned (mult = NaN -> every slice NaN -> every output fed by this unit NaN): a superset of the outputs the float32 kernel
// makes non-finite, never a finite wrong value (with mult = 2^-113 the finite voxels of the tile would silently flush to zero).
__device__ __forceinline__ void tile_scale(const unsigned *wave_maxima, float &mult, float &inv) {
  const split_u32x4 w4 = *reinterpret_cast<const split_u32x4 *>(wave_maxima);
  int e = (int)(max(max(w4[0], w4[1]), max(w4[2], w4[3])) >> 23);
  const bool infinite = e >= 255;
  e = e < 15 ? 15 : (e > 254 ? 254 : e);
  mult = __builtin_bit_cast(float, infinite ? 0x7fc00000u : (unsigned)(268 - e) << 23);
  inv = __builtin_bit_cast(float, (unsigned)(e - 14) << 23);
}

}  // namespace casmvs
