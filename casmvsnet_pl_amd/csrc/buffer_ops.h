// Raw buffer addressing shared by the tile-staging kernels (conv3d_mfma.hip, prob_regress.hip).
//
// address = descriptor base (SGPRs) + per-lane byte offset (VGPR, invariant for a whole tile) + scalar byte offset
// (SGPR).  A load then costs one scalar add and one buffer_load - no per-lane 64-bit address arithmetic, no
// predicate - and a lane whose position is outside the image carries the offset kOOB >= num_records, for which the
// hardware returns 0 (loads) or drops the access (stores): the convolutions' zero padding for free.
#pragma once
#include <hip/hip_runtime.h>

#ifndef CASMVS_CONV_STORE_AUX
#define CASMVS_CONV_STORE_AUX 0   // cache-policy bits of the activation stores (2 = nt; A/B builds)
#endif

namespace casmvs {
namespace buf {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kOOB = (int)0x80000000u;  // needs num_records <= 2^31 bytes (checked on the host)

__device__ __forceinline__ rsrc_t make_rsrc(const float *base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x2 buf_load2(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4v buf_load4(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_store(float v, rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, CASMVS_CONV_STORE_AUX);
}
__device__ __forceinline__ void buf_store2(f32x2 v, rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voff, soff, CASMVS_CONV_STORE_AUX);
}

// Workgroup b runs on XCD b % 8 and each XCD has its own 4 MiB L2: items are dealt XCD-major, XCD x owns the
// contiguous range [start(x), start(x+1)), so that the workgroups resident on it at one time walk neighbouring tiles
// and share their halos inside that L2.  Bijection on [0, total).
__device__ __forceinline__ int xcd_major(int v, int total) {
  const int xcd = v & 7, idx = v >> 3;
  const int q = total >> 3, r = total & 7;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// A/B builds (work-order experiments, tools/notorch/ab_step.py): of a persistent kernel with TWO resident workgroups per CU the
// hardware places dispatch slots idx and idx + 32 of an XCD on the same CU; this remap gives that pair NEIGHBOURING items (and
// with them shared halo lines in the CU's L1 / the XCD's L2).  A bijection inside full groups of 64 slots; v = blockIdx.x + n gridDim.x
// with gridDim.x a multiple of 8.  conv0_splitf16.hip measured +5 % with it at cin = 8 / 32 (profiles/r03_conv0_tile_order_ab.txt).
__device__ __forceinline__ int cu_pair_remap(int v, int total) {
  const int xcd = v & 7, idx = v >> 3;
  if ((idx | 63) < (total >> 3)) v = ((((idx & ~63) | ((idx & 31) << 1) | ((idx >> 5) & 1))) << 3) | xcd;
  return v;
}

}  // namespace buf
}  // namespace casmvs
