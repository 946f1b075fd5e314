// Debug tooling (tools/debug/disturber.py): small "disturber" kernels for the co-residency experiments of round 3 - a float32 layer kernel
// produced a few wrong values when workgroups of the split-f16 kernels ran beside it on another stream; these kernels isolate
// which property of the neighbour matters (matrix-instruction type, LDS footprint above 64 KiB, LDS traffic).  Not used by the engine and NOT part of
// the production library: compiled only into -DCASMVS_TRACE builds (tools/build_trace_lib.sh -> libcasmvs_trace.so, CASMVS_LIB_PATH).
#ifdef CASMVS_TRACE
#include <cstdint>

#include "buffer_ops.h"
#include "common.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4d __attribute__((ext_vector_type(4)));

// kind 0: v_mfma_f32_16x16x4_f32, 1: v_mfma_f32_16x16x32_f16, 2: v_mfma_f32_16x16x32_bf16, 3: LDS b128 traffic over the whole dynamic
// allocation, 4: VALU fma chain.  Every kind allocates `lds_bytes` of dynamic LDS (touched only by kind 3).
template <int KIND>
__global__ __launch_bounds__(256) void disturb_kernel(float *sink, int iters, int lds_units) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  u32x4d *lds = reinterpret_cast<u32x4d *>(dsm);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  u32x4d ua = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, ub = ua;
  if (KIND == 3) {
    for (int i = threadIdx.x; i < lds_units; i += 256) lds[i] = u32x4d{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    else if (KIND == 1) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ua), __builtin_bit_cast(f16x8, ub), acc, 0, 0, 0);
    else if (KIND == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc, 0, 0, 0);
    else if (KIND == 3) {
      const int i = (threadIdx.x + it * 263) % lds_units;
      const u32x4d v = lds[i];
      lds[(i + 97) % lds_units] = v;
      acc[0] += (float)v[0];
    } else {
      acc[0] = fmaf(acc[0], a, b);
      acc[1] = fmaf(acc[1], a, b);
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
}  // namespace

extern "C" int casmvs_debug_disturb(int kind, int blocks, int iters, int lds_bytes, float *sink, void *stream) {
  casmvs::clear_error();
  CASMVS_REQUIRE(kind >= 0 && kind <= 4 && blocks > 0 && iters > 0 && lds_bytes >= 16 && lds_bytes <= 160 * 1024 && sink, "debug_disturb: bad arguments");
  hipStream_t st = (hipStream_t)stream;
#define CASMVS_DD(K)                                                                                                        \
  {                                                                                                                         \
    auto kernel = disturb_kernel<K>;                                                                                        \
    if (int rc = casmvs::ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), (size_t)lds_bytes, "disturb_kernel")) return rc; \
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, st, sink, iters, lds_bytes / 16);                 \
  }
  switch (kind) {
    case 0: CASMVS_DD(0) break;
    case 1: CASMVS_DD(1) break;
    case 2: CASMVS_DD(2) break;
    case 3: CASMVS_DD(3) break;
    default: CASMVS_DD(4) break;
  }
#undef CASMVS_DD
  return casmvs::check_launch("disturb_kernel");
}
#endif  // CASMVS_TRACE
