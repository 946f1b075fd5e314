// Shared host-side helpers of libcasmvs_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "casmvs.h"

// A kernel's dynamic LDS array.  (tests/hipemu compiles the kernels for the host, where a translation unit can hold function-scope static LDS arrays
// - `__shared__` = `static` there - or `extern __shared__` declarations, not both: kernels that use both declare the dynamic one through this.)
#ifdef HIPEMU_LDS_BYTES
#define CASMVS_DYNAMIC_LDS(T, name) T *name = reinterpret_cast<T *>(hipemu::g_lds)
#else
#define CASMVS_DYNAMIC_LDS(T, name) extern __shared__ T name[]
#endif

// LDS request floor of the split-f16 kernels whose own need is below 56 KiB: at most two workgroups (two waves per SIMD) per CU.  Rounds 3-4 chose it
// because float32 staging arithmetic beside two waves' f16 matrix instructions came out wrong (lanes 48-63) - the packed-float32 op_sel fault that
// casmvsnet_pl_amd/build.py now assembles away; kept as the measured-best occupancy (profiles/r05_sf_lds_floor_ab.txt), 0 = none.
#ifndef CASMVS_SF_LDS_FLOOR
#define CASMVS_SF_LDS_FLOOR (56 * 1024)
#endif

namespace casmvs {

char *error_buffer();  // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);
void clear_error();

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(CASMVS_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return CASMVS_OK;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Per-(device, kernel) launch constants, cached under a mutex (abi.hip).  Both are properties of the DEVICE the
// calling thread is on: a process that drives several GPUs gets each one opted in / measured separately.
// ensure_dynamic_lds: kernels that use more than the default 64 KiB of dynamic LDS must opt in once per device.
int ensure_dynamic_lds(const void *kernel, size_t bytes, const char *what);
// Number of workgroups of `kernel` resident on the whole device at once (persistent kernels launch exactly that many).
int resident_blocks(const void *kernel, int threads, size_t lds_bytes);

}  // namespace casmvs

#define CASMVS_REQUIRE(cond, ...)                                            \
  do {                                                                       \
    if (!(cond)) return casmvs::fail(CASMVS_ERR_INVALID_ARG, __VA_ARGS__);   \
  } while (0)
