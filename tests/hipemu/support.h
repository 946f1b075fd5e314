// Shared by the translation units that run kernels under tests/hipemu/hip/hip_runtime.h: what abi.hip provides to the launch wrappers, the
// workgroup's LDS array, a small generator.  Include ONCE per translation unit, before the .hip files.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

// what abi.hip provides to the launch wrappers
namespace casmvs {
static thread_local char g_err[512];
char *error_buffer() { return g_err; }
int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
void clear_error() { g_err[0] = 0; }
int ensure_dynamic_lds(const void *, size_t bytes, const char *what) { return bytes <= 160 * 1024 ? 0 : fail(CASMVS_ERR_HIP, "%s: %zu bytes of LDS", what, bytes); }
int resident_blocks(const void *, int, size_t) { return 3; }   // three persistent workgroups: every one walks several items
}  // namespace casmvs
extern "C" const char *casmvs_last_error(void) { return casmvs::g_err; }

namespace {
alignas(64) unsigned char smem_raw[HIPEMU_LDS_BYTES];   // the kernels' `extern __shared__ smem_raw[]`
}


static uint32_t g_rng = 2463534242u;
static float rnd() {
  g_rng ^= g_rng << 13; g_rng ^= g_rng >> 17; g_rng ^= g_rng << 5;
  return (float)(int32_t)g_rng * (1.0f / 2147483648.0f);
}
static double lrelu(double v) { return v > 0 ? v : v * 0.01f; }

