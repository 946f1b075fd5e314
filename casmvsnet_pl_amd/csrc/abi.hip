// ABI housekeeping for libcasmvs_hip.so: version + thread-local error message.
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace casmvs {

char *error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

void clear_error() { error_buffer()[0] = '\0'; }

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}


namespace {
struct KernelCache {
  std::mutex m;
  std::map<std::pair<int, const void *>, size_t> lds_set;                          // (device, kernel) -> bytes opted in
  std::map<std::pair<int, std::pair<const void *, size_t>>, int> resident;         // (device, (kernel, lds)) -> blocks
};
KernelCache &kernel_cache() {
  static KernelCache c;
  return c;
}
}  // namespace

int ensure_dynamic_lds(const void *kernel, size_t bytes, const char *what) {
  if (bytes <= 64 * 1024) return CASMVS_OK;
  int dev = 0;
  (void)hipGetDevice(&dev);
  KernelCache &c = kernel_cache();
  std::lock_guard<std::mutex> lock(c.m);
  auto key = std::make_pair(dev, kernel);
  auto it = c.lds_set.find(key);
  if (it != c.lds_set.end() && it->second >= bytes) return CASMVS_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail(CASMVS_ERR_HIP, "%s: hipFuncSetAttribute(%zu B LDS): %s", what, bytes, hipGetErrorString(e));
  c.lds_set[key] = bytes;
  return CASMVS_OK;
}

int resident_blocks(const void *kernel, int threads, size_t lds_bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  KernelCache &c = kernel_cache();
  std::lock_guard<std::mutex> lock(c.m);
  auto key = std::make_pair(dev, std::make_pair(kernel, lds_bytes));
  auto it = c.resident.find(key);
  if (it != c.resident.end()) return it->second;
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  return c.resident[key] = per_cu * cus;
}

}  // namespace casmvs

extern "C" int casmvs_abi_version(void) { return CASMVS_ABI_VERSION; }

#ifndef CASMVS_PACKED_OPSEL_SAFE
#define CASMVS_PACKED_OPSEL_SAFE 0   // casmvsnet_pl_amd/build.py defines 1: it assembles the device code with the unsafe packed-float32 forms rewritten
#endif
extern "C" int casmvs_packed_opsel_safe(void) { return CASMVS_PACKED_OPSEL_SAFE; }

extern "C" const char *casmvs_last_error(void) { return casmvs::error_buffer(); }
