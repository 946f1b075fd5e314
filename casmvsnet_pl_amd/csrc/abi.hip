// ABI housekeeping for libcasmvs_hip.so: version + thread-local error message.
#include <cstring>

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

}  // namespace casmvs

extern "C" int casmvs_abi_version(void) { return CASMVS_ABI_VERSION; }

extern "C" const char *casmvs_last_error(void) { return casmvs::error_buffer(); }
