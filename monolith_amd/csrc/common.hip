// common.hip — error plumbing + workspace of libmonolith_hip.so.
#include "mono_common.h"

namespace mono {

std::string& last_error_ref() {
  static thread_local std::string err;
  return err;
}

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return 1;
}

int Workspace::ensure(size_t need) {
  if (need <= bytes) return 0;
  // grow geometrically so steady-state calls never allocate
  size_t want = bytes ? bytes : (size_t)1 << 20;
  while (want < need) want *= 2;
  if (ptr) {
    MONO_HIP(hipDeviceSynchronize());
    MONO_HIP(hipFree(ptr));
    ptr = nullptr;
    bytes = 0;
  }
  MONO_HIP(hipMalloc(&ptr, want));
  bytes = want;
  return 0;
}

void Workspace::release() {
  if (ptr) (void)hipFree(ptr);
  ptr = nullptr;
  bytes = 0;
}

}  // namespace mono

extern "C" {
const char* mono_last_error(void) { return mono::last_error_ref().c_str(); }
int mono_abi_version(void) { return 1; }
}
