// ABI plumbing: version, thread-local error text, launch check.
#include "common.h"

namespace mma {
char* err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(100 + (int)e, "%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}
}  // namespace mma

extern "C" int mma_abi_version(void) { return MMA_ABI_VERSION; }
extern "C" const char* mma_last_error(void) { return mma::err_buf(); }
