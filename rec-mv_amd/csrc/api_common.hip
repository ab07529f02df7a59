// Error plumbing + ABI version of librecmv_hip.so.
#include "common.h"
#include <stdarg.h>

namespace recmv {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace recmv

extern "C" int recmv_abi_version(void) { return 9; }
extern "C" const char* recmv_last_error(void) { return recmv::g_err; }

// 1 when EVERY kernel of this library was built without packed-f32 VALU instructions (rec-mv_amd/build.py under
// RECMV_NO_PACKED_F32=1) — the build the bf16x6 matrix mode needs to be reproducible, see build.py — else 0.  (ABI v7)
extern "C" int recmv_no_packed_f32(void) {
#ifdef RECMV_NO_PACKED_F32
  return 1;
#else
  return 0;
#endif
}
