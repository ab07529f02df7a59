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

extern "C" int recmv_abi_version(void) { return 7; }
extern "C" const char* recmv_last_error(void) { return recmv::g_err; }
