// Error text plumbing for the host-only sanitizer build (libtemfpy_host_asan.so): the full library takes
// these two functions from runtime.hip.
#include <stdarg.h>
#include <stdio.h>

namespace tmf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace tmf

extern "C" const char* tmf_last_error(void) { return tmf::g_err; }
extern "C" int tmf_version(void) { return 100; }
