#include "host_common.h"
#include <cstdarg>
#include <cstdio>
#include "dali_amd_host.h"

namespace daliamd_host {
static thread_local char g_err[1024] = "";
int Fail(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}
}  // namespace daliamd_host

extern "C" const char *daliamdHostGetLastErrorMessage(void) { return daliamd_host::g_err; }
