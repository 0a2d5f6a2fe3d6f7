// api.hip -- version + thread-local error string of the C ABI (include/tgs.h).
#include <stdarg.h>
#include "tgs_common.h"

static thread_local char g_err[512] = "";

void tgs_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int tgs_version(void) { return TGS_VERSION; }
extern "C" const char* tgs_last_error(void) { return g_err; }
