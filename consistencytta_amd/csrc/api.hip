// Error plumbing and version of libctta_hip.so (see include/ctta.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[1024] = "";

void ctta_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ctta_last_error(void) { return g_err; }
extern "C" int ctta_version(void) { return 100; }
