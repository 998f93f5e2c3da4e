// api.hip -- library identification and the last-error slot of the C ABI.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ag_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int ag_abi_version(void) { return 1; }
extern "C" const char* ag_arch(void) { return "gfx950"; }
extern "C" const char* ag_last_error(void) { return g_err; }
