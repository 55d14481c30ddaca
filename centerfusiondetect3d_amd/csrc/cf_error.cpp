// Thread-local last-error string of the C ABI.
#include <stdarg.h>
#include <stdio.h>
#include "cf_hip.h"

static thread_local char g_err[512] = "";

void cf_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* cf_last_error(void) { return g_err; }
extern "C" int cf_abi_version(void) { return CF_ABI_VERSION; }
