// Library-level helpers of libdm_amd.so: error reporting and version.
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void dm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dm_last_error(void) { return g_err; }
extern "C" int dm_version(void) { return 100; }
