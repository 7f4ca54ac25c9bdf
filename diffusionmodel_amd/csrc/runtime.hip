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

float* dm_g_ws = nullptr;
int64_t dm_g_ws_bytes = 0;
extern "C" int dm_set_workspace(void* ws, int64_t bytes) {
    DM_CHECK_ARG((ws == nullptr) == (bytes == 0) && bytes >= 0 && ((uintptr_t)ws & 15) == 0, "dm_set_workspace: need a 16-byte aligned buffer and its size (or NULL, 0)");
    dm_g_ws = (float*)ws;
    dm_g_ws_bytes = bytes;
    return DM_OK;
}
