// lib.hip -- library-level entry points: ABI version and thread-local last-error string.
#include "common.h"

static thread_local char g_err[512] = "";

void mv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mv_abi_version(void) { return MV_ABI_VERSION; }
extern "C" const char* mv_last_error(void) { return g_err; }
