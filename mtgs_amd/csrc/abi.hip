// abi.hip -- version + thread-local error string of libmtgs_rast.so.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

static thread_local char g_last_error[512] = "";

void mtgs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" int mtgs_rast_version(void) { return MTGS_RAST_ABI_VERSION; }
extern "C" int mtgs_rast_hot_version(void) { return MTGS_RAST_HOT_ABI_VERSION; }
extern "C" const char *mtgs_rast_last_error(void) { return g_last_error; }
