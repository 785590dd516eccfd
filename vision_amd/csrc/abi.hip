// abi.hip — version / error text of libdetgeom.
#include "common.hpp"
#include <string.h>

static thread_local char g_err[512] = "";

void dg_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *dg_version(void) { return "detgeom 0.1 (gfx950)"; }

extern "C" int dg_last_error(char *buf, size_t n)
{
    size_t len = strlen(g_err);
    if (buf && n > 0) {
        size_t c = len < n - 1 ? len : n - 1;
        memcpy(buf, g_err, c);
        buf[c] = 0;
    }
    return (int)len;
}
