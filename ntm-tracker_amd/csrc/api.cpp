// Error reporting and version for libntmtrack_hip.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ntmtrack.h"

static thread_local char g_err[512] = "";

void ntk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ntk_version(void) { return 100; }
extern "C" const char* ntk_last_error(void) { return g_err; }
