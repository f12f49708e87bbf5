// Error plumbing + version for libunigen_hip.so.
#include "ug_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ug_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* ug_last_error(void) { return g_err; }
extern "C" int ug_version(void) { return 200; /* 0.2.0: fp32 verification twins, pack/unpack, one-launch combine, strided expert modulation */ }
