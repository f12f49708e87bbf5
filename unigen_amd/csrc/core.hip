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

#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>

int ug_env_int(const char* name, int dflt) {
    static std::mutex mu;
    static std::map<std::string, int> cache;
    static int dynamic = -1;
    std::lock_guard<std::mutex> lk(mu);
    if (dynamic < 0) { const char* d = getenv("UG_ENV_DYNAMIC"); dynamic = (d && atoi(d) != 0) ? 1 : 0; }
    if (!dynamic) {
        auto it = cache.find(name);
        if (it != cache.end()) return it->second;
    }
    const char* e = getenv(name);
    const int v = (e && *e) ? atoi(e) : dflt;
    cache[name] = v;
    return v;
}

extern "C" const char* ug_last_error(void) { return g_err; }
extern "C" int ug_version(void) { return 200; /* 0.2.0: fp32 verification twins, pack/unpack, one-launch combine, strided expert modulation */ }
