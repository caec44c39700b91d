// lib.cpp — error plumbing and identification of libdt_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "../../include/dt_hip.h"

namespace dt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dt

extern "C" int dt_version(void) { return 1; }
extern "C" const char* dt_last_error(void) { return dt::g_err; }
extern "C" const char* dt_build_arch(void) { return "gfx950"; }
#ifndef DT_SOURCE_HASH
#define DT_SOURCE_HASH "unknown"
#endif
extern "C" const char* dt_source_hash(void) { return DT_SOURCE_HASH; }
