// lib.cpp — error plumbing and identification of libdt_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
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

// hipGraphUpload through THIS library's HIP runtime (the one torch loaded): a host binding that dlopens "libamdhip64.so" by
// name can get a second copy of the runtime (ROCm's next to torch's bundled one) and hand it the other copy's handles
extern "C" int dt_graph_upload(void* graph_exec, void* stream) {
    if (!graph_exec) {
        dt::set_error("dt_graph_upload: null graph");
        return DT_ERR_INVALID_ARG;
    }
    const hipError_t e = hipGraphUpload(reinterpret_cast<hipGraphExec_t>(graph_exec), reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        dt::set_error("dt_graph_upload: %s", hipGetErrorString(e));
        return DT_ERR_LAUNCH;
    }
    return DT_OK;
}
