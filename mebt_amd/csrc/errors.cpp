// Last-error bookkeeping behind the C ABI (thread-local message, like dlerror()).
#include "common.h"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void mebt_set_error(const char* msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
void mebt_set_hip_error(hipError_t e, const char* what) {
    snprintf(g_err, sizeof(g_err), "HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
}
extern "C" const char* mebt_last_error(void) { return g_err; }
