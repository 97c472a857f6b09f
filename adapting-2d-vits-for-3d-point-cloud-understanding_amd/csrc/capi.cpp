// Error channel and version of the C ABI (include/p4p_hip.h).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void p4p_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* p4p_last_error(void) { return g_err; }
extern "C" int p4p_abi_version(void) { return 1; }
