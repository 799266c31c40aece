// Error plumbing of the C-ABI (include/efgh_hip.h).
#include "common.h"

static thread_local char g_err[512] = "";

void efgh_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *efgh_last_error(void) { return g_err; }
extern "C" int efgh_version(void) { return EFGH_ABI_VERSION; }
