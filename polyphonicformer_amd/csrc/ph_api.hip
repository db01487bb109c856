// libpolyhead: version / error plumbing of the C ABI (include/polyhead.h).
#include <stdarg.h>

#include "ph_common.h"

static thread_local char g_err[512] = "";

void ph_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ph_version(void) { return PH_VERSION; }
extern "C" const char* ph_last_error_string(void) { return g_err; }
