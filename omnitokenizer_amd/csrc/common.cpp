// Error reporting shared by all C-ABI entry points.
#include "common.h"
#include <string.h>

namespace omnitok {
static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace omnitok

extern "C" const char *omnitok_last_error(void) { return omnitok::g_err; }
extern "C" const char *omnitok_version(void) { return "omnitok 0.1 gfx950 fp32-mfma"; }
