// gs_capi.hip -- library identity and the thread-local error channel of the C ABI
// (include/gs_raster.h).  The stage entry points live next to their kernels.
#include <stdarg.h>

#include "gs_common.h"

namespace gs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace gs

extern "C" int gs_version(void) { return 100; }
extern "C" const char* gs_last_error(void) { return gs::g_err; }
extern "C" const char* gs_arch(void) { return "gfx950"; }
