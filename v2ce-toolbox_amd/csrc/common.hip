// common.hip -- error reporting and version of libv2ce_hip.so.
#include "common.h"

#include <cstring>

namespace v2ce {
namespace {
thread_local char g_err[512] = "";
}
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void clear_error() { g_err[0] = '\0'; }
}  // namespace v2ce

extern "C" const char *v2ce_version(void) { return "v2ce-toolbox_amd 0.6 (gfx950)"; }
extern "C" const char *v2ce_last_error(void) { return v2ce::g_err; }
