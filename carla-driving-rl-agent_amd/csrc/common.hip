// Error reporting for libcdrl_hip.so
#include "cdrl_common.h"
#include <stdarg.h>
#include <string.h>

namespace cdrl {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* last_error() { return g_err; }

}  // namespace cdrl
