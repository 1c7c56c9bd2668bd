// Error reporting for libcdrl_hip.so
#include "cdrl_common.h"
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

namespace cdrl {

static thread_local char g_err[1024] = "";
thread_local TailEvents* tl_tail = nullptr;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* last_error() { return g_err; }

// Every tuning / diagnostic switch of the library is read through this function.  CDRL_DIAG_* switches produce WRONG RESULTS or
// races by design (timing diagnostics): they are ignored unless the master switch CDRL_DIAG=1 is set as well, so a stray
// exported variable cannot change what a benchmark measures (cdrl_diag_active / cdrl_env_overrides report what is in effect).
const char* cdrl_getenv(const char* name) {
    if (strncmp(name, "CDRL_DIAG_", 10) == 0) {
        const char* m = getenv("CDRL_DIAG");
        if (!m || atoi(m) == 0) return nullptr;
    }
    return getenv(name);
}

extern "C" char** environ;

int diag_active() {
    const char* m = getenv("CDRL_DIAG");
    if (!m || atoi(m) == 0) return 0;
    int n = 0;
    for (char** e = environ; e && *e; ++e)
        if (strncmp(*e, "CDRL_DIAG_", 10) == 0) {
            const char* eq = strchr(*e, '=');
            if (eq && atoi(eq + 1) != 0) ++n;
        }
    return n;
}

int env_overrides(char* buf, int cap) {
    int n = 0, off = 0;
    if (buf && cap > 0) buf[0] = 0;
    for (char** e = environ; e && *e; ++e)
        if (strncmp(*e, "CDRL_", 5) == 0) {
            ++n;
            if (buf && cap > 0) {
                const int w = snprintf(buf + off, (size_t)(cap - off), "%s%s", off ? " " : "", *e);
                if (w > 0 && off + w < cap) off += w;
            }
        }
    return n;
}

}  // namespace cdrl
