// Library-level entry points of libg4d_hip: version + thread-local error text.
#include <stdarg.h>

#include "g4d_common.h"

namespace g4d {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace g4d

extern "C" int g4d_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char *g4d_last_error(void) { return g4d::g_err; }
