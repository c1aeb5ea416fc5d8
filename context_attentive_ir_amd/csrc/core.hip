// Library-level entry points: version, thread-local error text.
#include "common.hpp"
#include <string.h>

namespace nir {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace nir

extern "C" int nir_version(void) { return 100; }
extern "C" const char* nir_last_error_string(void) { return nir::g_err; }
