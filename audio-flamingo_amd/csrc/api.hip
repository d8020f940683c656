// libafk.so: error reporting + version (the rest of the C ABI lives next to its kernels).
#include "common.h"
#include "../../include/afk.h"

static thread_local char g_err[512] = "";

int afk_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* afk_last_error(void) { return g_err; }
extern "C" int afk_version(void) { return 1; }
