// Error channel of the C-ABI: every entry point returns 0 on success; on failure the message is
// kept per host thread and read back with mmvid_last_error().
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void mmvid_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mmvid_last_error() { return g_err; }

extern "C" int mmvid_abi_version() { return 1; }

// Number of visible HIP devices (0 when none) -- lets the host side fail loudly and early.
extern "C" int mmvid_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
