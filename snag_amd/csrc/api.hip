// Error reporting and version for the C ABI (include/snag_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.hip.h"

namespace snag {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SNAG_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SNAG_OK;
}

}  // namespace snag

extern "C" SNAG_API int snag_version(void) { return SNAG_ABI_VERSION; }
extern "C" SNAG_API const char* snag_last_error(void) { return snag::g_err; }
