// libtspgnn: status / error plumbing of the C ABI (include/tspgnn.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace tspgnn {

static thread_local char g_err[512] = {0};

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return TSPGNN_OK;
    return fail(static_cast<int>(e), "%s: HIP launch failed: %s", what, hipGetErrorString(e));
}

}  // namespace tspgnn

extern "C" int tspgnn_version(void) { return TSPGNN_ABI_VERSION; }

extern "C" const char* tspgnn_last_error(void) { return tspgnn::g_err; }
