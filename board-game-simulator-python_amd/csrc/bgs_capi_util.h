// bgs_capi_util.h -- error reporting shared by the translation units that implement the C ABI (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/bgs.h"

namespace bgs {
// stores the message for bgs_last_error() (thread-local) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace bgs

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return bgs::fail(BGS_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define NEED(cond, ...)                                          \
    do {                                                         \
        if (!(cond)) return bgs::fail(BGS_ERR_ARG, __VA_ARGS__); \
    } while (0)
