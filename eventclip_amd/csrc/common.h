// Shared host-side helpers for libeventclip_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/eventclip_hip.h"

namespace ec {

// thread-local message behind ec_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

#define EC_CHECK_HIP(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return ::ec::fail(EC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                              hipGetErrorString(_e), __FILE__, __LINE__);                   \
    } while (0)

#define EC_REQUIRE(cond, ...)                                                               \
    do {                                                                                    \
        if (!(cond)) return ::ec::fail(EC_ERR_INVALID, __VA_ARGS__);                        \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

}  // namespace ec
