// Shared host-side helpers for libdvc_hip.so (gfx950 only; no CUDA/HIP dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "dvc_hip.h"

// Diagnostics (timeline stamps, timing-experiment variants that compute WRONG results) exist only in a -DDVC_DEBUG build
// (`make DEBUG=1` -> dvc_amd/libdvc_hip_debug.so, used by tools/); the production library carries neither the hooks nor
// the process-global state behind them.
#ifdef DVC_DEBUG
constexpr bool kDvcDebug = true;
#else
constexpr bool kDvcDebug = false;
#endif

char* dvc_err_buf();  // thread-local, 512 bytes

static inline int dvc_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dvc_err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

#define DVC_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) return dvc_fail("%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define DVC_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) return dvc_fail(__VA_ARGS__);   \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long cdivl(long a, long b) { return (a + b - 1) / b; }
