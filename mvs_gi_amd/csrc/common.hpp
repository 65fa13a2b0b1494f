// Shared host-side helpers of libmvsgi_hip: error reporting and argument checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "../../include/mvsgi.h"

namespace mvsgi {

std::string& last_error_ref();

inline int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return 1;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

#define MVSGI_REQUIRE(cond, ...)                      \
    do {                                              \
        if (!(cond)) return ::mvsgi::fail(__VA_ARGS__); \
    } while (0)

inline hipStream_t as_stream(mvsgi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace mvsgi
