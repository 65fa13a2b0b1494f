// Shared host-side helpers of libmvsgi_hip: error reporting and argument checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

// the range report's words (csrc/api.cpp; kinds in csrc/split_fmt.hpp): every launcher of a kernel that can clamp passes them on
constexpr int kSatWords = 8;
unsigned* sat_words();
#define MVSGI_SAT_WORDS(VAR)                                                            \
    unsigned* VAR = ::mvsgi::sat_words();                                               \
    MVSGI_REQUIRE(VAR != nullptr, "cannot allocate the range report's pinned host words")

inline hipStream_t as_stream(mvsgi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

// Experiment knobs (A/B switches whose measurement is recorded in DESIGN.md / DESIGN_HISTORY.md) exist only in diagnostic
// builds (-DMVSGI_EXPERIMENTAL, __graft_entry__.build_variant): the product library reads no environment variable.
#ifdef MVSGI_EXPERIMENTAL
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif

// Per-(kernel, device) launch set-up of the persistent kernels: the dynamic-LDS limit is a per-device function
// attribute and the persistent grid is sized from THAT device's CU count and residency (a process may drive several
// devices, or a partitioned / smaller one).  Each launcher keeps one `static PersistentGeom[kMaxDevices]`.
constexpr int kMaxDevices = 32;
struct PersistentGeom {
    int wgs_per_cu;   // 0 = not set up on this device yet
    int cus;
};
template <class K>
inline int persistent_geometry(K kern, int threads, size_t lds_bytes, int max_wgs_per_cu,
                               PersistentGeom (&cache)[kMaxDevices], const char* what, PersistentGeom& out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail("%s: hipGetDevice: %s", what, hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail("%s: device ordinal %d not supported (max %d)", what, dev, kMaxDevices);
    PersistentGeom g = cache[dev];           // benign race: the set-up is idempotent
    if (!g.wgs_per_cu) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
        if (e != hipSuccess) return fail("%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds_bytes);
        if (e != hipSuccess || occ < 1) occ = 1;
        int cus = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || cus < 1) return fail("%s: cannot read the device's CU count", what);
        g.wgs_per_cu = occ > max_wgs_per_cu ? max_wgs_per_cu : occ;
        g.cus = cus;
        cache[dev] = g;
    }
    out = g;
    return 0;
}

// CU count of the current device (cached per device ordinal; 256 when it cannot be read: MI355X)
inline int device_cus() {
    static int cache[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    if (!cache[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cache[dev] = cus;
    }
    return cache[dev];
}

// csrc/conv3d_rs.hip: the 32 -> 32 register-stationary kernel as the body of the polyphase ResizeConv3d (csrc/conv3d_up2poly.hip)
constexpr size_t kRs32PackedBytes = (size_t)2 * 2 * 14 * 2 * 64 * 16;      // one weight set in the kernel's lane order
void rs32_pack_weights_host(const float* w_oidhw_32x32x27, void* packed, bool f16);
int rs32_up2_launch(const void* x_split, const void* w_sets, const float* scale32, const float* shift32, void* y, int y_is_split,
                    int B, int D, int H, int W, float neg_slope, bool f16, hipStream_t st);

// csrc/conv3d_wino_up2.hip: the polyphase ResizeConv3d's main kernel in Winograd form (fp16 split, split-padded output)
constexpr size_t kWinoUp2RoleBytes = (size_t)4 * 4 * 3 * 2 * 2 * 64 * 16;      // one row phase's weights in the kernel's lane order
void wino_up2_pack_role(const float* w32_oidhw, unsigned short* wp, float* unscale32);
bool wino_up2_applies(int D, int H, int W);
int wino_up2_launch(const void* x_split, const void* w_roles, const float* unw, const float* scale16, const float* shift16, void* y_split,
                    int B, int D, int H, int W, float neg_slope, hipStream_t st);

}  // namespace mvsgi
