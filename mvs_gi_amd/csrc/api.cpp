// ABI bookkeeping of libmvsgi_hip (include/mvsgi.h): version, last error, the range report's words.
#include "common.hpp"

#include <atomic>
#include <mutex>

namespace mvsgi {
std::string& last_error_ref() {
    static thread_local std::string err;
    return err;
}

// The range report (csrc/split_fmt.hpp): kSatWords words of pinned, device-visible host memory, one per kind of clamp; a wave
// whose clamp engaged stores 1 into its kind's word.  One set per process (every device of the process writes the same words):
// the host reads them without a copy or a stream operation.  Allocated at the first call -- _lib.load() makes it, so that no
// allocation falls into a stream capture.
unsigned* sat_words() {
    static std::atomic<unsigned*> words{nullptr};
    unsigned* p = words.load(std::memory_order_acquire);
    if (p) return p;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    p = words.load(std::memory_order_acquire);
    if (p) return p;
    // (a failure is not cached: a process without a usable device yet -- the CPU-only symbol test -- may ask again later)
    if (hipHostMalloc(reinterpret_cast<void**>(&p), kSatWords * sizeof(unsigned), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    for (int i = 0; i < kSatWords; ++i) p[i] = 0u;
    words.store(p, std::memory_order_release);
    return p;
}
}  // namespace mvsgi

extern "C" int mvsgi_abi_version(void) { return MVSGI_ABI_VERSION; }
extern "C" const char* mvsgi_last_error(void) { return mvsgi::last_error_ref().c_str(); }

extern "C" int mvsgi_saturation_flags(int clear, unsigned* flags) {
    volatile unsigned* w = mvsgi::sat_words();
    MVSGI_REQUIRE(w != nullptr, "mvsgi_saturation_flags: cannot allocate the pinned report words");
    unsigned f = 0;
    for (int i = 0; i < mvsgi::kSatWords; ++i)
        if (w[i]) f |= 1u << i;
    if (clear)
        for (int i = 0; i < mvsgi::kSatWords; ++i) w[i] = 0u;
    if (flags) *flags = f;
    return 0;
}

// diagnostics: the raw words behind mvsgi_saturation_flags (a kernel only ever stores 1: any other value is a stray write)
extern "C" int mvsgi_saturation_words(unsigned* words8) {
    volatile unsigned* w = mvsgi::sat_words();
    MVSGI_REQUIRE(w != nullptr && words8 != nullptr, "mvsgi_saturation_words: no report words / null pointer");
    for (int i = 0; i < mvsgi::kSatWords; ++i) words8[i] = w[i];
    return 0;
}
