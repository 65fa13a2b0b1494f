// ABI bookkeeping of libmvsgi_hip (include/mvsgi.h).
#include "common.hpp"

namespace mvsgi {
std::string& last_error_ref() {
    static thread_local std::string err;
    return err;
}
}  // namespace mvsgi

extern "C" int mvsgi_abi_version(void) { return MVSGI_ABI_VERSION; }
extern "C" const char* mvsgi_last_error(void) { return mvsgi::last_error_ref().c_str(); }
