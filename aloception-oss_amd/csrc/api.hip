// Library-wide entry points of the C ABI: version and the thread-local error string.
#include "common.hpp"

namespace alo {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace alo

extern "C" int alo_abi_version(void) { return ALO_HOTPATH_ABI_VERSION; }
extern "C" const char* alo_last_error(void) { return alo::error_buffer(); }
