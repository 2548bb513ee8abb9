// Error reporting and version query of libfpn_hip.so.
#include "fpn_common.h"

namespace fpn {

char *error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace fpn

extern "C" {

int fpn_abi_version(void) { return 18; }

const char *fpn_last_error_string(void) { return fpn::error_buffer(); }

}
