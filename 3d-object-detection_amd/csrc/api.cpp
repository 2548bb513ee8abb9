// Error reporting and version query of libfpn_hip.so.
#include <atomic>
#include <map>
#include <mutex>
#include <utility>

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

namespace {
constexpr int kDevices = 64;
std::atomic<int> g_cus[kDevices];                       // 0 = not asked yet
std::mutex g_lds_lock;
std::map<std::pair<const void *, int>, size_t> g_lds;   // (kernel, device) -> bytes already allowed
}  // namespace

int current_device_cus()
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
    if (dev < kDevices) {
        const int known = g_cus[dev].load(std::memory_order_relaxed);
        if (known > 0) return known;
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) return 0;
    if (dev < kDevices) g_cus[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

bool allow_dynamic_lds(const void *fn, size_t bytes)
{
    if (bytes <= 64 * 1024) return true;        // within every kernel's default limit
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> hold(g_lds_lock);
    size_t &have = g_lds[std::make_pair(fn, dev)];
    if (have >= bytes) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();                 // the refusal is reported by the caller, not left pending
        return false;
    }
    have = bytes;
    return true;
}

}  // namespace fpn

extern "C" {

int fpn_abi_version(void) { return 18; }

const char *fpn_last_error_string(void) { return fpn::error_buffer(); }

}
