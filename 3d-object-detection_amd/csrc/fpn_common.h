// Shared host-side helpers of libfpn_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "fpn.h"

namespace fpn {

// Thread-local message behind fpn_last_error_string().
char *error_buffer();
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Every launcher ends with this: a launch the runtime rejected (bad grid, LDS
// over the limit, ...) becomes FPN_E_LAUNCH instead of a silent no-op.
inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(FPN_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return FPN_OK;
}

constexpr int kWave = 64;  // gfx950 wavefront width

// Launch-time facts the persistent kernels size their grids by, asked of the runtime ONCE
// per device (a step makes tens of such launches; api.cpp holds the tables):
// the CU count of the calling thread's current device (0 if the runtime cannot tell), and
// "kernel fn may use `bytes` of dynamic LDS per workgroup" - hipFuncSetAttribute is issued
// only the first time a (kernel, device) pair asks for that much; false if it is refused.
int current_device_cus();
bool allow_dynamic_lds(const void *fn, size_t bytes);

}  // namespace fpn

#define FPN_REQUIRE(cond, ...)                                         \
    do {                                                               \
        if (!(cond)) return ::fpn::fail(FPN_E_INVALID_ARG, __VA_ARGS__); \
    } while (0)
