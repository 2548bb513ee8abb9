// Grouping for set-abstraction modules as one pass: for every (query j, slot k)
// the output row is the concatenation of the neighbour's coordinates relative to
// the query and the neighbour's features,
//     out[b,j,k,:] = [ xyz[b,idx[b,j,k],:] - new_xyz[b,j,:] | points[b,idx[b,j,k],:] ]
// (or features first).  It replaces group_point(xyz) -> subtract -> group_point(points)
// -> concat, four passes over the largest tensor of the module.  Same thread
// mapping as group_point: LW lanes walk the channels of a row, rows are contiguous
// in memory, so stores are contiguous runs.
#include <algorithm>
#include <cstdint>

#include "fpn_common.h"

namespace fpn {
namespace {

constexpr int kThreads = 256;

template <int LW>
__global__ __launch_bounds__(kThreads) void group_concat_kernel(
    int b, int n, int m, int nsample, int c, int xyz_first, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out)
{
    constexpr int RPB = kThreads / LW;
    const int lx = threadIdx.x % LW, ly = threadIdx.x / LW;
    const int width = 3 + c;
    const int rows_per_batch = m * nsample;
    for (int bi = blockIdx.y; bi < b; bi += gridDim.y) {
        const float *px = xyz + (size_t)bi * n * 3;
        const float *pp = points ? points + (size_t)bi * n * c : nullptr;
        for (int r = blockIdx.x * RPB + ly; r < rows_per_batch; r += gridDim.x * RPB) {
            const size_t row = (size_t)bi * rows_per_batch + r;
            const int ii = idx[row];
            const float *q = new_xyz + ((size_t)bi * m + r / nsample) * 3;
            float *dst = out + row * width;
            for (int l = lx; l < width; l += LW) {
                const int lxyz = xyz_first ? l : l - c;   // position inside the xyz block
                float v;
                if (lxyz >= 0 && lxyz < 3) v = px[(size_t)ii * 3 + lxyz] - q[lxyz];
                else v = pp[(size_t)ii * c + (xyz_first ? l - 3 : l)];
                dst[l] = v;
            }
        }
    }
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" int fpn_group_concat(int b, int n, int m, int nsample, int c, int xyz_first, const float *xyz,
                                const float *new_xyz, const float *points, const int *idx, float *out,
                                void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0 && c >= 0, "group_concat: negative size");
    const long rows = (long)m * nsample;
    if (b == 0 || rows == 0) return FPN_OK;
    FPN_REQUIRE(xyz && new_xyz && idx && out && (points || c == 0), "group_concat: null pointer");
    FPN_REQUIRE(rows <= INT32_MAX, "group_concat: m*nsample too large");
    int lw = 1;
    while (lw < 3 + c && lw < kWave) lw <<= 1;
    const int rpb = kThreads / lw;
    dim3 grid((unsigned)std::min<long>((rows + rpb - 1) / rpb, 1 << 20), (unsigned)std::min(b, 65535));
    hipStream_t st = as_stream(stream);
#define FPN_GC(LW) hipLaunchKernelGGL((group_concat_kernel<LW>), grid, dim3(kThreads), 0, st, b, n, m, nsample, c, \
                                      xyz_first, xyz, new_xyz, points, idx, out)
    switch (lw) {
    case 4: FPN_GC(4); break;
    case 8: FPN_GC(8); break;
    case 16: FPN_GC(16); break;
    case 32: FPN_GC(32); break;
    default: FPN_GC(64); break;
    }
#undef FPN_GC
    return check_launch("fpn_group_concat");
}
