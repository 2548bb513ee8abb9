// Monte-Carlo ensemble statistics of the Bayesian detector: one thread per frustum
// folds the S sampled box vectors into the mean detection, the total variance
// (trace of C = 1/S sum v v^T - mean mean^T) and the Shannon entropies of the
// sample-mean class probabilities.  Moments are accumulated in double: centres are
// tens of metres and the spread over samples can be millimetres, so E[v^2]-mean^2
// cancels catastrophically in fp32.
#include <cmath>

#include "fpn_common.h"

namespace fpn {
namespace {

constexpr int NH = 12, NS = 8, BOX = 59;

// mean sizes (l,w,h) of the eight size clusters, class order Car, Van, Truck,
// Pedestrian, Person_sitting, Cyclist, Tram, Misc
__constant__ double kMeanSize[NS][3] = {
    {3.88311640418, 1.62856739989, 1.52563191462}, {5.06763659, 1.9007158, 2.20532825},
    {10.13586957, 2.58549199, 3.2520595},         {0.84422524, 0.66068622, 1.76255119},
    {0.80057803, 0.5983815, 1.27450867},          {1.76282397, 0.59706367, 1.73698127},
    {16.17150617, 2.53246914, 3.53079012},        {3.64300781, 1.54298177, 1.92320313}};

// One wave per frustum: lane t takes samples t, t + 64, ... (a sample's twenty exponentials are in double
// precision: one thread per frustum walking 32 samples took 0.1 ms for 128 frustums, two waves on the whole
// chip), the lanes' partial sums meet in a butterfly of fixed order - the result does not depend on timing.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(64) void mc_statistics_kernel(int s, int b, const float *__restrict__ box,
                                                           const float *__restrict__ c1, float *__restrict__ stats)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    double sum[7] = {0, 0, 0, 0, 0, 0, 0}, sq[7] = {0, 0, 0, 0, 0, 0, 0};
    double ph[NH], ps[NS];
    for (int i = 0; i < NH; ++i) ph[i] = 0;
    for (int i = 0; i < NS; ++i) ps[i] = 0;
    for (int t = lane; t < s; t += 64) {
        const float *o = box + ((size_t)t * b + f) * BOX;
        const float *c = c1 + ((size_t)t * b + f) * 3;
        double v[7];
        for (int d = 0; d < 3; ++d) v[d] = (double)o[d] + (double)c[d];
        int hc = 0, sc = 0;
        float hm = o[3], sm = o[3 + 2 * NH];
        for (int i = 1; i < NH; ++i)
            if (o[3 + i] > hm) { hm = o[3 + i]; hc = i; }
        for (int i = 1; i < NS; ++i)
            if (o[3 + 2 * NH + i] > sm) { sm = o[3 + 2 * NH + i]; sc = i; }
        for (int d = 0; d < 3; ++d)
            v[3 + d] = kMeanSize[sc][d] + (double)o[3 + 2 * NH + NS + sc * 3 + d] * kMeanSize[sc][d];
        v[6] = hc * (2.0 * M_PI / NH) + (double)o[3 + NH + hc] * (M_PI / NH);
        for (int d = 0; d < 7; ++d) { sum[d] += v[d]; sq[d] += v[d] * v[d]; }
        double zh = 0, zs = 0, eh[NH], es[NS];
        for (int i = 0; i < NH; ++i) { eh[i] = exp((double)o[3 + i] - (double)hm); zh += eh[i]; }
        for (int i = 0; i < NS; ++i) { es[i] = exp((double)o[3 + 2 * NH + i] - (double)sm); zs += es[i]; }
        for (int i = 0; i < NH; ++i) ph[i] += eh[i] / zh;
        for (int i = 0; i < NS; ++i) ps[i] += es[i] / zs;
    }
    for (int d = 0; d < 7; ++d) { sum[d] = wave_sum(sum[d]); sq[d] = wave_sum(sq[d]); }
    for (int i = 0; i < NH; ++i) ph[i] = wave_sum(ph[i]);
    for (int i = 0; i < NS; ++i) ps[i] = wave_sum(ps[i]);
    if (lane != 0) return;
    float *out = stats + (size_t)f * 30;
    double tv = 0;
    for (int d = 0; d < 7; ++d) {
        const double mu = sum[d] / s;
        out[d] = (float)mu;
        tv += sq[d] / s - mu * mu;
    }
    out[7] = (float)tv;
    double hh = 0, hs = 0;
    for (int i = 0; i < NH; ++i) {
        const double p = ph[i] / s;
        out[10 + i] = (float)p;
        if (p > 0) hh -= p * log(p);
    }
    for (int i = 0; i < NS; ++i) {
        const double p = ps[i] / s;
        out[22 + i] = (float)p;
        if (p > 0) hs -= p * log(p);
    }
    out[8] = (float)hh;
    out[9] = (float)hs;
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" int fpn_mc_statistics(int s, int b, const float *box_output, const float *stage1_center,
                                 float *stats, void *stream)
{
    FPN_REQUIRE(s >= 1 && b >= 0, "mc_statistics: need at least one sample");
    if (b == 0) return FPN_OK;
    FPN_REQUIRE(box_output && stage1_center && stats, "mc_statistics: null pointer");
    hipLaunchKernelGGL(mc_statistics_kernel, dim3((unsigned)b), dim3(64), 0, as_stream(stream), s, b, box_output,
                       stage1_center, stats);
    return check_launch("fpn_mc_statistics");
}
