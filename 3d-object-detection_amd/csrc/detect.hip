// Detection records: what the reference's detection driver computes on the host
// after `model.predict` (scripts/test.py:145-160) and in its label writer
// (scripts/provider.py:366-373), as one kernel on the network's outputs.
//
// A WAVE owns a frustum.  Phase 1 streams the frustum's (n,2) segmentation logits once
// (8 bytes per lane and step, a step = 64 consecutive points = one 512-byte run): two-way
// softmax in fp32 exactly as written in the reference (exp(x - max) / sum), the positive
// points' probabilities accumulated in double, the positives counted, and the step's 64
// mask bits leave as one ballot.  Phase 2 is per-frustum scalar work (12 + 8 soft-max
// terms, two arg-max picks, the label conversion in double); every lane computes it and
// lanes 0..19 store one field each.  HBM-bound by construction: 8n + 248 bytes in,
// 160 + n/8 bytes out per frustum, nothing re-read.
//
// Compiled without fused-multiply-add contraction: each product and sum of the label
// conversion is rounded on its own, in the order the reference writes them.
#include <cmath>
#include <cstdint>

#include "fpn_common.h"

namespace fpn {
namespace {

constexpr int NH = 12, NS = 8, BOX = 59;
constexpr int kWavesPerBlock = 4;

// mean sizes (l,w,h) of the eight size clusters (models/model_util.py:38-48), class
// order Car, Van, Truck, Pedestrian, Person_sitting, Cyclist, Tram, Misc
__constant__ double kMeanSize[NS][3] = {
    {3.88311640418, 1.62856739989, 1.52563191462}, {5.06763659, 1.9007158, 2.20532825},
    {10.13586957, 2.58549199, 3.2520595},         {0.84422524, 0.66068622, 1.76255119},
    {0.80057803, 0.5983815, 1.27450867},          {1.76282397, 0.59706367, 1.73698127},
    {16.17150617, 2.53246914, 3.53079012},        {3.64300781, 1.54298177, 1.92320313}};

// np.argmax: first maximum, and a NaN counts as the maximum (the first NaN wins).
template <int N>
__device__ inline int argmax_first(const float *v)
{
    int best = 0;
    float bv = v[0];
    for (int i = 1; i < N; ++i) {
        const float x = v[i];
        if (x > bv || (x != x && bv == bv)) { bv = x; best = i; }
    }
    return best;
}

// max over i of softmax(v)[i] in fp32 as scripts/test.py:38-43 forms it:
// e_i = exp(v_i - max), p_i = e_i / sum(e) with the sum taken in numpy's order for a
// contiguous row (fewer than 8 terms: left to right; otherwise eight partial sums over
// i mod 8, combined pairwise, then the remaining terms left to right).
template <int N>
__device__ inline float max_softmax(const float *v)
{
    float m = v[0];
    for (int i = 1; i < N; ++i) m = fmaxf(m, v[i]);   // finite inputs: np.max
    float e[N];
    for (int i = 0; i < N; ++i) e[i] = expf(v[i] - m);
    float sum;
    if (N < 8) {
        sum = e[0];
        for (int i = 1; i < N; ++i) sum += e[i];
    } else {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = e[j];
        int i = 8;
        for (; i + 8 <= N; i += 8)
            for (int j = 0; j < 8; ++j) r[j] += e[i + j];
        sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < N; ++i) sum += e[i];
    }
    float best = e[0] / sum;
    for (int i = 1; i < N; ++i) {
        const float p = e[i] / sum;
        if (p > best || (p != p && best == best)) best = p;
    }
    return best;
}

__global__ __launch_bounds__(kWavesPerBlock *kWave) void detection_records_kernel(
    int b, int n, const float *__restrict__ logits, const float *__restrict__ box,
    const float *__restrict__ c1, const float *__restrict__ rot, double *__restrict__ rec,
    unsigned *__restrict__ seg_bits)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int f = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (f >= b) return;   // whole waves leave together: no barrier below

    // ---- phase 1: segmentation logits -> positive count, sum of positive probabilities, mask bits
    const float2 *lg = reinterpret_cast<const float2 *>(logits) + (size_t)f * n;
    const int words = (n + 31) >> 5;
    unsigned *bits = seg_bits ? seg_bits + (size_t)f * words : nullptr;
    double psum = 0.0;
    int cnt = 0;
    for (int base = 0; base < n; base += kWave) {
        const int i = base + lane;
        bool pos = false;
        if (i < n) {
            const float2 l = lg[i];
            // np.argmax over (l.x, l.y): 1 iff l.y is the strict maximum (or the first NaN)
            pos = l.y > l.x || (l.y != l.y && l.x == l.x);
            const float m = fmaxf(l.x, l.y);
            const float e0 = expf(l.x - m), e1 = expf(l.y - m);
            const float p1 = e1 / (e0 + e1);
            if (pos) { psum += (double)p1; ++cnt; }
        }
        const unsigned long long bal = __ballot(pos);
        if (bits) {
            const int w = base >> 5;
            if (lane == 0) bits[w] = (unsigned)bal;
            if (lane == 1 && w + 1 < words) bits[w + 1] = (unsigned)(bal >> 32);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        psum += __shfl_xor(psum, off, kWave);
        cnt += __shfl_xor(cnt, off, kWave);
    }

    // ---- phase 2: per-frustum picks, score and label conversion (same values on every lane)
    const float *o = box + (size_t)f * BOX;
    float hs[NH], ss[NS];
    for (int i = 0; i < NH; ++i) hs[i] = o[3 + i];
    for (int i = 0; i < NS; ++i) ss[i] = o[3 + 2 * NH + i];
    const int hc = argmax_first<NH>(hs), sc = argmax_first<NS>(ss);
    const float hp = max_softmax<NH>(hs), sp = max_softmax<NS>(ss);
    // heading_residuals = normalised * (pi/12), size_residuals = normalised * mean size,
    // both fp32 products of the model graph (models/model_util.py:229-241)
    const float hres = o[3 + NH + hc] * (float)(M_PI / NH);
    float sres[3], cen[3];
    for (int d = 0; d < 3; ++d) {
        sres[d] = o[3 + 2 * NH + NS + sc * 3 + d] * (float)kMeanSize[sc][d];
        cen[d] = o[d] + c1[(size_t)f * 3 + d];    // center = center_boxnet + stage1_center
    }
    // scripts/test.py:147-152: 0/0 = NaN for a frustum without positives
    const double mask_mean_prob = psum / (double)cnt;
    const double score = (mask_mean_prob + (double)hp + (double)sp) / 3.0;

    // scripts/provider.py:366-373 on float64 copies of the fp32 outputs
    const double l = kMeanSize[sc][0] + (double)sres[0];
    const double w = kMeanSize[sc][1] + (double)sres[1];
    const double h = kMeanSize[sc][2] + (double)sres[2];
    double angle = (double)hc * (2.0 * M_PI / (double)NH) + (double)hres;   // :77-79
    if (angle > M_PI) angle = angle - 2.0 * M_PI;                             // :80-81
    const float r = rot ? rot[f] : 0.0f;
    const double ry = angle + (double)r;
    // rotate_pc_along_y(center, -rot) (:35-48): cos and sin of the fp32 angle are fp32
    // numbers there; here they are the correctly rounded ones
    const float nr = -r;
    const double cv = (double)(float)cos((double)nr), sv = (double)(float)sin((double)nr);
    const double x = (double)cen[0], z = (double)cen[2];
    const double tx = x * cv + z * (-sv);
    const double tz = x * sv + z * cv;
    const double ty = (double)cen[1] + h / 2.0;

    double v;
    switch (lane) {
        case 0: v = h; break;
        case 1: v = w; break;
        case 2: v = l; break;
        case 3: v = tx; break;
        case 4: v = ty; break;
        case 5: v = tz; break;
        case 6: v = ry; break;
        case 7: v = score; break;
        case 8: v = (double)hc; break;
        case 9: v = (double)sc; break;
        case 10: v = (double)hres; break;
        case 11: v = (double)sres[0]; break;
        case 12: v = (double)sres[1]; break;
        case 13: v = (double)sres[2]; break;
        case 14: v = (double)cen[0]; break;
        case 15: v = (double)cen[1]; break;
        case 16: v = (double)cen[2]; break;
        case 17: v = (double)cnt; break;
        case 18: v = (double)hp; break;
        default: v = (double)sp; break;
    }
    if (lane < FPN_DETECTION_RECORD) rec[(size_t)f * FPN_DETECTION_RECORD + lane] = v;
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" int fpn_detection_records(int b, int n, const float *mask_logits, const float *box_output,
                                     const float *stage1_center, const float *rot_angle, double *records,
                                     unsigned *seg_bits, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 1, "detection_records: need n >= 1 points per frustum (b=%d n=%d)", b, n);
    if (b == 0) return FPN_OK;
    FPN_REQUIRE(mask_logits && box_output && stage1_center && records, "detection_records: null pointer");
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(mask_logits) & 7) == 0, "detection_records: mask_logits must be 8-byte aligned");
    const int blocks = (b + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(detection_records_kernel, dim3(blocks), dim3(kWavesPerBlock * kWave), 0, as_stream(stream), b,
                       n, mask_logits, box_output, stage1_center, rot_angle, records, seg_bits);
    return check_launch("fpn_detection_records");
}
