// Register-resident layer chains: a WAVE carries 32 points through consecutive
// shared-MLP layers with every activation in registers - no LDS, no barriers.
//
// The trick is to compute each layer TRANSPOSED, Y^T = W^T . X^T, on
// v_mfma_f32_32x32x2_f32:
//   A operand (32 x 2) = W^T tile: lane l supplies W[k(l>>5)][32 t + (l&31)]   (weights)
//   B operand (2 x 32) = X^T     : lane l supplies X[point l&31][k(l>>5)]      (activations)
//   D (32 x 32)        = Y^T tile: lane l, register r holds
//                        Y[point l&31][32 t + (r&3) + 8 (r>>2) + 4 (l>>5)]
// so after the layer lane (p, h) owns, for point p, the 16 channels 8q + 4h + i
// (r = 4q + i) of every 32-channel tile - and that is exactly the shape of a B
// operand: register r of tile j can be fed to the next layer's MFMA as "k-step (j, r)",
// in which the lower lane half contributes channel 32j + 8q + i and the upper half
// channel 32j + 8q + 4 + i.  The matching weight rows are baked into the weight
// layout (fpn_tile_weights_reg_f32), so the only per-MFMA operand traffic is one
// weight register per lane, streamed from L2 as coalesced 16-byte loads two chunks
// (32 MFMAs) ahead.  Bias / batch norm / ReLU run on the accumulator registers in
// place, in the issue shadow of the next tile's MFMAs.
//
// The kernel is written as a sequence of DOUBLE STEPS of 32 MFMAs: one activation tile
// against two output tiles (two weight chunks, MFMAs alternating between the two
// accumulators).  Each double step also requests the two weight chunks of the next one
// and may carry the epilogue of an earlier tile and the loads of its constants.  Steps
// are fenced (sched_barrier) and the order inside a step is pinned with
// sched_group_barrier - one load and a few vector-ALU operations after each MFMA -
// because with one wave per SIMD (the activations of 32 points through 512 -> 256
// channels need ~400 registers) nothing else hides a misplaced wait.
//
// Measured on MI355X, seg tail of 1024 frustums x 1024 points: 3.16 ms = 141.7 TFLOP/s
// (90 % of the fp32 MFMA peak) against 3.49 ms for the LDS-resident version of the same
// chain that this replaced (8 waves sharing a 64-point tile, 7 barriers per tile).
#include <algorithm>
#include <cstdlib>
#include <string>
#include <cstdint>
#include <type_traits>
#include <utility>

#include "fpn_common.h"

namespace fpn {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Waves per workgroup: one per SIMD.  The four waves of a workgroup start together and
// run the same instruction stream, so they ask for the same weight chunk at nearly the
// same time and three of the four requests hit in the CU's L1.
constexpr int kRegWaves = 4;

// Optional phase time stamps (diagnostic builds only, -DFPN_REG_STAMPS; read back by
// tools/tail_reg_stamps.hip).  Compiled out of the library.
#ifdef FPN_REG_STAMPS
__device__ unsigned long long g_reg_stamps[1 << 20];
#define FPN_STAMP(i)                                                                           \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if ((threadIdx.x & 63) == 0)                                                           \
            g_reg_stamps[((size_t)blockIdx.x * kRegWaves + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#else
#define FPN_STAMP(i)
#endif

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(std::forward<F>(f));
    }
}

// LDS executes one wave's instructions in issue order: what is needed between the ds_max updates of a row and
// the wave's own reads of it (and between those reads and the next updates) is only that the COMPILER keeps
// the order.  (A fence, even at wavefront scope, also waits for the wave's outstanding memory reads - the
// next round's inputs.)
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

struct WChunk {          // the 16 weights of a lane for the 16 MFMA steps of one
    float4 q[4];         // (output tile, input tile) pair: step r = 4q + i
};

__device__ __forceinline__ float comp(const float4 &v, int i)
{
    return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

__device__ __forceinline__ void load_chunk(WChunk &c, const float4 *__restrict__ p, unsigned lane)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) c.q[q] = p[q * 64 + lane];
}

__device__ __forceinline__ void zero(floatx16 &a)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// per-channel constants of a lane for one 32-channel tile: channels ch0 + 8q + i
struct TileConst {
    float4 q[4];
};

__device__ __forceinline__ void load_const(TileConst &c, const float *__restrict__ p, int ch0)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) c.q[q] = *reinterpret_cast<const float4 *>(p + ch0 + 8 * q);
}

typedef float floatx2 __attribute__((ext_vector_type(2)));

// a = relu(a * sc + sh) in place.  The multiply-add is written on register pairs so that
// it becomes v_pk_fma_f32 (two results per instruction): vector-ALU instructions issued
// next to the MFMA stream are not free, each costs it a few cycles.
__device__ __forceinline__ void bn_relu(floatx16 &a, const TileConst &sc, const TileConst &sh)
{
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        const floatx2 v = {a[r], a[r + 1]};
        const floatx2 s = {comp(sc.q[r >> 2], r & 3), comp(sc.q[r >> 2], (r & 3) + 1)};
        const floatx2 b = {comp(sh.q[r >> 2], r & 3), comp(sh.q[r >> 2], (r & 3) + 1)};
        const floatx2 y = v * s + b;
        a[r] = fmaxf(y[0], 0.f);
        a[r + 1] = fmaxf(y[1], 0.f);
    }
}

// Order of one step of NMFMA MFMAs: after each MFMA one memory read (while there are
// any, NVMEM of them, in program order) and, from the VALU_FROM-th MFMA on, NVALU
// vector-ALU operations; nothing crosses the step's end.  An epilogue that rides on a
// step starts a few MFMAs into it (VALU_FROM = 8), so that the constants it needs -
// requested FIRST in the step before - have had more than a step's time to arrive.
template <int NMFMA, int NVMEM, int NVALU, int VALU_FROM = 0>
__device__ __forceinline__ void sched_step()
{
#pragma unroll
    for (int m = 0; m < NMFMA; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    // MFMA
        if (m < NVMEM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                     // VMEM read
        if (NVALU > 0 && m >= VALU_FROM) __builtin_amdgcn_sched_group_barrier(0x002, NVALU, 0);   // VALU
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Two output tiles against one activation tile, MFMAs alternating between the two
// accumulators so that consecutive MFMAs are independent: 32 MFMA steps.
__device__ __forceinline__ void mfma_pair(floatx16 &acc_a, floatx16 &acc_b, const WChunk &wa, const WChunk &wb,
                                          const floatx16 &act)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(wa.q[r >> 2], r & 3), act[r], acc_a, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(wb.q[r >> 2], r & 3), act[r], acc_b, 0, 0, 0);
    }
}

// One output tile against one activation tile (the odd last tile of a layer whose width is
// an odd number of 32-channel tiles): 16 MFMA steps on one accumulator.  Back-to-back
// dependent MFMAs of this shape issue at the pipe's rate (their latency equals their issue
// time), so the single chain costs nothing extra per MFMA.
__device__ __forceinline__ void mfma_single(floatx16 &acc, const WChunk &w, const floatx16 &act)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(w.q[r >> 2], r & 3), act[r], acc, 0, 0, 0);
}

// Flipout: each frustum has its own effective weights (fpn_flipout_weights_f32), so
// the weight stream of a wave is its frustum's copy.  The workgroups of one frustum are
// then placed on ONE XCD (consecutive workgroup ids go round the eight XCDs, so id i
// handles logical block (i % 8) * (grid / 8) + i / 8): its 0.2 - 0.9 MB of weights are
// read by all its waves out of that XCD's L2 instead of out of all eight.
__device__ __forceinline__ unsigned logical_block(int per_frustum)
{
    const unsigned g = gridDim.x, i = blockIdx.x;
    return (per_frustum && g % 8 == 0) ? (i % 8) * (g / 8) + i / 8 : i;
}

// the wave's frustum as a scalar (the wave index comes from threadIdx and would
// otherwise make every weight address a vector computation)
__device__ __forceinline__ size_t scalar_index(long v)
{
    return (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)v);
}

struct SegTailRegParams {
    long rows;
    int n;
    const float *point_feat;      // (rows,64)
    const float *gshift;          // (rows/n,512): conv6's shift with the frustum part folded in
    const float4 *w6;             // chunk (t, j) at (t*2 + j) * 256 float4
    const float *sc6;
    const float4 *w7;             // chunk (t, j) at (t*16 + j) * 256
    const float *sc7, *sh7;
    const float4 *w8;             // chunk (j, t) at (j*4 + t) * 256
    const float *sc8, *sh8;
    const float4 *w9;             // chunk (j, t) at (j*4 + t) * 256
    const float *sc9, *sh9;
    const float *w10, *b10;       // (128,2) row-major, (2)
    float *logits;                // (rows,2)
    int per_frustum;              // every weight array holds one copy per frustum (Flipout)
    const unsigned *drop;         // (rows,4) keep bits of conv9's 128 outputs, or NULL
};

// The loop body of conv6-chunk kc / conv7-partial kc is 20 double steps; double step D
// multiplies one activation tile into the output tiles (2x, 2x+1), U selects which:
//   D in [0,4)   conv6 output tiles 4kc + 2(D/2) + U, input tile D%2
//   D in [4,20)  conv7 output tiles 2((D-4)/4) + U, input tile 4kc + (D-4)%4
//   D = 20       what follows: double step 0 of kc+1, or of conv8 after kc = 3
template <int D, int U>
__device__ __forceinline__ const float4 *tail_body_chunk(const SegTailRegParams &p, int kc)
{
    if constexpr (D < 4) {
        return p.w6 + (size_t)((4 * kc + 2 * (D >> 1) + U) * 2 + (D & 1)) * 256;
    } else if constexpr (D < 20) {
        return p.w7 + (size_t)((2 * ((D - 4) >> 2) + U) * 16 + 4 * kc + ((D - 4) & 3)) * 256;
    } else {
        const float4 *next_body = p.w6 + (size_t)((4 * (kc + 1) + U) * 2) * 256;
        const float4 *conv8 = p.w8 + (size_t)U * 256;
        return kc < 3 ? next_body : conv8;
    }
}

__global__ __launch_bounds__(64 * kRegWaves) void seg_tail_reg_kernel(const SegTailRegParams pin)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const long row0 = ((long)logical_block(pin.per_frustum) * kRegWaves + (threadIdx.x >> 6)) * 32;
    if (row0 >= pin.rows) return;   // waves are independent (no barriers): a surplus wave just leaves
    const long frustum = row0 / pin.n;
    SegTailRegParams p = pin;
    if (pin.per_frustum) {
        const size_t f = scalar_index(frustum);
        p.w6 += f * (64 * 512 / 4);
        p.w7 += f * (512 * 256 / 4);
        p.w8 += f * (256 * 128 / 4);
        p.w9 += f * (128 * 128 / 4);
        p.w10 += f * (128 * 2);
    }
    const float *gs = p.gshift + frustum * 512;
    FPN_STAMP(0);

    // weight chunks of double step D live in ring[2 (D % 2)], ring[2 (D % 2) + 1];
    // those of D + 1 are requested while D computes
    WChunk ring[4];
    load_chunk(ring[0], tail_body_chunk<0, 0>(p, 0), lane);
    load_chunk(ring[1], tail_body_chunk<0, 1>(p, 0), lane);

    // point_feat of this wave's 32 points, as B operands: tile j, register 4q + i =
    // channel 32j + 8q + 4h + i
    floatx16 pf[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(p.point_feat + (row0 + pt) * 64 + 32 * j + 8 * q + 4 * h);
            pf[j][4 * q + 0] = v.x; pf[j][4 * q + 1] = v.y; pf[j][4 * q + 2] = v.z; pf[j][4 * q + 3] = v.w;
        }

    floatx16 acc7[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) zero(acc7[t]);
    TileConst c7sc, c7sh;   // conv7's constants of tile 0, requested near the end of the loop
    __builtin_amdgcn_sched_barrier(0);
    FPN_STAMP(1);

#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
        floatx16 hbuf[4];
        TileConst csc[2], cgs[2];   // constants of the conv6 tile pair in flight
        static_for<0, 20>([&](auto I) {
            constexpr int D = decltype(I)::value;
            constexpr int cur = 2 * (D & 1), nxt = 2 * ((D + 1) & 1);
            if constexpr (D == 1 || D == 3) {
                // constants of the conv6 tile pair that this step completes; its epilogue
                // rides on the next step
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ch0 = 128 * kc + 32 * (2 * (D >> 1) + u) + 4 * h;
                    load_const(csc[u], p.sc6, ch0);
                    load_const(cgs[u], gs, ch0);
                }
            }
            if constexpr (D == 18) {
                load_const(c7sc, p.sc7, 4 * h);
                load_const(c7sh, p.sh7, 4 * h);
            }
            load_chunk(ring[nxt], tail_body_chunk<D + 1, 0>(p, kc), lane);
            load_chunk(ring[nxt + 1], tail_body_chunk<D + 1, 1>(p, kc), lane);
            if constexpr (D == 2 || D == 4) {   // epilogue of the pair completed by the previous step
                bn_relu(hbuf[D - 2], csc[0], cgs[0]);
                bn_relu(hbuf[D - 1], csc[1], cgs[1]);
            }
            if constexpr (D < 4) {
                // ---- conv6 tile pair D/2 of this chunk, input tile D % 2
                constexpr int tp = D >> 1;
                if constexpr ((D & 1) == 0) {
                    zero(hbuf[2 * tp]);
                    zero(hbuf[2 * tp + 1]);
                }
                mfma_pair(hbuf[2 * tp], hbuf[2 * tp + 1], ring[cur], ring[cur + 1], pf[D & 1]);
            } else {
                // ---- conv7 += H . W7[128 kc : 128 kc + 128, :], output pair (D-4)/4, input tile (D-4)%4
                constexpr int m = D - 4;
                mfma_pair(acc7[2 * (m >> 2)], acc7[2 * (m >> 2) + 1], ring[cur], ring[cur + 1], hbuf[m & 3]);
            }
            sched_step<32, (D == 1 || D == 3) ? 24 : (D == 18 ? 16 : 8), (D == 2 || D == 4) ? 4 : 0, 8>();
        });
    }

    // ---- conv8: 256 -> 128 as 16 double steps E = (input tile j = E / 2, output pair E % 2);
    //      weight chunks 2E, 2E+1 (j-major).  conv7's batch norm + ReLU of input tile j+1
    //      rides on step (j, 1); its constants are requested in step (j, 0).
    FPN_STAMP(2);
    floatx16 acc8[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) zero(acc8[t]);
    bn_relu(acc7[0], c7sc, c7sh);
    __builtin_amdgcn_sched_barrier(0);
    TileConst ca, cb;
    static_for<0, 16>([&](auto I) {
        constexpr int E = decltype(I)::value, j = E >> 1, tp = E & 1;
        constexpr int cur = 2 * (E & 1), nxt = 2 * ((E + 1) & 1);
        // the step after E = 15 is conv9's first
        if constexpr (tp == 0) {
            // next input tile of conv8, or (last j) the first input tile of conv9
            load_const(ca, j < 7 ? p.sc7 : p.sc8, (j < 7 ? 32 * (j + 1) : 0) + 4 * h);
            load_const(cb, j < 7 ? p.sh7 : p.sh8, (j < 7 ? 32 * (j + 1) : 0) + 4 * h);
        }
        load_chunk(ring[nxt], E < 15 ? p.w8 + (size_t)(2 * E + 2) * 256 : p.w9, lane);
        load_chunk(ring[nxt + 1], E < 15 ? p.w8 + (size_t)(2 * E + 3) * 256 : p.w9 + 256, lane);
        if constexpr (tp == 1) {
            if constexpr (j < 7) bn_relu(acc7[j + 1], ca, cb);
            else bn_relu(acc8[0], ca, cb);      // complete since step (7, 0)
        }
        mfma_pair(acc8[2 * tp], acc8[2 * tp + 1], ring[cur], ring[cur + 1], acc7[j]);
        sched_step<32, tp == 0 ? 16 : 8, tp == 0 ? 0 : 2, 8>();
    });
    // ---- conv9: 128 -> 128 as 8 double steps, same scheme; conv8's epilogue of input tile
    //      j+1 rides on step (j, 1); in the last step the epilogue of conv9's own tile 0
    FPN_STAMP(3);
    floatx16 acc9[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) zero(acc9[t]);
    static_for<0, 8>([&](auto I) {
        constexpr int F = decltype(I)::value, j = F >> 1, tp = F & 1;
        constexpr int cur = 2 * (F & 1), nxt = 2 * ((F + 1) & 1);
        if constexpr (tp == 0) {
            load_const(ca, j < 3 ? p.sc8 : p.sc9, (j < 3 ? 32 * (j + 1) : 0) + 4 * h);
            load_const(cb, j < 3 ? p.sh8 : p.sh9, (j < 3 ? 32 * (j + 1) : 0) + 4 * h);
        }
        if constexpr (F < 7) {
            load_chunk(ring[nxt], p.w9 + (size_t)(2 * F + 2) * 256, lane);
            load_chunk(ring[nxt + 1], p.w9 + (size_t)(2 * F + 3) * 256, lane);
        }
        if constexpr (tp == 1) {
            if constexpr (j < 3) bn_relu(acc8[j + 1], ca, cb);
            else bn_relu(acc9[0], ca, cb);      // complete since step (3, 0)
        }
        mfma_pair(acc9[2 * tp], acc9[2 * tp + 1], ring[cur], ring[cur + 1], acc8[j]);
        sched_step<32, (F < 7 ? 8 : 0) + (tp == 0 ? 8 : 0), tp == 0 ? 0 : 2, 8>();
    });
    // ---- rest of conv9's epilogue + conv10 (128 -> 2, plain bias) on the vector ALU
    FPN_STAMP(4);
    float s0 = 0.f, s1 = 0.f;
    // Monte-Carlo dropout before the 2-channel head (models/FPointNet_Bayes.py:92, rate
    // 0.5 kept active at inference): bit c of word j of the point's mask keeps channel
    // 32 j + c, kept activations count twice
    uint4 keep = make_uint4(~0u, ~0u, ~0u, ~0u);
    if (p.drop) keep = *reinterpret_cast<const uint4 *>(p.drop + (row0 + pt) * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j > 0) {
            TileConst sc, sh;
            load_const(sc, p.sc9, 32 * j + 4 * h);
            load_const(sh, p.sh9, 32 * j + 4 * h);
            bn_relu(acc9[j], sc, sh);
        }
        if (p.drop) {
            const unsigned word = (j == 0 ? keep.x : j == 1 ? keep.y : j == 2 ? keep.z : keep.w) >> (4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc9[j][r] *= (word >> (8 * (r >> 2) + (r & 3))) & 1u ? 2.f : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 *w = reinterpret_cast<const float4 *>(p.w10 + (32 * j + 8 * q + 4 * h) * 2);
            const float4 wa = w[0], wb = w[1];   // channels c, c+1 | c+2, c+3, two outputs each
            s0 = fmaf(acc9[j][4 * q + 0], wa.x, s0); s1 = fmaf(acc9[j][4 * q + 0], wa.y, s1);
            s0 = fmaf(acc9[j][4 * q + 1], wa.z, s0); s1 = fmaf(acc9[j][4 * q + 1], wa.w, s1);
            s0 = fmaf(acc9[j][4 * q + 2], wb.x, s0); s1 = fmaf(acc9[j][4 * q + 2], wb.y, s1);
            s0 = fmaf(acc9[j][4 * q + 3], wb.z, s0); s1 = fmaf(acc9[j][4 * q + 3], wb.w, s1);
        }
    }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    if (h == 0) *reinterpret_cast<float2 *>(p.logits + (row0 + pt) * 2) = make_float2(s0 + p.b10[0], s1 + p.b10[1]);
    FPN_STAMP(5);
}

// ---------------------------------------------------------------------------
// Generic layers of a register chain.  Every layer is a run of double steps; on entry
// ring[0], ring[1] hold the weight chunks of its first double step, on exit those of
// the NEXT layer's first double step (next0 / next1), so that the weight stream never
// stalls at a layer boundary.  A layer whose number of double steps is odd hands the
// stream over in ring[2], ring[3]: P0 is the ring half (0 or 1) a layer starts in.
// ---------------------------------------------------------------------------

// Transposed layer: out[NT] (32-channel tiles of the 32 points) = relu(bn(W^T . in[NJ])).
// Weights: chunk (t, j) at (t*NJ + j).  The epilogue of output pair tp rides on the second
// double step of pair tp+1; the last pair's epilogue is exposed.
template <int NJ, int NT, int P0 = 0>
__device__ __forceinline__ void layer_t(floatx16 (&out)[NT], const floatx16 (&in)[NJ], const float4 *__restrict__ w,
                                        const float *__restrict__ sc, const float *__restrict__ sh, WChunk (&ring)[4],
                                        const float4 *next0, const float4 *next1, unsigned lane, int h)
{
    // NT odd: the last "pair" is a single tile - its steps are 16 MFMAs on one accumulator
    // and request one weight chunk instead of two
    static_assert(NJ >= 2 && NT >= 2, "unsupported layer shape");
    constexpr int NP = (NT + 1) / 2;             // output tile pairs (the last one may be half)
    constexpr int kSteps = NP * NJ;
    TileConst csc[2], csh[2];
    static_for<0, kSteps>([&](auto I) {
        constexpr int S = decltype(I)::value, tp = S / NJ, j = S % NJ;
        constexpr int cur = 2 * ((S + P0) & 1), nxt = 2 * ((S + P0 + 1) & 1);
        constexpr bool ride_load = tp > 0 && j == 0, ride = tp > 0 && j == 1;
        constexpr bool pair = 2 * tp + 1 < NT;
        if constexpr (ride_load) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                load_const(csc[u], sc, 32 * (2 * (tp - 1) + u) + 4 * h);
                load_const(csh[u], sh, 32 * (2 * (tp - 1) + u) + 4 * h);
            }
        }
        if constexpr (S + 1 < kSteps) {
            constexpr int tp1 = (S + 1) / NJ, j1 = (S + 1) % NJ;
            load_chunk(ring[nxt], w + (size_t)((2 * tp1) * NJ + j1) * 256, lane);
            if constexpr (2 * tp1 + 1 < NT) load_chunk(ring[nxt + 1], w + (size_t)((2 * tp1 + 1) * NJ + j1) * 256, lane);
        } else {
            load_chunk(ring[nxt], next0, lane);
            load_chunk(ring[nxt + 1], next1, lane);
        }
        if constexpr (ride) {
            bn_relu(out[2 * (tp - 1)], csc[0], csh[0]);
            bn_relu(out[2 * (tp - 1) + 1], csc[1], csh[1]);
        }
        if constexpr (j == 0) {
            zero(out[2 * tp]);
            if constexpr (pair) zero(out[2 * tp + 1]);
        }
        if constexpr (pair) {
            mfma_pair(out[2 * tp], out[2 * tp + 1], ring[cur], ring[cur + 1], in[j]);
            sched_step<32, ride_load ? 24 : 8, ride ? 4 : 0, 8>();
        } else {
            mfma_single(out[2 * tp], ring[cur], in[j]);
            sched_step<16, ride_load ? 16 : 8, ride ? 8 : 0, 4>();
        }
    });
    constexpr int kLast = 2 * (NP - 1);          // first tile of the last pair
#pragma unroll
    for (int u = 0; u < NT - kLast; ++u) {
        load_const(csc[u], sc, 32 * (kLast + u) + 4 * h);
        load_const(csh[u], sh, 32 * (kLast + u) + 4 * h);
    }
    bn_relu(out[kLast], csc[0], csh[0]);
    if constexpr (kLast + 1 < NT) bn_relu(out[kLast + 1], csc[1], csh[1]);
    __builtin_amdgcn_sched_barrier(0);
}

// The same two-tile double step with the operands swapped: the product comes out
// UN-transposed (lane = output channel, registers = the 32 points).
__device__ __forceinline__ void mfma_pair_n(floatx16 &acc_a, floatx16 &acc_b, const WChunk &wa, const WChunk &wb,
                                            const floatx16 &act)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc_a = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r], comp(wa.q[r >> 2], r & 3), acc_a, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(act[r], comp(wb.q[r >> 2], r & 3), acc_b, 0, 0, 0);
    }
}

// max over the wave's 32 points of relu(acc * sc + sh) for the lane's channel, merged
// into gmax[channel] with an integer atomic max: the values are >= +0, and for those
// the order of the bit patterns is the order of the floats.
__device__ __forceinline__ void max_points(const floatx16 &acc, float sc, float sh, int *gmax, int h)
{
    float m = 0.f;
    const floatx2 s2 = {sc, sc}, b2 = {sh, sh};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {     // v_pk_fma_f32 + v_max3_f32 per register pair
        const floatx2 v = {acc[r], acc[r + 1]};
        const floatx2 y = v * s2 + b2;
        m = fmaxf(m, fmaxf(y[0], y[1]));
    }
    // The two lane halves hold different points of the same channel.  Each issues its own
    // update: combining them first would cost a cross-lane exchange through the LDS pipe
    // whose wait stalls the MFMA stream, and a branch on the half would split the double
    // step into several scheduling regions.
    (void)h;
    atomicMax(gmax, __float_as_int(m));
}

// The four waves of a workgroup can merge their maxima in LDS first (ds_max without a
// return value: nothing waits on it) and then issue one memory atomic per channel for
// the whole workgroup, 8x fewer than one per wave and lane half.  That needs every wave
// of the workgroup to have rows, and the pooling groups (g rows each: a frustum's
// points, a ball's samples) to tile the workgroup's 128 rows evenly; otherwise the waves
// update memory directly.
struct WgPool {
    bool on;          // workgroup-uniform
    int groups;       // pooling groups inside this workgroup's rows (1, 2 or 4)
    int local;        // this wave's group among them
    long first;       // index of the workgroup's first group
};

__device__ __forceinline__ WgPool wg_pool(long rows, int g, unsigned block)
{
    constexpr int kRows = kRegWaves * 32;
    const long r0 = (long)block * kRows;
    WgPool w;
    w.on = r0 + kRows <= rows && (g % kRows == 0 || kRows % g == 0);
    w.groups = g >= kRows ? 1 : kRows / g;
    w.local = g >= kRows ? 0 : (int)((threadIdx.x >> 6) * 32 / g);
    w.first = r0 / g;
    return w;
}

template <int C>
__device__ __forceinline__ void wg_pool_clear(int (&lmax)[4 * C], const WgPool &w)
{
    for (int i = threadIdx.x; i < w.groups * C; i += 64 * kRegWaves) lmax[i] = 0;
    __syncthreads();
}

// after the last layer: LDS maxima -> memory; the row of group k is out + (first + k) * ld
template <int C>
__device__ __forceinline__ void wg_pool_flush(int (&lmax)[4 * C], const WgPool &w, float *out, int ld, int cvalid)
{
    __syncthreads();
    for (int k = 0; k < w.groups; ++k)
        for (int ch = threadIdx.x; ch < cvalid; ch += 64 * kRegWaves)
            atomicMax(reinterpret_cast<int *>(out + (w.first + k) * ld + ch), lmax[k * C + ch]);
}

// Last layer of a stack: gmax[0:32 NT] = max(gmax, max over the 32 points of
// relu(bn(in[NJ] . W))).  Output tiles are processed four at a time (two ping-pong
// accumulator pairs) in a rolled loop; the reduction of one pair rides on the second
// double step of the other.  Weights: chunk (t, j) at (t*NJ + j).
// Channels >= cvalid (output widths padded up to a multiple of 128) are not written.
template <int NJ, int NT, int P0 = 0>
__device__ __forceinline__ void layer_max(const floatx16 (&in)[NJ], const float4 *__restrict__ w,
                                          const float *__restrict__ sc, const float *__restrict__ sh,
                                          int *gmax, WChunk (&ring)[4], unsigned lane, int h,
                                          int cvalid = 32 * NT)
{
    static_assert(NJ >= 2 && NT % 4 == 0, "unsupported layer shape");
    const int c = lane & 31;
    floatx16 acc[2][2];
    zero(acc[1][0]);
    zero(acc[1][1]);
    float rsc[2], rsh[2];
#pragma unroll 1
    for (int t0 = 0; t0 < NT; t0 += 4) {
        static_for<0, 2 * NJ>([&](auto I) {
            constexpr int S = decltype(I)::value, set = S / NJ, j = S % NJ;
            constexpr int cur = 2 * ((S + P0) & 1), nxt = 2 * ((S + P0 + 1) & 1);
            // the other accumulator pair holds the previous two tiles; the very first time
            // there are none (and padded tiles are never wanted): constants of zero then make
            // the reduction a no-op (max with 0) without a branch in the middle of the step
            const int tcand = set == 0 ? t0 - 2 : t0;
            const bool has_prev = (set == 1 || t0 > 0) && 32 * tcand < cvalid;
            const int tprev = has_prev ? tcand : 0;
            if constexpr (j == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int ch = 32 * (tprev + u) + c;
                    rsc[u] = sc[ch];      // only requested here: touching the values in this
                    rsh[u] = sh[ch];      // step would wait for every load in flight
                }
                zero(acc[set][0]);
                zero(acc[set][1]);
            }
            // chunks of the next double step; past the end of the layer: re-read (harmless)
            constexpr int set1 = ((S + 1) / NJ) & 1, j1 = (S + 1) % NJ;
            const int tn = (S + 1 < 2 * NJ ? t0 : (t0 + 4 < NT ? t0 + 4 : t0)) + 2 * set1;
            load_chunk(ring[nxt], w + (size_t)(tn * NJ + j1) * 256, lane);
            load_chunk(ring[nxt + 1], w + (size_t)((tn + 1) * NJ + j1) * 256, lane);
            if constexpr (j == 1) {
                max_points(acc[set ^ 1][0], has_prev ? rsc[0] : 0.f, has_prev ? rsh[0] : 0.f, gmax + 32 * tprev + c, h);
                max_points(acc[set ^ 1][1], has_prev ? rsc[1] : 0.f, has_prev ? rsh[1] : 0.f,
                           gmax + 32 * (tprev + 1) + c, h);
            }
            mfma_pair_n(acc[set][0], acc[set][1], ring[cur], ring[cur + 1], in[j]);
            sched_step<32, j == 0 ? 12 : 8, j == 1 ? 5 : 0, 8>();
        });
    }
    if (32 * (NT - 2) < cvalid) {
        max_points(acc[1][0], sc[32 * (NT - 2) + c], sh[32 * (NT - 2) + c], gmax + 32 * (NT - 2) + c, h);
        max_points(acc[1][1], sc[32 * (NT - 1) + c], sh[32 * (NT - 1) + c], gmax + 32 * (NT - 1) + c, h);
    }
}

// First layer of a stack (K = 3..8 input channels straight from memory): out[NT] =
// relu(bn(W1^T . x^T)), transposed.  x0..x3: the lane's inputs for MFMA steps 0..3, i.e.
// channels 2s + h of its point (zero beyond the channel count); w1 (K, 32 NT) row-major.
// pre: NULL, or the lane's row of products that other inputs of the layer have already
// contributed (32 NT floats, before scale and shift): the accumulators start from them.
template <int NT, int NSTEP>
__device__ __forceinline__ void layer_first(floatx16 (&out)[NT], const float (&x)[NSTEP], const float *__restrict__ w1,
                                            int k_total, const float *__restrict__ sc, const float *__restrict__ sh,
                                            unsigned lane, int h, const float *__restrict__ pre = nullptr)
{
    const int c = lane & 31;
    if (pre) {          // (kernel-uniform) register r of tile t is channel 32 t + 8 (r >> 2) + 4 h + (r & 3)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(pre + 32 * t + 8 * q + 4 * h);
                out[t][4 * q + 0] = v.x; out[t][4 * q + 1] = v.y; out[t][4 * q + 2] = v.z; out[t][4 * q + 3] = v.w;
            }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (!pre) zero(out[t]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int k = 2 * s + h;
            // (an unconditional load - row 0 stands in past the last channel - and a select: a load inside a branch
            // makes the wave wait for every load requested before it, a trip to memory per MFMA step here)
            const float wk = w1[(k < k_total ? k : 0) * (32 * NT) + 32 * t + c];
            const float wv = k < k_total ? wk : 0.f;
            out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv, x[s], out[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        TileConst csc, csh;
        load_const(csc, sc, 32 * t + 4 * h);
        load_const(csh, sh, 32 * t + 4 * h);
        bn_relu(out[t], csc, csh);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// The same layer with its weights and constants loaded ahead: they are the same for every
// point of the wave and depend on nothing the kernel computes, so a kernel whose inputs come
// through a chain of dependent loads (ball index -> coordinates) issues load() BEFORE that
// chain and the latencies overlap (inside layer_first the loads sit between the MFMAs).
template <int NT, int NSTEP>
struct FirstLayerF32 {
    float wv[NT][NSTEP];
    TileConst csc[NT], csh[NT];

    __device__ __forceinline__ void load(const float *__restrict__ w1, int k_total, const float *__restrict__ sc,
                                         const float *__restrict__ sh, unsigned lane, int h)
    {
        const int c = lane & 31;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                const int k = 2 * s + h;
                const float wk = w1[(k < k_total ? k : 0) * (32 * NT) + 32 * t + c];      // (unconditional, as in layer_first)
                wv[t][s] = k < k_total ? wk : 0.f;
            }
            load_const(csc[t], sc, 32 * t + 4 * h);
            load_const(csh[t], sh, 32 * t + 4 * h);
        }
    }

    // pre: as for layer_first
    __device__ __forceinline__ void apply(floatx16 (&out)[NT], const float (&x)[NSTEP], int h,
                                          const float *__restrict__ pre = nullptr) const
    {
        if (pre) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *reinterpret_cast<const float4 *>(pre + 32 * t + 8 * q + 4 * h);
                    out[t][4 * q + 0] = v.x; out[t][4 * q + 1] = v.y; out[t][4 * q + 2] = v.z; out[t][4 * q + 3] = v.w;
                }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!pre) zero(out[t]);
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][s], x[s], out[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) bn_relu(out[t], csc[t], csh[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
};

// ---------------------------------------------------------------------------
// seg head: conv1 .. conv5 + max over points
// ---------------------------------------------------------------------------
struct SegHeadRegParams {
    long rows;
    int n, cpc;
    const float *pc;
    const float *w1, *sc1, *sh1;           // (cpc,64) row-major
    const float4 *w2; const float *sc2, *sh2;   // chunked, output-tile major
    const float4 *w3; const float *sc3, *sh3;
    const float4 *w4; const float *sc4, *sh4;
    const float4 *w5; const float *sc5, *sh5;
    float *point_feat;                     // (rows,64)
    float *gmax;                           // (rows/n, ldg): running maximum of conv5's output
    int ldg;
    int per_frustum;                       // every weight array holds one copy per frustum (Flipout)
};

// The head needs few enough registers for two waves per SIMD (two workgroups per CU).
// 8-wave workgroups - both waves of a SIMD walking the weight stream together - were
// measured slower (2.48 vs 2.36 ms): the two waves then also stall together.
constexpr int kHeadWaves = kRegWaves;

__global__ __launch_bounds__(64 * kHeadWaves) void seg_head_reg_kernel(const SegHeadRegParams pin)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const unsigned block = logical_block(pin.per_frustum);
    const long row0 = ((long)block * kHeadWaves + (threadIdx.x >> 6)) * 32;
    __shared__ int lmax[4 * 1024];
    const WgPool pool = wg_pool(pin.rows, pin.n, block);
    if (!pool.on && row0 >= pin.rows) return;
    SegHeadRegParams p = pin;
    if (pin.per_frustum) {
        const size_t f = scalar_index(row0 / pin.n);
        p.w1 += f * (size_t)(pin.cpc * 64);
        p.w2 += f * (64 * 64 / 4);
        p.w3 += f * (64 * 64 / 4);
        p.w4 += f * (64 * 128 / 4);
        p.w5 += f * (128 * 1024 / 4);
    }
    if (pool.on) wg_pool_clear<1024>(lmax, pool);
    int *gmax = reinterpret_cast<int *>(p.gmax + (row0 / p.n) * p.ldg);
    FPN_STAMP(0);

    WChunk ring[4];
    load_chunk(ring[0], p.w2, lane);                 // conv2 chunks (0,0), (1,0)
    load_chunk(ring[1], p.w2 + 2 * 256, lane);

    // conv1: lane (point, h) feeds channels h, 2+h, 4+h, 6+h of its point
    float x[4];
    const float *src = p.pc + (row0 + pt) * p.cpc;
#pragma unroll
    for (int s = 0; s < 4; ++s) {       // (unconditional loads, as in layer_first)
        const float xk = src[2 * s + h < p.cpc ? 2 * s + h : 0];
        x[s] = 2 * s + h < p.cpc ? xk : 0.f;
    }
    floatx16 a1[2], a2[2], a3[2], a4[4];
    layer_first<2, 4>(a1, x, p.w1, p.cpc, p.sc1, p.sh1, lane, h);
    FPN_STAMP(1);
    layer_t<2, 2>(a2, a1, p.w2, p.sc2, p.sh2, ring, p.w3, p.w3 + 2 * 256, lane, h);
    layer_t<2, 2>(a3, a2, p.w3, p.sc3, p.sh3, ring, p.w4, p.w4 + 2 * 256, lane, h);
    FPN_STAMP(2);
    // conv3's output is point_feat (models/frustum_pointnets_v1.py:49): tile t, register
    // 4q + i = channel 32t + 8q + 4h + i
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(p.point_feat + (row0 + pt) * 64 + 32 * t + 8 * q + 4 * h) =
                make_float4(a3[t][4 * q], a3[t][4 * q + 1], a3[t][4 * q + 2], a3[t][4 * q + 3]);
    FPN_STAMP(3);
    layer_t<2, 4>(a4, a3, p.w4, p.sc4, p.sh4, ring, p.w5, p.w5 + 4 * 256, lane, h);
    FPN_STAMP(4);
    if (pool.on) {     // two copies of the layer: LDS and memory atomics are different instructions
        layer_max<4, 32>(a4, p.w5, p.sc5, p.sh5, lmax + pool.local * 1024, ring, lane, h);
        wg_pool_flush<1024>(lmax, pool, p.gmax, p.ldg, 1024);
    } else {
        layer_max<4, 32>(a4, p.w5, p.sc5, p.sh5, gmax, ring, lane, h);
    }
    FPN_STAMP(5);
}

// ---------------------------------------------------------------------------
// object-point stacks: T-Net 3 -> 128 -> 128 -> 256 + max, box net 3 -> 128 -> 128 ->
// 256 -> 512 + max
// ---------------------------------------------------------------------------
struct ObjChainRegParams {
    long rows;
    int m;
    const float *obj, *delta;
    const float *w1, *sc1, *sh1;                // (3,128) row-major
    const float4 *w2; const float *sc2, *sh2;
    const float4 *w3; const float *sc3, *sh3;
    const float4 *w4; const float *sc4, *sh4;   // box net only
    float *gmax;                                // (rows/m, ldg)
    int ldg;
    int per_frustum;                            // every weight array holds one copy per frustum (Flipout)
};

template <bool BOX>
__global__ __launch_bounds__(64 * kRegWaves) void obj_chain_reg_kernel(const ObjChainRegParams pin)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const unsigned block = logical_block(pin.per_frustum);
    const long row0 = ((long)block * kRegWaves + (threadIdx.x >> 6)) * 32;
    constexpr int C = BOX ? 512 : 256;
    __shared__ int lmax[4 * C];
    const WgPool pool = wg_pool(pin.rows, pin.m, block);
    if (!pool.on && row0 >= pin.rows) return;
    ObjChainRegParams p = pin;
    if (pin.per_frustum) {
        const size_t f = scalar_index(row0 / pin.m);
        p.w1 += f * (3 * 128);
        p.w2 += f * (128 * 128 / 4);
        p.w3 += f * (128 * 256 / 4);
        if (BOX) p.w4 += f * (256 * 512 / 4);
    }
    if (pool.on) wg_pool_clear<C>(lmax, pool);
    const long frustum = row0 / p.m;
    int *gmax = reinterpret_cast<int *>(p.gmax + frustum * p.ldg);

    WChunk ring[4];
    load_chunk(ring[0], p.w2, lane);                 // layer 2 chunks (0,0), (1,0)
    load_chunk(ring[1], p.w2 + 4 * 256, lane);

    // layer 1: coordinates h and 2+h of the lane's point (minus the frustum's delta)
    float x[2];
    const float *src = p.obj + (row0 + pt) * 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int k = 2 * s + h;
        // (unconditional loads, as in layer_first; without a delta the frustum's own point stands in)
        const int kc = k < 3 ? k : 0;
        const float xk = src[kc], dk = (p.delta ? p.delta + frustum * 3 : src)[kc];
        x[s] = k < 3 ? xk - (p.delta ? dk : 0.f) : 0.f;
    }
    floatx16 a1[4], a2[4];
    layer_first<4, 2>(a1, x, p.w1, 3, p.sc1, p.sh1, lane, h);
    layer_t<4, 4>(a2, a1, p.w2, p.sc2, p.sh2, ring, p.w3, p.w3 + 4 * 256, lane, h);
    if constexpr (!BOX) {
        if (pool.on) layer_max<4, 8>(a2, p.w3, p.sc3, p.sh3, lmax + pool.local * C, ring, lane, h);
        else layer_max<4, 8>(a2, p.w3, p.sc3, p.sh3, gmax, ring, lane, h);
    } else {
        floatx16 a3[8];
        layer_t<4, 8>(a3, a2, p.w3, p.sc3, p.sh3, ring, p.w4, p.w4 + 8 * 256, lane, h);
        if (pool.on) layer_max<8, 16>(a3, p.w4, p.sc4, p.sh4, lmax + pool.local * C, ring, lane, h);
        else layer_max<8, 16>(a3, p.w4, p.sc4, p.sh4, gmax, ring, lane, h);
    }
    if (pool.on) wg_pool_flush<C>(lmax, pool, p.gmax, p.ldg, C);
}

// ---------------------------------------------------------------------------
// Set-abstraction module (models/pointnet_util.py:86-195) as one kernel: grouping
// (gather of the ball's points, translation to the centroid, concatenation with the
// point features), the three shared-MLP layers and the max over the ball's samples.
// A wave takes 32 consecutive samples of one ball; the grouped tensor never exists.
//   NJF > 0: the point features (32 NJF channels) are gathered as activation tiles and
//            the 3 relative coordinates ride in one more tile (its other 29 channels are
//            zero, as are the matching rows of the caller's padded first-layer weights);
//   NJF = 0: at most 8 input channels [dx,dy,dz, features...]: first layer from memory.
//            With `pre`, wide point features have gone through their rows of the first
//            layer ONCE PER POINT (the layer is linear before its scale, shift and ReLU, and
//            a point is gathered into many balls: 8,192 gathered rows from 128 points per
//            frustum in the second multi-scale level); the kernel gathers that row of
//            products as the accumulators' starting values and adds the coordinates' part.
// Layer widths are 32 NT1, 32 NT2, 32 NT3 (the caller pads narrower layers with zero
// weights / scale / shift); columns >= c3 of the last layer are not written.
// ---------------------------------------------------------------------------
struct SaChainParams {
    long rows;            // b * m * ns grouped rows
    int n, m, ns, cf;     // dataset points, balls per frustum, samples per ball, feature channels
    const float *xyz;     // (b,n,3)
    const float *new_xyz; // (b,m,3)
    const float *feat;    // (b,n,cf) or NULL
    const float *pre;     // NJF = 0: (b*n, ldp >= 32 NT1) what wide point features contribute to the first layer, or NULL
    int ldp;
    const int *idx;       // (b,m,ns)
    const float *w1; const float *sc1, *sh1;     // NJF > 0: chunked ((NJF+1)*32, 32 NT1); NJF = 0: plain (3+cf, 32 NT1)
    const float4 *w2; const float *sc2, *sh2;
    const float4 *w3; const float *sc3, *sh3;
    float *out;           // (b*m, ldo): running maximum, columns [0, c3) of the caller's slice
    int ldo, c3;
};

// (NJF = 0: all four shapes fit two waves per SIMD - the 128-128-256 one with ten registers squeezed out)
template <int NJF, int NT1, int NT2, int NT3>
__global__ __launch_bounds__(64 * kRegWaves, NJF == 0 ? 2 : 1) void sa_chain_kernel(const SaChainParams p)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const unsigned block = logical_block(1);      // a frustum's balls gather from the same arrays: one XCD
    const long row0 = ((long)block * kRegWaves + (threadIdx.x >> 6)) * 32;
    constexpr int C = 32 * NT3;
    __shared__ int lmax[4 * C];
    const WgPool pool = wg_pool(p.rows, p.ns, block);
    if (!pool.on && row0 >= p.rows) return;
    if (pool.on) wg_pool_clear<C>(lmax, pool);
    const long ball = row0 / p.ns;                    // = b * m + j
    const long b = ball / p.m;
    const int pid = p.idx[row0 + pt];                 // this lane's neighbour
    const float *ctr = p.new_xyz + ball * 3;
    const float *src = p.xyz + (b * p.n + pid) * 3;
    int *gmax = reinterpret_cast<int *>(p.out + ball * p.ldo);
#ifdef FPN_SA_STAMPS      // timing-only build: cycle stamps of one wave into the first output row (wrong results)
    long long stamp[6];
    stamp[0] = clock64();
#define FPN_STAMP(i) stamp[i] = clock64()
#else
#define FPN_STAMP(i)
#endif

    WChunk ring[4];
    constexpr int NJ1 = NJF > 0 ? NJF + 1 : 2;       // input tiles of layer 1 (unused when NJF = 0)
    floatx16 a1[NT1], a2[NT2];
    if constexpr (NJF > 0) {
        const float4 *w1c = reinterpret_cast<const float4 *>(p.w1);
        load_chunk(ring[0], w1c, lane);
        load_chunk(ring[1], w1c + (size_t)NJ1 * 256, lane);
        floatx16 in[NJ1];
        const float *f = p.feat + (b * p.n + pid) * (long)p.cf;
#pragma unroll
        for (int t = 0; t < NJF; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(f + 32 * t + 8 * q + 4 * h);
                in[t][4 * q + 0] = v.x; in[t][4 * q + 1] = v.y; in[t][4 * q + 2] = v.z; in[t][4 * q + 3] = v.w;
            }
        zero(in[NJF]);
        // channels 0..2 of the last tile live in registers 0..2 of the lower lane half
        in[NJF][0] = h == 0 ? src[0] - ctr[0] : 0.f;
        in[NJF][1] = h == 0 ? src[1] - ctr[1] : 0.f;
        in[NJF][2] = h == 0 ? src[2] - ctr[2] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        layer_t<NJ1, NT1, 0>(a1, in, w1c, p.sc1, p.sh1, ring, p.w2, p.w2 + (size_t)NT1 * 256, lane, h);
        constexpr int P1 = ((NT1 / 2) * NJ1) & 1;
        layer_t<NT1, NT2, P1>(a2, a1, p.w2, p.sc2, p.sh2, ring, p.w3, p.w3 + (size_t)NT2 * 256, lane, h);
        constexpr int P2 = (P1 + (NT2 / 2) * NT1) & 1;
        if (pool.on) layer_max<NT2, NT3, P2>(a2, p.w3, p.sc3, p.sh3, lmax + pool.local * C, ring, lane, h, p.c3);
        else layer_max<NT2, NT3, P2>(a2, p.w3, p.sc3, p.sh3, gmax, ring, lane, h, p.c3);
    } else {
        FirstLayerF32<NT1, 4> first;
        first.load(p.w1, 3 + p.cf, p.sc1, p.sh1, lane, h);
        load_chunk(ring[0], p.w2, lane);
        load_chunk(ring[1], p.w2 + (size_t)NT1 * 256, lane);
        // input channel k = 2s + h of this lane: dx, dy, dz, then the point's features
        float x[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 2 * s + h;
            float v = 0.f;
            if (k < 3) v = src[k] - ctr[k];
            else if (k < 3 + p.cf) v = p.feat[(b * p.n + pid) * (long)p.cf + k - 3];
            x[s] = v;
        }
        FPN_STAMP(1);
        first.apply(a1, x, h, p.pre ? p.pre + (b * p.n + pid) * (long)p.ldp : nullptr);
        FPN_STAMP(2);
        layer_t<NT1, NT2, 0>(a2, a1, p.w2, p.sc2, p.sh2, ring, p.w3, p.w3 + (size_t)NT2 * 256, lane, h);
        FPN_STAMP(3);
        constexpr int P2 = ((NT2 / 2) * NT1) & 1;
        if (pool.on) layer_max<NT2, NT3, P2>(a2, p.w3, p.sc3, p.sh3, lmax + pool.local * C, ring, lane, h, p.c3);
        else layer_max<NT2, NT3, P2>(a2, p.w3, p.sc3, p.sh3, gmax, ring, lane, h, p.c3);
        FPN_STAMP(4);
    }
    if (pool.on) wg_pool_flush<C>(lmax, pool, p.out, p.ldo, p.c3);
#ifdef FPN_SA_STAMPS
    stamp[5] = clock64();
    if (NJF == 0 && blockIdx.x == gridDim.x / 2 + 8 && threadIdx.x == 64)
        for (int i = 0; i < 6; ++i) reinterpret_cast<long long *>(p.out)[i] = stamp[i];
#endif
}

// ---------------------------------------------------------------------------
// sa_chain_kernel's NJF = 0 form as a PERSISTENT kernel, for the stacks whose second and
// third layer fit LDS (64-64-128: 48 KiB, 64-96-128: 72 KiB).  A wave launched per 32 samples
// spends a quarter to a third of its life outside MFMAs - ball index -> coordinates -> first
// layer, the first weight chunks from memory, the workgroup's barriers around the pooled
// maxima.  Here a workgroup copies both layers' chunks and constants into LDS once; a wave
// walks whole balls, 32 samples a round, its weight ring refilled from LDS; the only memory
// reads inside a round are the next round's coordinates (and first-layer products) and the
// round after's ball indices, not waited for until the next round begins; a ball's maxima meet
// in the wave's own LDS row and the wave stores the finished row itself (no memory atomics,
// no barrier after the set-up).  Same arithmetic as sa_chain_kernel, operation by operation.
// ---------------------------------------------------------------------------
// W3LDS = false (128-128-256: 64 + 128 KiB of weights): only the second layer lies in LDS, the third layer's
// chunks come from memory through the ring as in sa_chain_kernel; one workgroup per CU (the registers of
// four accumulating tile sets and the next round's products do not fit two).
template <int NT1, int NT2, int NT3, bool PRE, bool W3LDS = true>
__global__ __launch_bounds__(64 * kRegWaves, W3LDS ? 2 : 1) void sa_persist_kernel(const SaChainParams p, unsigned balls,
                                                                                   unsigned balls_per_wave)
{
    constexpr int C = 32 * NT3, W2 = NT1 * NT2 * 256, W3 = W3LDS ? NT2 * NT3 * 256 : 0;     // float4 per layer in LDS
    extern __shared__ float4 sa_persist_lds[];
    float4 *lw2 = sa_persist_lds;
    const float4 *lw3 = W3LDS ? lw2 + W2 : p.w3;
    float *lsc2 = reinterpret_cast<float *>(lw2 + W2 + W3), *lsh2 = lsc2 + 32 * NT2, *lsc3 = lsh2 + 32 * NT2, *lsh3 = lsc3 + C;
    float *lsc1 = lsh3 + C, *lsh1 = lsc1 + 32 * NT1;
    int *lmax = reinterpret_cast<int *>(lsh1 + 32 * NT1) + (threadIdx.x >> 6) * C;
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < W2; i += 64 * kRegWaves) lw2[i] = p.w2[i];
    for (int i = threadIdx.x; i < W3; i += 64 * kRegWaves) lw2[W2 + i] = p.w3[i];
    for (int i = threadIdx.x; i < 32 * NT2; i += 64 * kRegWaves) {
        lsc2[i] = p.sc2[i];
        lsh2[i] = p.sh2[i];
    }
    for (int i = threadIdx.x; i < C; i += 64 * kRegWaves) {
        lsc3[i] = p.sc3[i];
        lsh3[i] = p.sh3[i];
    }
    for (int i = threadIdx.x; i < 32 * NT1; i += 64 * kRegWaves) {
        lsc1[i] = p.sc1[i];
        lsh1[i] = p.sh1[i];
    }
    for (int i = lane; i < C; i += 64) lmax[i] = 0;
    __syncthreads();

    const unsigned gw = __builtin_amdgcn_readfirstlane(logical_block(1) * kRegWaves + (threadIdx.x >> 6));
    const unsigned ball0 = gw * balls_per_wave, ball1 = min(balls, ball0 + balls_per_wave);
    if (ball0 >= ball1) return;
    const unsigned rpb = p.ns / 32;             // rounds per ball
    const unsigned c0 = ball0 * rpb, c1 = ball1 * rpb;

    int pid = p.idx[(size_t)c0 * 32 + pt];
    // the first layer's weights stay in registers (one per tile and MFMA step); its scales and shifts are read
    // from LDS every round
    float wv[NT1][4];
#pragma unroll
    for (int t = 0; t < NT1; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 2 * s + h;
            const float wk = p.w1[(k < 3 + p.cf ? k : 0) * (32 * NT1) + 32 * t + (lane & 31)];      // (unconditional, as in layer_first)
            wv[t][s] = k < 3 + p.cf ? wk : 0.f;
        }
    // input channel k = 2 s + h of this lane: dx, dy, dz, then the point's direct features
    float xr[4], cr[4];
    float4 pr[PRE ? NT1 : 1][4];                // PRE: the next round's row of first-layer products
    auto load_x = [&](unsigned c) {             // raw values: nothing here waits for a load it has issued itself
        const unsigned ball = c / rpb, b = ball / p.m;
        const size_t q = (size_t)b * p.n + pid;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 2 * s + h;
            float v = 0.f, ctr = 0.f;
            if (k < 3) {
                v = p.xyz[q * 3 + k];
                ctr = p.new_xyz[(size_t)ball * 3 + k];
            } else if (k < 3 + p.cf) {
                v = p.feat[q * p.cf + k - 3];
            }
            xr[s] = v;
            cr[s] = ctr;
        }
        if constexpr (PRE) {
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    pr[t][i] = *reinterpret_cast<const float4 *>(p.pre + q * p.ldp + 32 * t + 8 * i + 4 * h);
        }
    };
    load_x(c0);
    pid = p.idx[(size_t)min(c0 + 1, c1 - 1) * 32 + pt];

    WChunk ring[4];
    unsigned ball = ball0, in_ball = 0;
#pragma unroll 1
    for (unsigned c = c0; c < c1; ++c) {
        float x[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) x[s] = xr[s] - cr[s];
        floatx16 a1[NT1], a2[NT2];
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            if constexpr (PRE) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a1[t][4 * i + 0] = pr[t][i].x; a1[t][4 * i + 1] = pr[t][i].y;
                    a1[t][4 * i + 2] = pr[t][i].z; a1[t][4 * i + 3] = pr[t][i].w;
                }
            } else {
                zero(a1[t]);
            }
        }
        load_x(min(c + 1, c1 - 1));
        pid = p.idx[(size_t)min(c + 2, c1 - 1) * 32 + pt];
        load_chunk(ring[0], lw2, lane);
        load_chunk(ring[1], lw2 + (size_t)NT1 * 256, lane);
        __builtin_amdgcn_sched_barrier(0);

#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) a1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][s], x[s], a1[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
            TileConst csc, csh;
            load_const(csc, lsc1, 32 * t + 4 * h);
            load_const(csh, lsh1, 32 * t + 4 * h);
            bn_relu(a1[t], csc, csh);
        }
        __builtin_amdgcn_sched_barrier(0);
        layer_t<NT1, NT2, 0>(a2, a1, lw2, lsc2, lsh2, ring, lw3, lw3 + (size_t)NT2 * 256, lane, h);
        constexpr int P2 = (((NT2 + 1) / 2) * NT1) & 1;
        layer_max<NT2, NT3, P2>(a2, lw3, lsc3, lsh3, lmax, ring, lane, h, p.c3);

        if (++in_ball == rpb) {                 // the ball is complete: its row leaves, the LDS row starts again from zero
            lds_order();
            float *out = p.out + (size_t)ball * p.ldo;
#pragma unroll
            for (int ch = lane; ch < C; ch += 64) {
                const int v = lmax[ch];
                lmax[ch] = 0;
                if (ch < p.c3) out[ch] = __int_as_float(v);
            }
            lds_order();
            in_ball = 0;
            ++ball;
        }
    }
}

// ---------------------------------------------------------------------------
// The narrow stack of the first multi-scale level, (3 + cf) -> 32 -> 32 -> 64 + max, at its
// own widths.  Padded to the 64-64-128 kernel it ran 200 MFMAs per 32 rows, three quarters
// of them on zeros; here it is 4 + 16 + 32.  All its weights fit a lane's registers (one
// chunk for the 32 x 32 layer, two for the 32 x 64 one), so the kernel is persistent: a wave
// loads them once and walks consecutive groups of 32 samples, the next group's coordinates
// and the one after's ball indices in flight while it multiplies.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kRegWaves, 2) void sa_small_kernel(const SaChainParams p, unsigned groups,
                                                                     unsigned groups_per_wave)
{
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5, c = lane & 31;
    const unsigned gw = __builtin_amdgcn_readfirstlane(logical_block(1) * kRegWaves + (threadIdx.x >> 6));
    const unsigned g0 = gw * groups_per_wave, g1 = min(groups, g0 + groups_per_wave);
    if (g0 >= g1) return;
    const unsigned gpb = p.ns / 32;         // groups per ball

    int pid = p.idx[(size_t)g0 * 32 + pt];
    FirstLayerF32<1, 4> first;
    first.load(p.w1, 3 + p.cf, p.sc1, p.sh1, lane, h);
    WChunk w2, w3a, w3b;
    load_chunk(w2, p.w2, lane);
    load_chunk(w3a, p.w3, lane);
    load_chunk(w3b, p.w3 + 256, lane);
    TileConst csc2, csh2;
    load_const(csc2, p.sc2, 4 * h);
    load_const(csh2, p.sh2, 4 * h);
    const float sc3a = p.sc3[c], sh3a = p.sh3[c], sc3b = p.sc3[32 + c], sh3b = p.sh3[32 + c];

    // input channel k = 2 s + h of this lane: dx, dy, dz, then the point's direct features
    float xr[4], cr[4];
    auto load_x = [&](unsigned g) {         // raw values: nothing here waits for a load it has issued itself
        const unsigned ball = g / gpb, b = ball / p.m;
        const size_t q = (size_t)b * p.n + pid;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 2 * s + h;
            float v = 0.f, ctr = 0.f;
            if (k < 3) {
                v = p.xyz[q * 3 + k];
                ctr = p.new_xyz[(size_t)ball * 3 + k];
            } else if (k < 3 + p.cf) {
                v = p.feat[q * p.cf + k - 3];
            }
            xr[s] = v;
            cr[s] = ctr;
        }
    };
    load_x(g0);
    pid = p.idx[(size_t)min(g0 + 1, g1 - 1) * 32 + pt];
#pragma unroll 1
    for (unsigned g = g0; g < g1; ++g) {
        float x[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) x[s] = xr[s] - cr[s];
        load_x(min(g + 1, g1 - 1));
        pid = p.idx[(size_t)min(g + 2, g1 - 1) * 32 + pt];
        __builtin_amdgcn_sched_barrier(0);

        floatx16 a1[1], a2, acc0, acc1;
        first.apply(a1, x, h);
        zero(a2);
        mfma_single(a2, w2, a1[0]);
        bn_relu(a2, csc2, csh2);
        zero(acc0);
        zero(acc1);
        mfma_pair_n(acc0, acc1, w3a, w3b, a2);
        int *gmax = reinterpret_cast<int *>(p.out + (size_t)(g / gpb) * p.ldo);
        max_points(acc0, sc3a, sh3a, gmax + c, h);
        max_points(acc1, sc3b, sh3b, gmax + 32 + c, h);
    }
}

// ---------------------------------------------------------------------------
// Per-point head after a feature-propagation step: the concatenation
// [interpolated features (32 NJA channels) | the point's own features (cb <= 32)]
// (models/pointnet_util.py:218) through three 128-wide layers and a final 128 -> 2
// product: the last FP module's MLP and the two 1-wide convolutions that follow it in a
// PointNet++ segmentation net, as one kernel.  The concatenated tensor never exists.
// ---------------------------------------------------------------------------
struct FpHeadParams {
    long rows;
    int cb;
    const float *xa, *xb;                       // (rows, 32 NJA), (rows, cb)
    const float4 *w1; const float *sc1, *sh1;   // chunked ((NJA+1)*32, 128): rows [xa | xb | zeros]
    const float4 *w2; const float *sc2, *sh2;   // chunked (128,128)
    const float4 *w3; const float *sc3, *sh3;
    const float *w4, *b4;                       // (128,2) row-major, (2)
    float *logits;                              // (rows,2)
};

template <int NJA>
__global__ __launch_bounds__(64 * kRegWaves, 2) void fp_head_reg_kernel(const FpHeadParams p)
{
    constexpr int NJ = NJA + 1;
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    const long row0 = ((long)blockIdx.x * kRegWaves + (threadIdx.x >> 6)) * 32;
    if (row0 >= p.rows) return;

    WChunk ring[4];
    load_chunk(ring[0], p.w1, lane);
    load_chunk(ring[1], p.w1 + (size_t)NJ * 256, lane);
    floatx16 in[NJ], a1[4], a2[4], a3[4];
    const float *xa = p.xa + (row0 + pt) * (32 * NJA);
#pragma unroll
    for (int t = 0; t < NJA; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(xa + 32 * t + 8 * q + 4 * h);
            in[t][4 * q + 0] = v.x; in[t][4 * q + 1] = v.y; in[t][4 * q + 2] = v.z; in[t][4 * q + 3] = v.w;
        }
    const float *xb = p.xb + (row0 + pt) * (long)p.cb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = 8 * (r >> 2) + 4 * h + (r & 3);
        in[NJA][r] = ch < p.cb ? xb[ch] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    layer_t<NJ, 4>(a1, in, p.w1, p.sc1, p.sh1, ring, p.w2, p.w2 + 4 * 256, lane, h);
    layer_t<4, 4>(a2, a1, p.w2, p.sc2, p.sh2, ring, p.w3, p.w3 + 4 * 256, lane, h);
    layer_t<4, 4>(a3, a2, p.w3, p.sc3, p.sh3, ring, p.w3, p.w3 + 4 * 256, lane, h);   // nothing follows: re-read
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 *w = reinterpret_cast<const float4 *>(p.w4 + (32 * j + 8 * q + 4 * h) * 2);
            const float4 wa = w[0], wb = w[1];
            s0 = fmaf(a3[j][4 * q + 0], wa.x, s0); s1 = fmaf(a3[j][4 * q + 0], wa.y, s1);
            s0 = fmaf(a3[j][4 * q + 1], wa.z, s0); s1 = fmaf(a3[j][4 * q + 1], wa.w, s1);
            s0 = fmaf(a3[j][4 * q + 2], wb.x, s0); s1 = fmaf(a3[j][4 * q + 2], wb.y, s1);
            s0 = fmaf(a3[j][4 * q + 3], wb.z, s0); s1 = fmaf(a3[j][4 * q + 3], wb.w, s1);
        }
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    if (h == 0) *reinterpret_cast<float2 *>(p.logits + (row0 + pt) * 2) = make_float2(s0 + p.b4[0], s1 + p.b4[1]);
}

// w (cin,cout) row-major -> 4 KiB chunks for the register chains, one per
// (32-row input tile j, 32-column output tile t):
//   element (q, lane, i) of a chunk = w[32j + 8q + 4(lane>>5) + i][32t + (lane&31)]
// chunk (t, j) is stored at index t*(cin/32) + j, or, j-major, at j*(cout/32) + t.
__global__ void tile_weights_reg_kernel(int cin, int cout, int jmajor, const float *__restrict__ w, float *__restrict__ wt)
{
    const long total = (long)cin * cout;
    const int nj = cin / 32, nt = cout / 32;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int i = e & 3, l = (e >> 2) & 63, q = (e >> 8) & 3;
        const long chunk = e >> 10;
        const int t = jmajor ? (int)(chunk % nt) : (int)(chunk / nj);
        const int j = jmajor ? (int)(chunk / nt) : (int)(chunk % nj);
        const int k = 32 * j + 8 * q + 4 * (l >> 5) + i, col = 32 * t + (l & 31);
        wt[e] = w[(long)k * cout + col];
    }
}

// Flipout's effective weights, one copy per frustum, written in the layout a chain
// kernel reads: out[f][e] = wm[k][c] + dw[k][c] * sin[f][k] * sout[f][c] with (k, c) the
// matrix element that lives at position e of `layout` (0 row-major, 1 / 2 the chunked
// layout of tile_weights_reg_kernel, output-tile / input-tile major).
// Flipout's sign vectors from random bits: sign i = +1 where bit (i % 32) of word i / 32 is set, else -1; a
// thread writes four signs.  (Drawn as floats by the framework they took three launches over every sign - random
// integers, times two, minus one; the bits are a 32nd of that.)
__global__ void flipout_signs_kernel(long n, const unsigned *__restrict__ bits, float *__restrict__ signs)
{
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const unsigned w = bits[i >> 5] >> (i & 31);
    const float4 v = make_float4(w & 1u ? 1.f : -1.f, w & 2u ? 1.f : -1.f, w & 4u ? 1.f : -1.f, w & 8u ? 1.f : -1.f);
    if (i + 3 < n) {
        *reinterpret_cast<float4 *>(signs + i) = v;
    } else {
        signs[i] = v.x;
        if (i + 1 < n) signs[i + 1] = v.y;
        if (i + 2 < n) signs[i + 2] = v.z;
    }
}

// blockIdx.y = a group of kFlipCopies consecutive frustum copies: the mean and the
// perturbation are read once per group (copies of one Monte-Carlo sample are neighbours)
// and only the signs differ, so the kernel is write-bound instead of re-reading both
// matrices through L2 for every copy.
constexpr int kFlipCopies = 8;

// BITS (chunked layouts): the signs as one bit each (set = +1) in the words of sb.bits - copy f's input sign k at
// bit in0 + f * ldi + k (in0, ldi multiples of 4: a thread's four lie in one word), its output sign at bit
// out0 + f * ldo + col.  As floats the signs of a pass are read from memory between the kernel's writes; as bits
// they stay in the L2.
struct SignBitsF32 {
    const unsigned *bits;
    long in0, out0;
};

template <bool BITS>
__global__ void flipout_weights_kernel(int b, int cin, int cout, int layout, const float *__restrict__ wm,
                                       const float *__restrict__ dw, const float *__restrict__ sin, int ldi,
                                       const float *__restrict__ sout, int ldo, float *__restrict__ out,
                                       int per_delta, long delta_stride, SignBitsF32 sb)
{
    const long total = (long)cin * cout;
    const long f0 = (long)blockIdx.y * kFlipCopies;
    const int nf = (int)min((long)kFlipCopies, b - f0);
    if (layout == 0) {
        for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < (unsigned)total; e += gridDim.x * blockDim.x) {
            const unsigned k = e / (unsigned)cout, col = e % (unsigned)cout;
            const float m = wm[e];
            long sample = -1;
            float d = 0.f;
            for (int c = 0; c < nf; ++c) {
                const long f = f0 + c;
                if (f / per_delta != sample) {
                    sample = f / per_delta;
                    d = dw[sample * delta_stride + e];
                }
                out[f * total + e] = m + d * (sin[f * ldi + k] * sout[f * ldo + col]);
            }
        }
        return;
    }
    // chunked layouts: a thread makes one 16-byte group = rows k .. k+3 of one column
    const unsigned nj = cin / 32, nt = cout / 32;
    for (unsigned g = blockIdx.x * blockDim.x + threadIdx.x; g < (unsigned)(total / 4); g += gridDim.x * blockDim.x) {
        const unsigned l = g & 63, q = (g >> 6) & 3, chunk = g >> 8;
        const unsigned t = layout == 2 ? chunk % nt : chunk / nj;
        const unsigned j = layout == 2 ? chunk / nt : chunk % nj;
        const unsigned k = 32 * j + 8 * q + 4 * (l >> 5), col = 32 * t + (l & 31);
        const size_t src = (size_t)k * cout + col;
        const float m0 = wm[src], m1 = wm[src + cout], m2 = wm[src + 2 * (size_t)cout], m3 = wm[src + 3 * (size_t)cout];
        long sample = -1;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
        for (int c = 0; c < nf; ++c) {
            const long f = f0 + c;
            if (f / per_delta != sample) {
                sample = f / per_delta;
                const float *d = dw + sample * delta_stride + src;
                d0 = d[0]; d1 = d[cout]; d2 = d[2 * (size_t)cout]; d3 = d[3 * (size_t)cout];
            }
            float so;
            float4 si;
            if constexpr (BITS) {
                const long ia = sb.in0 + f * ldi + k, io = sb.out0 + f * ldo + col;
                const unsigned wa = sb.bits[ia >> 5] >> (ia & 31), wo = sb.bits[io >> 5] >> (io & 31);
                si = make_float4(wa & 1u ? 1.f : -1.f, wa & 2u ? 1.f : -1.f, wa & 4u ? 1.f : -1.f, wa & 8u ? 1.f : -1.f);
                so = wo & 1u ? 1.f : -1.f;
            } else {
                so = sout[f * ldo + col];
                si = *reinterpret_cast<const float4 *>(sin + f * ldi + k);
            }
            float4 v;
            v.x = m0 + d0 * (si.x * so);
            v.y = m1 + d1 * (si.y * so);
            v.z = m2 + d2 * (si.z * so);
            v.w = m3 + d3 * (si.w * so);
            reinterpret_cast<float4 *>(out + f * total)[g] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// The per-point head with the interpolated features' share of its first layer taken out:
// three_interpolate is linear (weights that sum to one over three known points), and so is
// the first layer before its batch norm and ReLU, so the 128 known points of a frustum go
// through their rows of W1 ONCE and the products are interpolated to the 2048 query points
// (pre), instead of interpolating the features and multiplying 2048 rows.  The kernel starts
// the first layer from pre, adds the point's own (at most 8) channels, and runs the two
// remaining 128 x 128 layers and the final 128 -> 2 product.  Persistent: both layers'
// chunks (128 KiB) and all constants lie in LDS, a wave walks consecutive groups of 32
// points with the next group's rows in flight.
// ---------------------------------------------------------------------------
struct FpHeadPreParams {
    long rows;
    int cb;
    const float *pre, *xb;                      // (rows,128) first-layer products, (rows,cb)
    const float *w1b, *sc1, *sh1;               // (cb,128) row-major: the own channels' rows of W1
    const float4 *w2; const float *sc2, *sh2;   // chunked (128,128)
    const float4 *w3; const float *sc3, *sh3;
    const float *w4, *b4;                       // (128,2) row-major, (2)
    float *logits;                              // (rows,2)
};

__global__ __launch_bounds__(64 * kRegWaves, 1) void fp_head_pre_kernel(const FpHeadPreParams p, unsigned rounds,
                                                                        unsigned rounds_per_wave)
{
    constexpr int W = 4 * 4 * 256;              // float4 per 128 x 128 layer
    extern __shared__ float4 fp_head_pre_lds[];
    float4 *lw2 = fp_head_pre_lds, *lw3 = lw2 + W;
    float *lc = reinterpret_cast<float *>(lw3 + W);     // sc1, sh1, sc2, sh2, sc3, sh3 (128 each), w4 (256)
    const unsigned lane = threadIdx.x & 63;
    const int pt = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < W; i += 64 * kRegWaves) {
        lw2[i] = p.w2[i];
        lw3[i] = p.w3[i];
    }
    for (int i = threadIdx.x; i < 128; i += 64 * kRegWaves) {
        lc[i] = p.sc1[i];       lc[128 + i] = p.sh1[i];
        lc[256 + i] = p.sc2[i]; lc[384 + i] = p.sh2[i];
        lc[512 + i] = p.sc3[i]; lc[640 + i] = p.sh3[i];
        lc[768 + i] = p.w4[i];  lc[896 + i] = p.w4[128 + i];
    }
    __syncthreads();
    const unsigned gw = __builtin_amdgcn_readfirstlane(blockIdx.x * kRegWaves + (threadIdx.x >> 6));
    const unsigned r0 = gw * rounds_per_wave, r1 = min(rounds, r0 + rounds_per_wave);
    if (r0 >= r1) return;
    const float b40 = p.b4[0], b41 = p.b4[1];
    float wv[4][4];             // the own channels' weights: tile t, MFMA step s = channel 2 s + h
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {       // (unconditional loads, as in layer_first)
            const float wk = p.w1b[(2 * s + h < p.cb ? 2 * s + h : 0) * 128 + 32 * t + (lane & 31)];
            wv[t][s] = 2 * s + h < p.cb ? wk : 0.f;
        }

    float4 pr[4][4];
    float xr[4];
    auto load_in = [&](unsigned r) {
        const size_t row = (size_t)r * 32 + pt;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) pr[t][i] = *reinterpret_cast<const float4 *>(p.pre + row * 128 + 32 * t + 8 * i + 4 * h);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float xk = p.xb[row * p.cb + (2 * s + h < p.cb ? 2 * s + h : 0)];
            xr[s] = 2 * s + h < p.cb ? xk : 0.f;
        }
    };
    load_in(r0);
    WChunk ring[4];
#pragma unroll 1
    for (unsigned r = r0; r < r1; ++r) {
        floatx16 a1[4], a2[4], a3[4];
        float x[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) x[s] = xr[s];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a1[t][4 * i + 0] = pr[t][i].x; a1[t][4 * i + 1] = pr[t][i].y;
                a1[t][4 * i + 2] = pr[t][i].z; a1[t][4 * i + 3] = pr[t][i].w;
            }
        load_in(min(r + 1, r1 - 1));
        load_chunk(ring[0], lw2, lane);
        load_chunk(ring[1], lw2 + 4 * 256, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) a1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[t][s], x[s], a1[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            TileConst csc, csh;
            load_const(csc, lc, 32 * t + 4 * h);
            load_const(csh, lc + 128, 32 * t + 4 * h);
            bn_relu(a1[t], csc, csh);
        }
        __builtin_amdgcn_sched_barrier(0);
        layer_t<4, 4>(a2, a1, lw2, lc + 256, lc + 384, ring, lw3, lw3 + 4 * 256, lane, h);
        layer_t<4, 4>(a3, a2, lw3, lc + 512, lc + 640, ring, lw2, lw2 + 4 * 256, lane, h);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 *w = reinterpret_cast<const float4 *>(lc + 768 + (32 * j + 8 * q + 4 * h) * 2);
                const float4 wa = w[0], wb = w[1];
                s0 = fmaf(a3[j][4 * q + 0], wa.x, s0); s1 = fmaf(a3[j][4 * q + 0], wa.y, s1);
                s0 = fmaf(a3[j][4 * q + 1], wa.z, s0); s1 = fmaf(a3[j][4 * q + 1], wa.w, s1);
                s0 = fmaf(a3[j][4 * q + 2], wb.x, s0); s1 = fmaf(a3[j][4 * q + 2], wb.y, s1);
                s0 = fmaf(a3[j][4 * q + 3], wb.z, s0); s1 = fmaf(a3[j][4 * q + 3], wb.w, s1);
            }
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (h == 0) *reinterpret_cast<float2 *>(p.logits + ((size_t)r * 32 + pt) * 2) = make_float2(s0 + b40, s1 + b41);
    }
}

// FPN_F32_SA=launch keeps the per-group kernel for the stacks over coordinates (a measurement switch; the
// twin of the tests)
inline bool sa_persistent_f32()
{
    const char *e = std::getenv("FPN_F32_SA");
    return !(e && std::string(e) == "launch");
}

template <int NT1, int NT2, int NT3, bool W3LDS = true>
int launch_sa_persist(const SaChainParams &p, hipStream_t st)
{
    constexpr size_t lds = (size_t)(NT1 * NT2 + (W3LDS ? NT2 * NT3 : 0)) * 256 * 16 + (2 * 32 * (NT1 + NT2 + NT3)) * 4 +
                           kRegWaves * 32 * NT3 * 4;
    constexpr int per_cu = W3LDS ? 2 : 1;       // workgroups per CU (the kernel's launch bounds)
    static_assert(lds <= 160 * 1024 / per_cu, "the workgroups of a CU must fit LDS");
    const int cus = current_device_cus();
    if (cus < 1)
        return fail(FPN_E_LAUNCH, "sa_chain: cannot read the device's CU count");
    const unsigned balls = (unsigned)(p.rows / p.ns);
    unsigned wgs = std::min<unsigned>((unsigned)cus * per_cu, (balls + kRegWaves - 1) / kRegWaves);
    if (wgs >= 8) wgs -= wgs % 8;               // whole rounds over the XCDs (logical_block)
    const unsigned per_wave = (balls + wgs * kRegWaves - 1) / (wgs * kRegWaves);
    const void *fn = p.pre ? reinterpret_cast<const void *>(&sa_persist_kernel<NT1, NT2, NT3, true, W3LDS>)
                           : reinterpret_cast<const void *>(&sa_persist_kernel<NT1, NT2, NT3, false, W3LDS>);
    if (!allow_dynamic_lds(fn, lds))
        return fail(FPN_E_LAUNCH, "sa_chain: the device refuses %zu bytes of LDS per workgroup", lds);
    if (p.pre)
        hipLaunchKernelGGL((sa_persist_kernel<NT1, NT2, NT3, true, W3LDS>), dim3(wgs), dim3(64 * kRegWaves), lds, st, p, balls,
                           per_wave);
    else
        hipLaunchKernelGGL((sa_persist_kernel<NT1, NT2, NT3, false, W3LDS>), dim3(wgs), dim3(64 * kRegWaves), lds, st, p, balls,
                           per_wave);
    return check_launch("fpn_sa_chain_f32");
}

template <int NJF, int NT1, int NT2, int NT3>
int launch_sa_chain(const SaChainParams &p, hipStream_t st)
{
    const long waves = p.rows / 32;
    hipLaunchKernelGGL((sa_chain_kernel<NJF, NT1, NT2, NT3>), dim3((unsigned)((waves + kRegWaves - 1) / kRegWaves)),
                       dim3(64 * kRegWaves), 0, st, p);
    return check_launch("fpn_sa_chain_f32");
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" {

int fpn_tile_weights_reg_f32(int cin, int cout, int jmajor, const float *w, float *w_tiled, void *stream)
{
    FPN_REQUIRE(cin >= 32 && cin % 32 == 0 && cout >= 32 && cout % 32 == 0,
                "tile_weights_reg: cin (%d) and cout (%d) must be multiples of 32", cin, cout);
    FPN_REQUIRE(w && w_tiled && w != w_tiled, "tile_weights_reg: null or aliased pointer");
    const long total = (long)cin * cout;
    hipLaunchKernelGGL(tile_weights_reg_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
                       as_stream(stream), cin, cout, jmajor, w, w_tiled);
    return check_launch("fpn_tile_weights_reg_f32");
}

int fpn_flipout_signs(long n, const void *bits, float *signs, void *stream)
{
    FPN_REQUIRE(n >= 0 && n < (1L << 40), "flipout_signs: bad size");
    if (n == 0) return FPN_OK;
    FPN_REQUIRE(bits && signs, "flipout_signs: null pointer");
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(bits) & 3) == 0 && (reinterpret_cast<uintptr_t>(signs) & 15) == 0,
                "flipout_signs: bits must be 4-byte aligned, signs 16-byte aligned");
    const long threads = (n + 3) / 4;
    hipLaunchKernelGGL(flipout_signs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, as_stream(stream), n,
                       static_cast<const unsigned *>(bits), signs);
    return check_launch("fpn_flipout_signs");
}

int fpn_flipout_weights_f32(int b, int cin, int cout, const float *w_mean, const float *w_delta,
                            int frustums_per_delta, long delta_stride, const float *sign_in, int ld_in,
                            const float *sign_out, int ld_out, int layout, float *out, void *stream)
{
    FPN_REQUIRE(b >= 0 && cin >= 1 && cout >= 1, "flipout_weights: bad size");
    FPN_REQUIRE(frustums_per_delta >= 1 && delta_stride >= 0, "flipout_weights: bad delta grouping %d / stride %ld",
                frustums_per_delta, delta_stride);
    FPN_REQUIRE(layout >= 0 && layout <= 2, "flipout_weights: layout %d (0 row-major, 1 chunked, 2 chunked input-tile major)", layout);
    FPN_REQUIRE(layout == 0 || (cin % 32 == 0 && cout % 32 == 0),
                "flipout_weights: the chunked layouts need cin (%d) and cout (%d) to be multiples of 32", cin, cout);
    FPN_REQUIRE(ld_in >= cin && ld_out >= cout, "flipout_weights: leading dimensions %d, %d too small", ld_in, ld_out);
    FPN_REQUIRE(layout == 0 || (ld_in % 4 == 0 && (reinterpret_cast<uintptr_t>(sign_in) & 15) == 0 &&
                                (reinterpret_cast<uintptr_t>(out) & 15) == 0),
                "flipout_weights: the chunked layouts read sign_in and write out in 16-byte groups (ld_in %d)", ld_in);
    FPN_REQUIRE((long)cin * cout < (1L << 31), "flipout_weights: layer too large");
    if (b == 0) return FPN_OK;
    FPN_REQUIRE(b <= 65535 * kFlipCopies, "flipout_weights: at most %d frustums per call (%d)", 65535 * kFlipCopies, b);
    FPN_REQUIRE(w_mean && w_delta && sign_in && sign_out && out, "flipout_weights: null pointer");
    const long total = (long)cin * cout;
    hipLaunchKernelGGL(flipout_weights_kernel<false>,
                       dim3((unsigned)std::min<long>((total + 255) / 256, 1024), (unsigned)((b + kFlipCopies - 1) / kFlipCopies)),
                       dim3(256), 0, as_stream(stream), b, cin, cout, layout, w_mean, w_delta, sign_in, ld_in, sign_out, ld_out,
                       out, frustums_per_delta, delta_stride, SignBitsF32{});
    return check_launch("fpn_flipout_weights_f32");
}

int fpn_flipout_weights_bits_f32(int b, int cin, int cout, const float *w_mean, const float *w_delta,
                                 int frustums_per_delta, long delta_stride, const void *bits, long in_bit0, int ld_in,
                                 long out_bit0, int ld_out, int layout, float *out, void *stream)
{
    FPN_REQUIRE(b >= 0 && cin >= 1 && cout >= 1, "flipout_weights_bits: bad size");
    FPN_REQUIRE(frustums_per_delta >= 1 && delta_stride >= 0, "flipout_weights_bits: bad delta grouping %d / stride %ld",
                frustums_per_delta, delta_stride);
    FPN_REQUIRE(layout == 1 || layout == 2, "flipout_weights_bits: layout %d (1 chunked, 2 chunked input-tile major)", layout);
    FPN_REQUIRE(cin % 32 == 0 && cout % 32 == 0,
                "flipout_weights_bits: the chunked layouts need cin (%d) and cout (%d) to be multiples of 32", cin, cout);
    FPN_REQUIRE(ld_in >= cin && ld_out >= cout && in_bit0 >= 0 && out_bit0 >= 0, "flipout_weights_bits: bad sign position or row length");
    FPN_REQUIRE(in_bit0 % 4 == 0 && ld_in % 4 == 0,
                "flipout_weights_bits: a copy's input signs must start at a multiple of four bits (bit %ld, %d a copy)", in_bit0, ld_in);
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(bits) & 3) == 0,
                "flipout_weights_bits: out must be 16-byte aligned, bits 4-byte aligned");
    FPN_REQUIRE((long)cin * cout < (1L << 31), "flipout_weights_bits: layer too large");
    if (b == 0) return FPN_OK;
    FPN_REQUIRE(b <= 65535 * kFlipCopies, "flipout_weights_bits: at most %d frustums per call (%d)", 65535 * kFlipCopies, b);
    FPN_REQUIRE(w_mean && w_delta && bits && out, "flipout_weights_bits: null pointer");
    const long total = (long)cin * cout;
    hipLaunchKernelGGL(flipout_weights_kernel<true>,
                       dim3((unsigned)std::min<long>((total + 255) / 256, 1024), (unsigned)((b + kFlipCopies - 1) / kFlipCopies)),
                       dim3(256), 0, as_stream(stream), b, cin, cout, layout, w_mean, w_delta, nullptr, ld_in, nullptr, ld_out,
                       out, frustums_per_delta, delta_stride, SignBitsF32{static_cast<const unsigned *>(bits), in_bit0, out_bit0});
    return check_launch("fpn_flipout_weights_bits_f32");
}

int fpn_v1_seg_tail_f32(long rows, int n, const float *point_feat, const float *group_shift,
                        const float *const *layers, int per_frustum, const void *drop_mask, float *logits,
                        void *stream)
{
    FPN_REQUIRE(rows >= 0 && n >= 1, "v1_seg_tail: bad size");
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(drop_mask) & 15) == 0, "v1_seg_tail: drop_mask must be 16-byte aligned");
    FPN_REQUIRE(n % 32 == 0 && rows % n == 0, "v1_seg_tail: points per frustum (%d) must be a multiple of 32", n);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(point_feat && group_shift && layers && logits, "v1_seg_tail: null pointer");
    for (int i = 0; i < 13; ++i) FPN_REQUIRE(layers[i], "v1_seg_tail: layer parameter %d is null", i);
    SegTailRegParams p{};
    p.rows = rows; p.n = n; p.point_feat = point_feat; p.gshift = group_shift;
    p.w6 = reinterpret_cast<const float4 *>(layers[0]); p.sc6 = layers[1];
    p.w7 = reinterpret_cast<const float4 *>(layers[2]); p.sc7 = layers[3]; p.sh7 = layers[4];
    p.w8 = reinterpret_cast<const float4 *>(layers[5]); p.sc8 = layers[6]; p.sh8 = layers[7];
    p.w9 = reinterpret_cast<const float4 *>(layers[8]); p.sc9 = layers[9]; p.sh9 = layers[10];
    p.w10 = layers[11]; p.b10 = layers[12];
    p.logits = logits; p.per_frustum = per_frustum != 0;
    p.drop = static_cast<const unsigned *>(drop_mask);
    const long waves = rows / 32;
    hipLaunchKernelGGL(seg_tail_reg_kernel, dim3((unsigned)((waves + kRegWaves - 1) / kRegWaves)), dim3(64 * kRegWaves), 0,
                       as_stream(stream), p);
    return check_launch("fpn_v1_seg_tail_f32");
}

int fpn_sa_chain_supported(int cf, int c1, int c2, int c3)
{
    if (cf <= 5 && c1 == 32 && c2 == 32 && c3 == 64) return 1;      // the narrow stack at its own widths
    const int njf = cf >= 32 ? cf / 32 : 0, t1 = c1 / 32, t2 = c2 / 32, t3 = c3 / 32;
    if (cf >= 32 ? cf % 32 != 0 : cf > 5) return 0;
    if (c1 % 64 || c2 % 32 || c3 % 128) return 0;
    if (c2 % 64) return njf == 0 && t1 == 2 && t2 == 3 && t3 == 4;      // a 96-wide middle layer, unpadded
    return (njf == 0 && t1 == 2 && t2 == 2 && t3 == 4) || (njf == 0 && t1 == 2 && t2 == 4 && t3 == 4) ||
           (njf == 0 && t1 == 4 && t2 == 4 && t3 == 8) ||
           (njf == 10 && t1 == 2 && t2 == 2 && t3 == 4) || (njf == 10 && t1 == 4 && t2 == 4 && t3 == 8) ||
           (njf == 4 && t1 == 4 && t2 == 4 && t3 == 8);
}

int fpn_sa_chain_overwrites(long rows, int cf, int c1, int c2, int c3)
{
    if (!fpn_sa_chain_supported(cf, c1, c2, c3) || cf >= 32 || c1 == 32) return 0;
    const int t1 = c1 / 32, t2 = c2 / 32;
    // (the dispatch of sa_chain_f32 below: the persistent kernels store a ball's pooled row, the others merge maxima)
    return ((t1 == 2 && t2 <= 3) || t1 == 4) && sa_persistent_f32() && rows < (1L << 31);
}

static int sa_chain_f32(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                        const float *new_xyz, const float *feat, const float *pre, int ld_pre, const int *idx,
                        const float *const *layers, float *out, int ldo, int c3_valid, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 1 && m >= 1 && ns >= 1 && cf >= 0, "sa_chain: bad size");
    FPN_REQUIRE(!pre || (cf < 32 && (reinterpret_cast<uintptr_t>(pre) & 15) == 0 && ld_pre >= c1 && ld_pre % 4 == 0),
                "sa_chain: first-layer products go with fewer than 32 direct feature channels, a 16-byte aligned array and "
                "a row stride (%d) of at least c1 floats, a multiple of 4", ld_pre);
    FPN_REQUIRE(ns % 32 == 0, "sa_chain: samples per ball (%d) must be a multiple of 32", ns);
    // the last layer works on pairs of 32-channel tiles: a valid width ends on a pair
    FPN_REQUIRE(c3_valid >= 64 && c3_valid % 64 == 0 && c3_valid <= c3 && ldo >= c3_valid,
                "sa_chain: bad output width %d (a multiple of 64; c3 %d, ld %d)", c3_valid, c3, ldo);
    if (!fpn_sa_chain_supported(cf, c1, c2, c3))
        return fail(FPN_E_UNSUPPORTED, "sa_chain: no kernel for %d feature channels and layer widths %d,%d,%d", cf, c1, c2, c3);
    if (b == 0) return FPN_OK;
    FPN_REQUIRE(xyz && new_xyz && idx && layers && out && (cf == 0 || feat), "sa_chain: null pointer");
    for (int i = 0; i < 9; ++i) FPN_REQUIRE(layers[i], "sa_chain: layer parameter %d is null", i);
    SaChainParams p{};
    p.rows = (long)b * m * ns; p.n = n; p.m = m; p.ns = ns; p.cf = cf;
    p.xyz = xyz; p.new_xyz = new_xyz; p.feat = feat; p.pre = pre; p.ldp = ld_pre; p.idx = idx;
    p.w1 = layers[0]; p.sc1 = layers[1]; p.sh1 = layers[2];
    p.w2 = reinterpret_cast<const float4 *>(layers[3]); p.sc2 = layers[4]; p.sh2 = layers[5];
    p.w3 = reinterpret_cast<const float4 *>(layers[6]); p.sc3 = layers[7]; p.sh3 = layers[8];
    p.out = out; p.ldo = ldo; p.c3 = c3_valid;
    const int njf = cf >= 32 ? cf / 32 : 0, t1 = c1 / 32, t2 = c2 / 32;
    hipStream_t st = as_stream(stream);
    if (c1 == 32) {
        FPN_REQUIRE(!pre, "sa_chain: the 32-32-64 stack takes no first-layer products");
        FPN_REQUIRE(p.rows / 32 < (1L << 31), "sa_chain: too many rows");
        const int cus = current_device_cus();
        if (cus < 1)
            return fail(FPN_E_LAUNCH, "sa_chain: cannot read the device's CU count");
        const unsigned groups = (unsigned)(p.rows / 32);
        unsigned wgs = std::min<unsigned>((unsigned)cus * 2, (groups + kRegWaves - 1) / kRegWaves);
        if (wgs >= 8) wgs -= wgs % 8;           // whole rounds over the XCDs (logical_block)
        const unsigned per_wave = (groups + wgs * kRegWaves - 1) / (wgs * kRegWaves);
        hipLaunchKernelGGL(sa_small_kernel, dim3(wgs), dim3(64 * kRegWaves), 0, st, p, groups, per_wave);
        return check_launch("fpn_sa_chain_f32");
    }
    if (njf == 0 && t1 == 2 && t2 <= 3 && sa_persistent_f32() && p.rows < (1L << 31))
        return t2 == 2 ? launch_sa_persist<2, 2, 4>(p, st) : launch_sa_persist<2, 3, 4>(p, st);
    if (njf == 0 && t1 == 4 && sa_persistent_f32() && p.rows < (1L << 31)) return launch_sa_persist<4, 4, 8, false>(p, st);
    if (njf == 0 && t1 == 4) return launch_sa_chain<0, 4, 4, 8>(p, st);
    if (njf == 0 && t2 == 2) return launch_sa_chain<0, 2, 2, 4>(p, st);
    if (njf == 0 && t2 == 3) return launch_sa_chain<0, 2, 3, 4>(p, st);
    if (njf == 0) return launch_sa_chain<0, 2, 4, 4>(p, st);
    if (njf == 10 && t1 == 2) return launch_sa_chain<10, 2, 2, 4>(p, st);
    if (njf == 10) return launch_sa_chain<10, 4, 4, 8>(p, st);
    return launch_sa_chain<4, 4, 4, 8>(p, st);
}

int fpn_sa_chain_f32(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                     const float *new_xyz, const float *feat, const int *idx, const float *const *layers,
                     float *out, int ldo, int c3_valid, void *stream)
{
    return sa_chain_f32(b, n, m, ns, cf, c1, c2, c3, xyz, new_xyz, feat, nullptr, 0, idx, layers, out, ldo, c3_valid, stream);
}

int fpn_sa_chain_pre_f32(int b, int n, int m, int ns, int cf, int c1, int c2, int c3, const float *xyz,
                         const float *new_xyz, const float *feat, const float *pre, int ld_pre, const int *idx,
                         const float *const *layers, float *out, int ldo, int c3_valid, void *stream)
{
    FPN_REQUIRE(pre || b == 0, "sa_chain_pre: null pointer");
    return sa_chain_f32(b, n, m, ns, cf, c1, c2, c3, xyz, new_xyz, feat, pre, ld_pre, idx, layers, out, ldo, c3_valid, stream);
}

int fpn_fp_head_chain_f32(long rows, int ca, int cb, const float *xa, const float *xb,
                          const float *const *layers, float *logits, void *stream)
{
    FPN_REQUIRE(rows >= 0 && rows % 32 == 0, "fp_head_chain: rows (%ld) must be a multiple of 32", rows);
    FPN_REQUIRE(ca == 128, "fp_head_chain: %d interpolated channels, a kernel exists for 128", ca);
    FPN_REQUIRE(cb >= 0 && cb <= 32, "fp_head_chain: at most 32 own channels per point (%d)", cb);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(xa && (cb == 0 || xb) && layers && logits, "fp_head_chain: null pointer");
    for (int i = 0; i < 11; ++i) FPN_REQUIRE(layers[i], "fp_head_chain: layer parameter %d is null", i);
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(xa) & 15) == 0, "fp_head_chain: xa must be 16-byte aligned");
    FpHeadParams p{};
    p.rows = rows; p.cb = cb; p.xa = xa; p.xb = cb ? xb : xa;
    p.w1 = reinterpret_cast<const float4 *>(layers[0]); p.sc1 = layers[1]; p.sh1 = layers[2];
    p.w2 = reinterpret_cast<const float4 *>(layers[3]); p.sc2 = layers[4]; p.sh2 = layers[5];
    p.w3 = reinterpret_cast<const float4 *>(layers[6]); p.sc3 = layers[7]; p.sh3 = layers[8];
    p.w4 = layers[9]; p.b4 = layers[10];
    p.logits = logits;
    const long waves = rows / 32;
    hipLaunchKernelGGL(fp_head_reg_kernel<4>, dim3((unsigned)((waves + kRegWaves - 1) / kRegWaves)), dim3(64 * kRegWaves), 0,
                       as_stream(stream), p);
    return check_launch("fpn_fp_head_chain_f32");
}

int fpn_fp_head_chain_pre_f32(long rows, int cb, const float *pre, const float *xb, const float *const *layers,
                              float *logits, void *stream)
{
    FPN_REQUIRE(rows >= 0 && rows % 32 == 0 && rows / 32 < (1L << 31), "fp_head_chain_pre: rows (%ld) must be a multiple of 32", rows);
    FPN_REQUIRE(cb >= 0 && cb <= 8, "fp_head_chain_pre: at most 8 own channels per point (%d)", cb);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(pre && (cb == 0 || xb) && layers && logits, "fp_head_chain_pre: null pointer");
    for (int i = (cb ? 0 : 1); i < 11; ++i) FPN_REQUIRE(layers[i], "fp_head_chain_pre: layer parameter %d is null", i);
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(pre) & 15) == 0, "fp_head_chain_pre: pre must be 16-byte aligned");
    FpHeadPreParams p{};
    p.rows = rows; p.cb = cb; p.pre = pre; p.xb = cb ? xb : pre;
    p.w1b = cb ? layers[0] : layers[1]; p.sc1 = layers[1]; p.sh1 = layers[2];
    p.w2 = reinterpret_cast<const float4 *>(layers[3]); p.sc2 = layers[4]; p.sh2 = layers[5];
    p.w3 = reinterpret_cast<const float4 *>(layers[6]); p.sc3 = layers[7]; p.sh3 = layers[8];
    p.w4 = layers[9]; p.b4 = layers[10];
    p.logits = logits;
    constexpr size_t lds = (size_t)2 * 4 * 4 * 256 * 16 + 1024 * 4;
    const int cus = current_device_cus();
    if (cus < 1)
        return fail(FPN_E_LAUNCH, "fp_head_chain_pre: cannot read the device's CU count");
    const unsigned rounds = (unsigned)(rows / 32);
    const unsigned wgs = std::min<unsigned>((unsigned)cus, (rounds + kRegWaves - 1) / kRegWaves);
    const unsigned per_wave = (rounds + wgs * kRegWaves - 1) / (wgs * kRegWaves);
    if (!allow_dynamic_lds(reinterpret_cast<const void *>(&fp_head_pre_kernel), lds))
        return fail(FPN_E_LAUNCH, "fp_head_chain_pre: the device refuses %zu bytes of LDS per workgroup", lds);
    hipLaunchKernelGGL(fp_head_pre_kernel, dim3(wgs), dim3(64 * kRegWaves), lds, as_stream(stream), p, rounds, per_wave);
    return check_launch("fpn_fp_head_chain_pre_f32");
}

int fpn_v1_seg_head_f32(long rows, int n, int cpc, const float *pc, const float *const *layers,
                        int per_frustum, float *point_feat, float *global_max, int ldg, void *stream)
{
    FPN_REQUIRE(rows >= 0 && n >= 1 && cpc >= 3 && cpc <= 8, "v1_seg_head: bad size rows=%ld n=%d cpc=%d", rows, n, cpc);
    FPN_REQUIRE(n % 32 == 0 && rows % n == 0, "v1_seg_head: points per frustum (%d) must be a multiple of 32", n);
    FPN_REQUIRE(ldg >= 1024, "v1_seg_head: leading dimension of global_max (%d) must be >= 1024", ldg);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(pc && layers && point_feat && global_max, "v1_seg_head: null pointer");
    for (int i = 0; i < 15; ++i) FPN_REQUIRE(layers[i], "v1_seg_head: layer parameter %d is null", i);
    SegHeadRegParams p{};
    p.rows = rows; p.n = n; p.cpc = cpc; p.pc = pc;
    p.w1 = layers[0]; p.sc1 = layers[1]; p.sh1 = layers[2];
    p.w2 = reinterpret_cast<const float4 *>(layers[3]); p.sc2 = layers[4]; p.sh2 = layers[5];
    p.w3 = reinterpret_cast<const float4 *>(layers[6]); p.sc3 = layers[7]; p.sh3 = layers[8];
    p.w4 = reinterpret_cast<const float4 *>(layers[9]); p.sc4 = layers[10]; p.sh4 = layers[11];
    p.w5 = reinterpret_cast<const float4 *>(layers[12]); p.sc5 = layers[13]; p.sh5 = layers[14];
    p.point_feat = point_feat; p.gmax = global_max; p.ldg = ldg; p.per_frustum = per_frustum != 0;
    const long waves = rows / 32;
    hipLaunchKernelGGL(seg_head_reg_kernel, dim3((unsigned)((waves + kHeadWaves - 1) / kHeadWaves)), dim3(64 * kHeadWaves), 0,
                       as_stream(stream), p);
    return check_launch("fpn_v1_seg_head_f32");
}

int fpn_v1_obj_chain_f32(long rows, int m, int box, const float *obj, const float *delta,
                         const float *const *layers, int per_frustum, float *feat_max, int ldg, void *stream)
{
    FPN_REQUIRE(rows >= 0 && m >= 1, "v1_obj_chain: bad size");
    FPN_REQUIRE(m % 32 == 0 && rows % m == 0, "v1_obj_chain: object points per frustum (%d) must be a multiple of 32", m);
    FPN_REQUIRE(ldg >= (box ? 512 : 256), "v1_obj_chain: leading dimension of feat_max (%d) too small", ldg);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(obj && layers && feat_max, "v1_obj_chain: null pointer");
    const int nl = box ? 12 : 9;
    for (int i = 0; i < nl; ++i) FPN_REQUIRE(layers[i], "v1_obj_chain: layer parameter %d is null", i);
    ObjChainRegParams p{};
    p.rows = rows; p.m = m; p.obj = obj; p.delta = delta;
    p.w1 = layers[0]; p.sc1 = layers[1]; p.sh1 = layers[2];
    p.w2 = reinterpret_cast<const float4 *>(layers[3]); p.sc2 = layers[4]; p.sh2 = layers[5];
    p.w3 = reinterpret_cast<const float4 *>(layers[6]); p.sc3 = layers[7]; p.sh3 = layers[8];
    if (box) { p.w4 = reinterpret_cast<const float4 *>(layers[9]); p.sc4 = layers[10]; p.sh4 = layers[11]; }
    p.gmax = feat_max; p.ldg = ldg; p.per_frustum = per_frustum != 0;
    const long waves = rows / 32;
    const dim3 grid((unsigned)((waves + kRegWaves - 1) / kRegWaves));
    if (box) hipLaunchKernelGGL(obj_chain_reg_kernel<true>, grid, dim3(64 * kRegWaves), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(obj_chain_reg_kernel<false>, grid, dim3(64 * kRegWaves), 0, as_stream(stream), p);
    return check_launch("fpn_v1_obj_chain_f32");
}

}  // extern "C"
