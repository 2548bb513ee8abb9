// Layer-chain kernels of the v1 networks: a workgroup carries a tile of points
// through SEVERAL consecutive shared-MLP layers with the activations resident in
// LDS, so that only the tile's input and its final output touch HBM.
//
//   seg head  : conv1 -> conv2 -> conv3 (point_feat, also stored) -> conv4 ->
//               conv5 + max over points                 128 points per workgroup
//   obj chain : T-Net / box-net convolution stacks + max  64 object points per workgroup
//
// Matrix work runs on v_mfma_f32_32x32x2_f32 (exact fp32).  One MFMA multiplies, for
// every lane, A[row = lane&31][k(lane>>5)] by B[k(lane>>5)][col = lane&31]: the two
// lane halves supply two different k.  The kernels walk K in blocks of 32 and give
// step s (0..15) of a block the pair k = {s, 16+s}: lane half h then needs, for the
// 16 steps, the 16 CONSECUTIVE k values 16h..16h+15 of both operands, which is what
// makes wide loads possible:
//   A (activations): LDS image, row-major, row stride K+4 words (16-byte aligned rows;
//                    the 16 lanes of a ds_read_b128 group fall on 16 different
//                    4-bank slots), read as 4 x ds_read_b128 per 32-k block;
//   B (weights)    : global memory in a TILED layout prepared once by
//                    fpn_tile_weights_f32 - for each (32-column tile, 32-k block) a
//                    4 KiB chunk ordered [q][lane][4 floats], so that each of the 4
//                    global_load_dwordx4 of a block is one fully coalesced 1 KiB read.
//                    Each wave owns distinct output columns, so weights go straight
//                    to registers (no LDS staging) and are requested one block ahead.
// (Measured on MI355X: with per-k dword weight loads the same loop was limited by
// the global-load path at ~87 % of the MFMA rate; the LDS path alone reaches 98 %.)
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "fpn_common.h"

namespace fpn {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// Optional in-kernel time stamps (diagnostic builds only, -DFPN_CHAIN_STAMPS): lane 0
// of every wave records s_memtime before and after each barrier into a buffer of its
// own; tools/chain_stamps.hip reads it back.  Compiled out of the library.
#ifdef FPN_CHAIN_STAMPS
__device__ unsigned long long g_stamps[1 << 22];
__device__ __forceinline__ void fpn_stamp(int &i)
{
    if ((threadIdx.x & 63) == 0)
        g_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + i] = __builtin_amdgcn_s_memtime();
    ++i;
}
#define FPN_SYNC()            \
    do {                      \
        fpn_stamp(stamp_i);   \
        __syncthreads();      \
        fpn_stamp(stamp_i);   \
    } while (0)
#define FPN_KERNEL_BEGIN \
    int stamp_i = 0;     \
    fpn_stamp(stamp_i);
#define FPN_KERNEL_END fpn_stamp(stamp_i);
#else
#define FPN_SYNC() __syncthreads()
#define FPN_KERNEL_BEGIN
#define FPN_KERNEL_END
#endif

constexpr int kChainThreads = 512;  // 8 waves, 2 per SIMD

// row (0..31) of accumulator register r in a 32x32 tile
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(floatx16 (&acc)[TM][TN])
{
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
}

// One 32-k block of weights for one 32-column tile: 4 coalesced 16-byte loads per lane.
// wt points at the tile's chunk for k-block 0; chunks of consecutive k-blocks follow
// each other (256 float4 each).
struct WBlock {
    float4 q[4];
};

__device__ __forceinline__ void load_wblock(WBlock &b, const float4 *__restrict__ wt, int kb, int lane)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) b.q[i] = wt[kb * 256 + i * 64 + lane];
}

__device__ __forceinline__ float wb_get(const WBlock &b, int s)
{
    const float4 &v = b.q[s >> 2];
    return (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
}

// 16 MFMA steps of one 32-k block.  xs points at &X[first row + (lane&31)][16*(lane>>5)]
// of k-block 0 (LDS, row stride XS words, XS % 4 == 0).
template <int TM>
struct ABlock {
    float4 v[TM][4];
};

template <int TM, int XS>
__device__ __forceinline__ void load_ablock(ABlock<TM> &a, const float *xs, int kb)
{
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            a.v[t][i] = *reinterpret_cast<const float4 *>(xs + t * 32 * XS + kb * 32 + 4 * i);
}

template <int TM>
__device__ __forceinline__ void mfma_regs(floatx16 (&acc)[TM][1], const ABlock<TM> &a, const WBlock &b)
{
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const float4 &v = a.v[t][s >> 2];
            const float x = (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, wb_get(b, s), acc[t][0], 0, 0, 0);
        }
}

template <int TM, int XS>
__device__ __forceinline__ void mfma_block(floatx16 (&acc)[TM][1], const float *xs, int kb, const WBlock &b)
{
    ABlock<TM> a;
    load_ablock<TM, XS>(a, xs, kb);
    mfma_regs<TM>(acc, a, b);
}

// acc[tm][0] += X[tm-th 32-row tile][0:32*NKB] . W[0:32*NKB][one 32-column tile]
// b0 must already hold k-block 0 of the tile (load_wblock(b0, wt, 0, lane)): the
// caller requests it EARLY - before the epilogue and barrier that precede this
// product - so that its latency is covered by work.  b0 is consumed.
//
// FPN_CHAIN_PIPELINE 1: software pipeline at half-block granularity.  While the 8*TM
// MFMAs of one half-block run, the activations of the NEXT half-block (2*TM
// ds_read_b128) and, every other half, the weights of the next block (4
// global_load_dwordx4) are requested into a second register set; the order is pinned
// with sched_group_barrier (one load in the shadow of each MFMA), because left alone
// the scheduler sinks the loads to just before their use.  Loads past the end are
// clamped to the last block (harmless re-reads) so that the loop body is one basic
// block.  FPN_CHAIN_PIPELINE 0: plain form, scheduling left to the compiler.
// Measured on MI355X (1024 frustums x 1024 points): 128.2k frustums/s pipelined,
// 130.7k plain - with two waves per SIMD the other wave already covers the load
// latency, and what is lost is lost at the phase boundaries, not inside this loop.
#ifndef FPN_CHAIN_PIPELINE
#define FPN_CHAIN_PIPELINE 0
#endif

template <int TM>
struct AHalf {
    float4 v[TM][2];
};

// half-block h covers, for lane half hl, k = 32*(h/2) + 16*hl + 8*(h%2) + [0,8)
template <int TM, int XS>
__device__ __forceinline__ void load_ahalf(AHalf<TM> &a, const float *xs, int h)
{
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            a.v[t][i] = *reinterpret_cast<const float4 *>(xs + t * 32 * XS + (h >> 1) * 32 + (h & 1) * 8 + 4 * i);
}

template <int TM, int HSEL>
__device__ __forceinline__ void mfma_half(floatx16 (&acc)[TM][1], const AHalf<TM> &a, const WBlock &b)
{
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const float4 &v = a.v[t][s >> 2];
            const float x = (s & 3) == 0 ? v.x : (s & 3) == 1 ? v.y : (s & 3) == 2 ? v.z : v.w;
            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, wb_get(b, 8 * HSEL + s), acc[t][0], 0, 0, 0);
        }
}

// scheduling groups of one half-block: one load in the shadow of each of the first
// MFMAs, then the rest of the MFMAs
template <int TM, bool VMEM>
__device__ __forceinline__ void sched_half()
{
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    }
    if (VMEM) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
        }
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8 * TM - 2 * TM - (VMEM ? 4 : 0), 0);
}

template <int TM, int NKB, int XS>
__device__ __forceinline__ void wave_gemm(floatx16 (&acc)[TM][1], const float *xs, const float4 *__restrict__ wt,
                                          int lane, WBlock &b0)
{
    static_assert(NKB % 2 == 0, "K must be a multiple of 64");
    WBlock b1;
#if FPN_CHAIN_PIPELINE
    AHalf<TM> ae, ao;
    load_ahalf<TM, XS>(ae, xs, 0);
#pragma unroll 1
    for (int kb = 0; kb < NKB; kb += 2) {
        const int kbn = kb + 2 < NKB ? kb + 2 : kb;
        load_ahalf<TM, XS>(ao, xs, 2 * kb + 1);
        load_wblock(b1, wt, kb + 1, lane);
        mfma_half<TM, 0>(acc, ae, b0);
        load_ahalf<TM, XS>(ae, xs, 2 * kb + 2);
        mfma_half<TM, 1>(acc, ao, b0);
        load_ahalf<TM, XS>(ao, xs, 2 * kb + 3);
        load_wblock(b0, wt, kbn, lane);
        mfma_half<TM, 0>(acc, ae, b1);
        load_ahalf<TM, XS>(ae, xs, 2 * kbn);
        mfma_half<TM, 1>(acc, ao, b1);
        sched_half<TM, true>();
        sched_half<TM, false>();
        sched_half<TM, true>();
        sched_half<TM, false>();
    }
#else
#pragma unroll 1
    for (int kb = 0; kb < NKB; kb += 2) {
        load_wblock(b1, wt, kb + 1, lane);
        mfma_block<TM, XS>(acc, xs, kb, b0);
        if (kb + 2 < NKB) load_wblock(b0, wt, kb + 2, lane);
        mfma_block<TM, XS>(acc, xs, kb + 1, b1);
    }
#endif
}

// start of the tiled weights of column tile ct of a layer with NKB k-blocks
__device__ __forceinline__ const float4 *wtile(const float *wt, int ct, int nkb)
{
    return reinterpret_cast<const float4 *>(wt) + (size_t)ct * nkb * 256;
}

// relu(acc*scale+shift) of one 32x32 tile into an LDS activation image
template <int XS>
__device__ __forceinline__ void store_tile_lds(const floatx16 &acc, float *dst /* &Y[tile row 0][tile col 0] */,
                                               float sc, float sh, int l31, int half)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[acc_row(r, half) * XS + l31] = fmaxf(acc[r] * sc + sh, 0.f);
}

// ---------------------------------------------------------------------------
// seg head
// ---------------------------------------------------------------------------
struct SegHeadParams {
    long rows;
    int n;    // points per frustum (multiple of 128)
    int cpc;  // channels of the point cloud (3 or 4 ... <= 8)
    const float *pc;
    const float *w1, *sc1, *sh1;  // (cpc,64) plain
    const float *w2, *sc2, *sh2;  // (64,64)    tiled
    const float *w3, *sc3, *sh3;  // (64,64)    tiled
    const float *w4, *sc4, *sh4;  // (64,128)   tiled
    const float *w5, *sc5, *sh5;  // (128,1024) tiled
    float *point_feat;            // (rows,64)
    float *partial;               // (rows/128,1024) per-workgroup column maxima
};

constexpr int kHeadRows = 128;

__global__ __launch_bounds__(kChainThreads) void seg_head_kernel(const SegHeadParams p)
{
    __shared__ __attribute__((aligned(16))) float B0[kHeadRows * 68];
    __shared__ __attribute__((aligned(16))) float B1[kHeadRows * 68];
    __shared__ __attribute__((aligned(16))) float B2[kHeadRows * 132];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    FPN_KERNEL_BEGIN
    const int l31 = lane & 31, half = lane >> 5;
    const long row0 = (long)blockIdx.x * kHeadRows;

    // first weight block of each matrix product, requested one phase early
    WBlock wq;
    load_wblock(wq, wtile(p.w2, w >> 2, 2), 0, lane);

    // conv1 (K = cpc <= 8) on the matrix cores too, operands straight from memory:
    // step t multiplies channels {2t, 2t+1}; 4 row tiles x 2 column tiles, one per wave
    {
        const int rt = w & 3, ct = w >> 2;
        const int col = ct * 32 + l31;
        const float sc = p.sc1[col], sh = p.sh1[col];
        const float *src = p.pc + (row0 + rt * 32 + l31) * p.cpc;
        float a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int k = 2 * t + half;
            a[t] = k < p.cpc ? src[k] : 0.f;
            b[t] = k < p.cpc ? p.w1[k * 64 + col] : 0.f;
        }
        floatx16 acc[1][1];
        zero_acc(acc);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (2 * t < p.cpc) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[0][0], 0, 0, 0);
        store_tile_lds<68>(acc[0][0], B0 + rt * 32 * 68 + ct * 32, sc, sh, l31, half);
    }
    FPN_SYNC();

    // conv2: B0 -> B1   (4 row tiles x 2 column tiles, one per wave)
    {
        const int rt = w & 3, ct = w >> 2;
        const int col = ct * 32 + l31;
        const float sc = p.sc2[col], sh = p.sh2[col];
        floatx16 acc[1][1];
        zero_acc(acc);
        wave_gemm<1, 2, 68>(acc, B0 + (rt * 32 + l31) * 68 + 16 * half, wtile(p.w2, ct, 2), lane, wq);
        load_wblock(wq, wtile(p.w3, ct, 2), 0, lane);
        store_tile_lds<68>(acc[0][0], B1 + rt * 32 * 68 + ct * 32, sc, sh, l31, half);
    }
    FPN_SYNC();
    // conv3: B1 -> B0 (= point_feat, also written to memory for the tail kernel)
    {
        const int rt = w & 3, ct = w >> 2;
        const int col = ct * 32 + l31;
        const float sc = p.sc3[col], sh = p.sh3[col];
        floatx16 acc[1][1];
        zero_acc(acc);
        wave_gemm<1, 2, 68>(acc, B1 + (rt * 32 + l31) * 68 + 16 * half, wtile(p.w3, ct, 2), lane, wq);
        load_wblock(wq, wtile(p.w4, w & 3, 2), 0, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + acc_row(r, half);
            const float v = fmaxf(acc[0][0][r] * sc + sh, 0.f);
            B0[row * 68 + col] = v;
            p.point_feat[(row0 + row) * 64 + col] = v;
        }
    }
    FPN_SYNC();
    // conv4: B0 -> B2   (4 row tiles x 4 column tiles, two row tiles per wave)
    {
        const int ct = w & 3, rt0 = (w >> 2) * 2;
        const int col = ct * 32 + l31;
        const float sc = p.sc4[col], sh = p.sh4[col];
        floatx16 acc[2][1];
        zero_acc(acc);
        wave_gemm<2, 2, 68>(acc, B0 + (rt0 * 32 + l31) * 68 + 16 * half, wtile(p.w4, ct, 2), lane, wq);
        load_wblock(wq, wtile(p.w5, w, 4), 0, lane);
#pragma unroll
        for (int a = 0; a < 2; ++a) store_tile_lds<132>(acc[a][0], B2 + (rt0 + a) * 32 * 132 + ct * 32, sc, sh, l31, half);
    }
    FPN_SYNC();
    // conv5 + max over the 128 points: wave w owns column tiles w, w+8, w+16, w+24
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const int ct = w + 8 * i;
        const int col = ct * 32 + l31;
        const float sc = p.sc5[col], sh = p.sh5[col];
        floatx16 acc[4][1];
        zero_acc(acc);
        wave_gemm<4, 4, 132>(acc, B2 + l31 * 132 + 16 * half, wtile(p.w5, ct, 4), lane, wq);
        if (i < 3) load_wblock(wq, wtile(p.w5, ct + 8, 4), 0, lane);
        float m = 0.f;  // ReLU output is >= 0
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[a][0][r] * sc + sh);
        m = fmaxf(m, __shfl_xor(m, 32));
        if (half == 0) p.partial[(long)blockIdx.x * 1024 + col] = m;
    }
    FPN_KERNEL_END
}

// ---------------------------------------------------------------------------
// Object-point chains: the convolution stacks of the T-Net (3 -> 128 -> 128 ->
// 256, max over points) and of the box net (3 -> 128 -> 128 -> 256 -> 512, max
// over points), 64 object points per workgroup.  The box net's input is the
// object cloud minus the T-Net's centre estimate; that subtraction happens while
// the tile is loaded.
// ---------------------------------------------------------------------------
struct ObjChainParams {
    long rows;
    int m;                // object points per frustum (multiple of 64)
    const float *obj;     // (rows,3)
    const float *delta;   // (rows/m,3) subtracted from every point of the frustum, or NULL
    const float *w1, *sc1, *sh1;  // (3,128) plain
    const float *w2, *sc2, *sh2;  // (128,128) tiled
    const float *w3, *sc3, *sh3;  // (128,256) tiled
    const float *w4, *sc4, *sh4;  // (256,512) tiled, box net only
    float *partial;               // (rows/64, 256 or 512) per-workgroup column maxima
};

constexpr int kObjRows = 64;

template <bool BOX>
__global__ __launch_bounds__(kChainThreads) void obj_chain_kernel(const ObjChainParams p)
{
    __shared__ __attribute__((aligned(16))) float A[kObjRows * 132];
    __shared__ __attribute__((aligned(16))) float B[kObjRows * 132];
    __shared__ __attribute__((aligned(16))) float C[BOX ? kObjRows * 260 : 4];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    FPN_KERNEL_BEGIN
    const int l31 = lane & 31, half = lane >> 5;
    const long row0 = (long)blockIdx.x * kObjRows;

    WBlock wq;  // first weight block of the next product, requested one phase early
    load_wblock(wq, wtile(p.w2, w >> 1, 4), 0, lane);

    {   // layer 1 (K = 3, zero-padded to 4) on the matrix cores, operands straight from
        // memory: step t multiplies coordinates {2t, 2t+1}; 2 row tiles x 4 column tiles
        const int rt = w & 1, ct = w >> 1;
        const int col = ct * 32 + l31;
        const float sc = p.sc1[col], sh = p.sh1[col];
        const float *src = p.obj + (row0 + rt * 32 + l31) * 3;
        const float *d = p.delta ? p.delta + (row0 / p.m) * 3 : nullptr;
        float a[2], b[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = 2 * t + half;
            a[t] = k < 3 ? src[k] - (d ? d[k] : 0.f) : 0.f;
            b[t] = k < 3 ? p.w1[k * 128 + col] : 0.f;
        }
        floatx16 acc[1][1];
        zero_acc(acc);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[0][0], 0, 0, 0);
        store_tile_lds<132>(acc[0][0], A + rt * 32 * 132 + ct * 32, sc, sh, l31, half);
    }
    FPN_SYNC();
    {   // layer 2: A -> B, 2 row tiles x 4 column tiles
        const int rt = w & 1, ct = w >> 1;
        const int col = ct * 32 + l31;
        const float sc = p.sc2[col], sh = p.sh2[col];
        floatx16 acc[1][1];
        zero_acc(acc);
        wave_gemm<1, 4, 132>(acc, A + (rt * 32 + l31) * 132 + 16 * half, wtile(p.w2, ct, 4), lane, wq);
        load_wblock(wq, wtile(p.w3, w, 4), 0, lane);
        store_tile_lds<132>(acc[0][0], B + rt * 32 * 132 + ct * 32, sc, sh, l31, half);
    }
    FPN_SYNC();
    {   // layer 3: B -> 256 columns, wave w owns column tile w and both row tiles
        const int col = w * 32 + l31;
        const float sc = p.sc3[col], sh = p.sh3[col];
        floatx16 acc[2][1];
        zero_acc(acc);
        wave_gemm<2, 4, 132>(acc, B + l31 * 132 + 16 * half, wtile(p.w3, w, 4), lane, wq);
        if (BOX) load_wblock(wq, wtile(p.w4, w, 8), 0, lane);
        if (!BOX) {
            float m = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[a][0][r] * sc + sh);
            m = fmaxf(m, __shfl_xor(m, 32));
            if (half == 0) p.partial[(long)blockIdx.x * 256 + col] = m;
            return;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) store_tile_lds<260>(acc[a][0], C + a * 32 * 260 + w * 32, sc, sh, l31, half);
    }
    if (BOX) {
        FPN_SYNC();
        // layer 4: C -> 512 columns + max over the 64 points; column tiles w and w + 8
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            const int ct = w + 8 * i;
            const int col = ct * 32 + l31;
            const float sc = p.sc4[col], sh = p.sh4[col];
            floatx16 acc[2][1];
            zero_acc(acc);
            wave_gemm<2, 8, 260>(acc, C + l31 * 260 + 16 * half, wtile(p.w4, ct, 8), lane, wq);
            if (i == 0) load_wblock(wq, wtile(p.w4, ct + 8, 8), 0, lane);
            float m = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[a][0][r] * sc + sh);
            m = fmaxf(m, __shfl_xor(m, 32));
            if (half == 0) p.partial[(long)blockIdx.x * 512 + col] = m;
        }
    }
}

// w (cin,cout) row-major -> tiled: for column tile ct, k-block kb, load index q,
// lane l, element j:  wt[(((ct*NKB + kb)*4 + q)*64 + l)*4 + j] =
//                     w[kb*32 + 16*(l>>5) + 4*q + j][ct*32 + (l&31)]
__global__ void tile_weights_kernel(int cin, int cout, const float *__restrict__ w, float *__restrict__ wt)
{
    const long total = (long)cin * cout;
    const int nkb = cin / 32;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = e & 3, l = (e >> 2) & 63, q = (e >> 8) & 3;
        const long blk = e >> 10;
        const int kb = (int)(blk % nkb), ct = (int)(blk / nkb);
        const int k = kb * 32 + 16 * (l >> 5) + 4 * q + j, col = ct * 32 + (l & 31);
        wt[e] = w[(long)k * cout + col];
    }
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" {

int fpn_tile_weights_f32(int cin, int cout, const float *w, float *w_tiled, void *stream)
{
    FPN_REQUIRE(cin >= 64 && cin % 64 == 0 && cout >= 32 && cout % 32 == 0,
                "tile_weights: cin (%d) must be a multiple of 64 and cout (%d) a multiple of 32", cin, cout);
    FPN_REQUIRE(w && w_tiled && w != w_tiled, "tile_weights: null or aliased pointer");
    const long total = (long)cin * cout;
    hipLaunchKernelGGL(tile_weights_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
                       as_stream(stream), cin, cout, w, w_tiled);
    return check_launch("fpn_tile_weights_f32");
}

int fpn_v1_seg_head_f32(long rows, int n, int cpc, const float *pc, const float *const *layers,
                        float *point_feat, float *partial_max, void *stream)
{
    FPN_REQUIRE(rows >= 0 && n >= 1 && cpc >= 3 && cpc <= 8, "v1_seg_head: bad size rows=%ld n=%d cpc=%d", rows, n, cpc);
    FPN_REQUIRE(n % kHeadRows == 0 && rows % n == 0, "v1_seg_head: points per frustum (%d) must be a multiple of %d", n, kHeadRows);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(pc && layers && point_feat && partial_max, "v1_seg_head: null pointer");
    for (int i = 0; i < 15; ++i) FPN_REQUIRE(layers[i], "v1_seg_head: layer parameter %d is null", i);
    SegHeadParams p{};
    p.rows = rows; p.n = n; p.cpc = cpc; p.pc = pc;
    p.w1 = layers[0]; p.sc1 = layers[1]; p.sh1 = layers[2];
    p.w2 = layers[3]; p.sc2 = layers[4]; p.sh2 = layers[5];
    p.w3 = layers[6]; p.sc3 = layers[7]; p.sh3 = layers[8];
    p.w4 = layers[9]; p.sc4 = layers[10]; p.sh4 = layers[11];
    p.w5 = layers[12]; p.sc5 = layers[13]; p.sh5 = layers[14];
    p.point_feat = point_feat; p.partial = partial_max;
    hipLaunchKernelGGL(seg_head_kernel, dim3((unsigned)(rows / kHeadRows)), dim3(kChainThreads), 0, as_stream(stream), p);
    return check_launch("fpn_v1_seg_head_f32");
}

// rows of partial_max per workgroup: which = 0 seg head, 1 object chains
int fpn_v1_chain_rows(int which) { return which == 0 ? kHeadRows : kObjRows; }

int fpn_v1_obj_chain_f32(long rows, int m, int box, const float *obj, const float *delta,
                         const float *const *layers, float *partial_max, void *stream)
{
    FPN_REQUIRE(rows >= 0 && m >= 1, "v1_obj_chain: bad size");
    FPN_REQUIRE(m % kObjRows == 0 && rows % m == 0, "v1_obj_chain: object points per frustum (%d) must be a multiple of %d", m, kObjRows);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(obj && layers && partial_max, "v1_obj_chain: null pointer");
    const int nl = box ? 12 : 9;
    for (int i = 0; i < nl; ++i) FPN_REQUIRE(layers[i], "v1_obj_chain: layer parameter %d is null", i);
    ObjChainParams p{};
    p.rows = rows; p.m = m; p.obj = obj; p.delta = delta;
    p.w1 = layers[0]; p.sc1 = layers[1]; p.sh1 = layers[2];
    p.w2 = layers[3]; p.sc2 = layers[4]; p.sh2 = layers[5];
    p.w3 = layers[6]; p.sc3 = layers[7]; p.sh3 = layers[8];
    if (box) { p.w4 = layers[9]; p.sc4 = layers[10]; p.sh4 = layers[11]; }
    p.partial = partial_max;
    const dim3 grid((unsigned)(rows / kObjRows));
    if (box) hipLaunchKernelGGL(obj_chain_kernel<true>, grid, dim3(kChainThreads), 0, as_stream(stream), p);
    else hipLaunchKernelGGL(obj_chain_kernel<false>, grid, dim3(kChainThreads), 0, as_stream(stream), p);
    return check_launch("fpn_v1_obj_chain_f32");
}

}  // extern "C"
