// Layer-chain kernels of the v1 instance-segmentation net: a workgroup carries a
// tile of points through SEVERAL consecutive shared-MLP layers with the
// activations resident in LDS, so that only the tile's input and its final
// output touch HBM.
//
//   seg head : conv1 -> conv2 -> conv3 (point_feat, also stored) -> conv4 ->
//              conv5 + max over points                128 points per workgroup
//   seg tail : conv6 (point part + per-frustum bias) -> conv7 -> conv8 -> conv9
//              -> conv10                               64 points per workgroup
//
// Matrix work runs on v_mfma_f32_32x32x2_f32 (exact fp32).  Per 32x32 output tile
// a wave reads its A operand (activations) from the LDS image of the previous
// layer - row-major with an odd row stride (K+1 words), so the 32 rows a
// half-wave reads fall on 32 banks - and its B operand (weights) straight from
// global memory into registers: each wave owns distinct output columns, so a
// weight element is needed by one wave only (two in the layers where two waves
// share a column tile) and staging it through LDS would buy nothing.  Weight
// loads are issued one 32-deep K block ahead of the MFMAs that consume them.
// With 8 waves per workgroup (2 per SIMD) one wave's LDS/global latency is
// covered by the other's MFMAs.
#include <algorithm>
#include <cstdint>

#include "fpn_common.h"

namespace fpn {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kChainThreads = 512;
constexpr int kKB = 16;  // MFMA k-steps (2 k each) per weight prefetch block

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

// acc[tm][tn] += X[tm-th 32-row tile][0:2*KSTEPS] . W[0:2*KSTEPS][tn-th 32-col tile]
//   xs : LDS, already offset to &X[first row + (lane&31)][lane>>5]; row stride XS
//   wg : global, already offset to &W[lane>>5][first col + (lane&31)]; row stride LDW
template <int TM, int TN, int XS>
__device__ __forceinline__ void mfma_block(floatx16 (&acc)[TM][TN], const float *xs, const float (&bq)[TN][kKB])
{
#pragma unroll
    for (int s = 0; s < kKB; ++s) {
        float av[TM];
#pragma unroll
        for (int a = 0; a < TM; ++a) av[a] = xs[a * 32 * XS + 2 * s];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bq[b][s], acc[a][b], 0, 0, 0);
    }
}

template <int TN, int LDW>
__device__ __forceinline__ void load_wblock(float (&bq)[TN][kKB], const float *__restrict__ wg)
{
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int s = 0; s < kKB; ++s) bq[b][s] = wg[(long)(2 * s) * LDW + b * 32];
}

template <int TM, int TN, int KSTEPS, int XS, int LDW>
__device__ __forceinline__ void wave_gemm(floatx16 (&acc)[TM][TN], const float *xs,
                                          const float *__restrict__ wg)
{
    static_assert(KSTEPS % (2 * kKB) == 0, "K must be a multiple of 64");
    constexpr int NB = KSTEPS / kKB;
    // two weight register sets, filled alternately one block ahead of their use;
    // the pair loop is not unrolled so that only one block of loads is in flight
    float b0[TN][kKB], b1[TN][kKB];
    load_wblock<TN, LDW>(b0, wg);
#pragma unroll 1
    for (int kb = 0; kb < NB; kb += 2) {
        load_wblock<TN, LDW>(b1, wg + (long)(2 * (kb + 1) * kKB) * LDW);
        mfma_block<TM, TN, XS>(acc, xs + 2 * kb * kKB, b0);
        if (kb + 2 < NB) load_wblock<TN, LDW>(b0, wg + (long)(2 * (kb + 2) * kKB) * LDW);
        mfma_block<TM, TN, XS>(acc, xs + 2 * (kb + 1) * kKB, b1);
    }
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
    const float *w1, *sc1, *sh1;  // (cpc,64)
    const float *w2, *sc2, *sh2;  // (64,64)
    const float *w3, *sc3, *sh3;  // (64,64)
    const float *w4, *sc4, *sh4;  // (64,128)
    const float *w5, *sc5, *sh5;  // (128,1024)
    float *point_feat;            // (rows,64)
    float *partial;               // (rows/128,1024) per-workgroup column maxima
};

constexpr int kHeadRows = 128;

__global__ __launch_bounds__(kChainThreads) void seg_head_kernel(const SegHeadParams p)
{
    __shared__ float B0[kHeadRows * 65];
    __shared__ float B1[kHeadRows * 65];
    __shared__ float B2[kHeadRows * 129];
    __shared__ float W1s[8 * 64 + 128];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const long row0 = (long)blockIdx.x * kHeadRows;

    // conv1 on the vector ALU (K = 3 or 4): stage W1 / scale / shift in LDS
    for (int i = tid; i < p.cpc * 64; i += kChainThreads) W1s[i] = p.w1[i];
    if (tid < 64) {
        W1s[8 * 64 + tid] = p.sc1[tid];
        W1s[8 * 64 + 64 + tid] = p.sh1[tid];
    }
    __syncthreads();
    {
        const int row = tid >> 2, cg = tid & 3;
        float x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = c < p.cpc ? p.pc[(row0 + row) * p.cpc + c] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int col = cg * 16 + j;
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < p.cpc) v = fmaf(x[c], W1s[c * 64 + col], v);
            B0[row * 65 + col] = fmaxf(v * W1s[8 * 64 + col] + W1s[8 * 64 + 64 + col], 0.f);
        }
    }
    __syncthreads();

    // conv2: B0 -> B1   (4 row tiles x 2 column tiles, one per wave)
    {
        const int rt = w & 3, ct = w >> 2;
        floatx16 acc[1][1];
        zero_acc(acc);
        wave_gemm<1, 1, 32, 65, 64>(acc, B0 + (rt * 32 + l31) * 65 + half, p.w2 + half * 64 + ct * 32 + l31);
        const int col = ct * 32 + l31;
        store_tile_lds<65>(acc[0][0], B1 + rt * 32 * 65 + ct * 32, p.sc2[col], p.sh2[col], l31, half);
    }
    __syncthreads();
    // conv3: B1 -> B0 (= point_feat, also written to memory for the tail kernel)
    {
        const int rt = w & 3, ct = w >> 2;
        floatx16 acc[1][1];
        zero_acc(acc);
        wave_gemm<1, 1, 32, 65, 64>(acc, B1 + (rt * 32 + l31) * 65 + half, p.w3 + half * 64 + ct * 32 + l31);
        const int col = ct * 32 + l31;
        const float sc = p.sc3[col], sh = p.sh3[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + acc_row(r, half);
            const float v = fmaxf(acc[0][0][r] * sc + sh, 0.f);
            B0[row * 65 + col] = v;
            p.point_feat[(row0 + row) * 64 + col] = v;
        }
    }
    __syncthreads();
    // conv4: B0 -> B2   (4 row tiles x 4 column tiles, two row tiles per wave)
    {
        const int ct = w & 3, rt0 = (w >> 2) * 2;
        floatx16 acc[2][1];
        zero_acc(acc);
        wave_gemm<2, 1, 32, 65, 128>(acc, B0 + (rt0 * 32 + l31) * 65 + half, p.w4 + half * 128 + ct * 32 + l31);
        const int col = ct * 32 + l31;
        const float sc = p.sc4[col], sh = p.sh4[col];
#pragma unroll
        for (int a = 0; a < 2; ++a) store_tile_lds<129>(acc[a][0], B2 + (rt0 + a) * 32 * 129 + ct * 32, sc, sh, l31, half);
    }
    __syncthreads();
    // conv5 + max over the 128 points: wave w owns column tiles w, w+8, w+16, w+24
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const int ct = w + 8 * i;
        floatx16 acc[4][1];
        zero_acc(acc);
        wave_gemm<4, 1, 64, 129, 1024>(acc, B2 + l31 * 129 + half, p.w5 + half * 1024 + ct * 32 + l31);
        const int col = ct * 32 + l31;
        const float sc = p.sc5[col], sh = p.sh5[col];
        float m = 0.f;  // ReLU output is >= 0
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[a][0][r] * sc + sh);
        m = fmaxf(m, __shfl_xor(m, 32));
        if (half == 0) p.partial[(long)blockIdx.x * 1024 + col] = m;
    }
}

// ---------------------------------------------------------------------------
// seg tail
// ---------------------------------------------------------------------------
struct SegTailParams {
    long rows;
    int n;  // points per frustum (multiple of 64)
    const float *point_feat;  // (rows,64)
    const float *gbias;       // (rows/n,512): [global_feat | one_hot] . W6[64:]
    const float *w6, *sc6, *sh6;  // (64,512) point rows of conv6
    const float *w7, *sc7, *sh7;  // (512,256)
    const float *w8, *sc8, *sh8;  // (256,128)
    const float *w9, *sc9, *sh9;  // (128,128)
    const float *w10, *b10;       // (128,2), (2)
    float *logits;                // (rows,2)
};

constexpr int kTailRows = 64;

__global__ __launch_bounds__(kChainThreads) void seg_tail_kernel(const SegTailParams p)
{
    __shared__ float PF[kTailRows * 65];
    __shared__ float H[kTailRows * 129];   // conv6 chunk (128 wide); later conv8's output
    __shared__ float A7[kTailRows * 257];  // conv7's output; later conv9's output (stride 129)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const long row0 = (long)blockIdx.x * kTailRows;
    const long frustum = row0 / p.n;

    // point_feat tile -> LDS
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + kChainThreads * i;
        const int row = e >> 4, q = e & 15;
        const float4 v = *reinterpret_cast<const float4 *>(p.point_feat + (row0 + row) * 64 + 4 * q);
        float *d = PF + row * 65 + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();

    floatx16 acc7[2][1];
    zero_acc(acc7);
    const int rt = w & 1, ct = w >> 1;  // tile of a 64 x 128 output (conv6 chunk, conv8, conv9)
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
        // conv6, output columns [128 kc, 128 kc + 128): PF -> H
        {
            floatx16 acc[1][1];
            zero_acc(acc);
            const int col = kc * 128 + ct * 32 + l31;
            wave_gemm<1, 1, 32, 65, 512>(acc, PF + (rt * 32 + l31) * 65 + half, p.w6 + half * 512 + col);
            const float g = p.gbias[frustum * 512 + col], sc = p.sc6[col], sh = p.sh6[col];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                H[(rt * 32 + acc_row(r, half)) * 129 + ct * 32 + l31] = fmaxf((acc[0][0][r] + g) * sc + sh, 0.f);
        }
        __syncthreads();
        // conv7 += H . W7[128 kc : 128 kc + 128, :]; wave w owns output columns [32 w, 32 w + 32)
        wave_gemm<2, 1, 64, 129, 256>(acc7, H + l31 * 129 + half, p.w7 + (long)(kc * 128 + half) * 256 + w * 32 + l31);
        __syncthreads();
    }
    {
        const int col = w * 32 + l31;
        const float sc = p.sc7[col], sh = p.sh7[col];
#pragma unroll
        for (int a = 0; a < 2; ++a) store_tile_lds<257>(acc7[a][0], A7 + a * 32 * 257 + w * 32, sc, sh, l31, half);
    }
    __syncthreads();
    // conv8: A7 -> H (stride 129)
    {
        floatx16 acc[1][1];
        zero_acc(acc);
        const int col = ct * 32 + l31;
        wave_gemm<1, 1, 128, 257, 128>(acc, A7 + (rt * 32 + l31) * 257 + half, p.w8 + half * 128 + col);
        store_tile_lds<129>(acc[0][0], H + rt * 32 * 129 + ct * 32, p.sc8[col], p.sh8[col], l31, half);
    }
    __syncthreads();
    // conv9: H -> A7 region (stride 129)
    float *A9 = A7;
    {
        floatx16 acc[1][1];
        zero_acc(acc);
        const int col = ct * 32 + l31;
        wave_gemm<1, 1, 64, 129, 128>(acc, H + (rt * 32 + l31) * 129 + half, p.w9 + half * 128 + col);
        store_tile_lds<129>(acc[0][0], A9 + rt * 32 * 129 + ct * 32, p.sc9[col], p.sh9[col], l31, half);
    }
    __syncthreads();
    // conv10 (128 -> 2, no batch norm, no activation) on the vector ALU: 8 lanes per point
    {
        const int row = tid >> 3, part = tid & 7;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = part * 16 + j;
            const float a = A9[row * 129 + k];
            s0 = fmaf(a, p.w10[k * 2 + 0], s0);
            s1 = fmaf(a, p.w10[k * 2 + 1], s1);
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            s0 += __shfl_xor(s0, off);
            s1 += __shfl_xor(s1, off);
        }
        if (part == 0) {
            float2 o = make_float2(s0 + p.b10[0], s1 + p.b10[1]);
            *reinterpret_cast<float2 *>(p.logits + (row0 + row) * 2) = o;
        }
    }
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
    const float *w1, *sc1, *sh1;  // (3,128)
    const float *w2, *sc2, *sh2;  // (128,128)
    const float *w3, *sc3, *sh3;  // (128,256)
    const float *w4, *sc4, *sh4;  // (256,512)  box net only
    float *partial;               // (rows/64, 256 or 512) per-workgroup column maxima
};

constexpr int kObjRows = 64;

template <bool BOX>
__global__ __launch_bounds__(kChainThreads) void obj_chain_kernel(const ObjChainParams p)
{
    __shared__ float A[kObjRows * 129];
    __shared__ float B[kObjRows * 129];
    __shared__ float C[BOX ? kObjRows * 257 : 1];
    __shared__ float W1s[3 * 128 + 256];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const long row0 = (long)blockIdx.x * kObjRows;

    if (tid < 3 * 128) W1s[tid] = p.w1[tid];
    if (tid < 128) {
        W1s[3 * 128 + tid] = p.sc1[tid];
        W1s[3 * 128 + 128 + tid] = p.sh1[tid];
    }
    __syncthreads();
    {   // layer 1 (K = 3) on the vector ALU: 8 lanes per point, 16 columns each
        const int row = tid >> 3, cg = tid & 7;
        const float *src = p.obj + (row0 + row) * 3;
        float x0 = src[0], x1 = src[1], x2 = src[2];
        if (p.delta) {
            const float *d = p.delta + (row0 / p.m) * 3;
            x0 -= d[0]; x1 -= d[1]; x2 -= d[2];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int col = cg * 16 + j;
            float v = fmaf(x2, W1s[256 + col], fmaf(x1, W1s[128 + col], x0 * W1s[col]));
            A[row * 129 + col] = fmaxf(v * W1s[3 * 128 + col] + W1s[3 * 128 + 128 + col], 0.f);
        }
    }
    __syncthreads();
    {   // layer 2: A -> B, 2 row tiles x 4 column tiles
        const int rt = w & 1, ct = w >> 1;
        floatx16 acc[1][1];
        zero_acc(acc);
        const int col = ct * 32 + l31;
        wave_gemm<1, 1, 64, 129, 128>(acc, A + (rt * 32 + l31) * 129 + half, p.w2 + half * 128 + col);
        store_tile_lds<129>(acc[0][0], B + rt * 32 * 129 + ct * 32, p.sc2[col], p.sh2[col], l31, half);
    }
    __syncthreads();
    {   // layer 3: B -> 256 columns, wave w owns column tile w and both row tiles
        floatx16 acc[2][1];
        zero_acc(acc);
        const int col = w * 32 + l31;
        wave_gemm<2, 1, 64, 129, 256>(acc, B + l31 * 129 + half, p.w3 + half * 256 + col);
        const float sc = p.sc3[col], sh = p.sh3[col];
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
        for (int a = 0; a < 2; ++a) store_tile_lds<257>(acc[a][0], C + a * 32 * 257 + w * 32, sc, sh, l31, half);
    }
    if (BOX) {
        __syncthreads();
        // layer 4: C -> 512 columns + max over the 64 points; column tiles w and w + 8
#pragma unroll 1
        for (int i = 0; i < 2; ++i) {
            floatx16 acc[2][1];
            zero_acc(acc);
            const int col = (w + 8 * i) * 32 + l31;
            wave_gemm<2, 1, 128, 257, 512>(acc, C + l31 * 257 + half, p.w4 + half * 512 + col);
            const float sc = p.sc4[col], sh = p.sh4[col];
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

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" {

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

int fpn_v1_seg_tail_f32(long rows, int n, const float *point_feat, const float *group_bias,
                        const float *const *layers, float *logits, void *stream)
{
    FPN_REQUIRE(rows >= 0 && n >= 1, "v1_seg_tail: bad size");
    FPN_REQUIRE(n % kTailRows == 0 && rows % n == 0, "v1_seg_tail: points per frustum (%d) must be a multiple of %d", n, kTailRows);
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(point_feat && group_bias && layers && logits, "v1_seg_tail: null pointer");
    for (int i = 0; i < 14; ++i) FPN_REQUIRE(layers[i], "v1_seg_tail: layer parameter %d is null", i);
    SegTailParams p{};
    p.rows = rows; p.n = n; p.point_feat = point_feat; p.gbias = group_bias;
    p.w6 = layers[0]; p.sc6 = layers[1]; p.sh6 = layers[2];
    p.w7 = layers[3]; p.sc7 = layers[4]; p.sh7 = layers[5];
    p.w8 = layers[6]; p.sc8 = layers[7]; p.sh8 = layers[8];
    p.w9 = layers[9]; p.sc9 = layers[10]; p.sh9 = layers[11];
    p.w10 = layers[12]; p.b10 = layers[13];
    p.logits = logits;
    hipLaunchKernelGGL(seg_tail_kernel, dim3((unsigned)(rows / kTailRows)), dim3(kChainThreads), 0, as_stream(stream), p);
    return check_launch("fpn_v1_seg_tail_f32");
}

}  // extern "C"

extern "C" int fpn_v1_obj_chain_f32(long rows, int m, int box, const float *obj, const float *delta,
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
