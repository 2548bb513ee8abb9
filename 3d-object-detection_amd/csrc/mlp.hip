// Shared-MLP layer (1x1 convolution / fully connected) on the gfx950 fp32
// matrix cores, with bias + folded batch norm + ReLU and the max-pool over
// points fused into the epilogue, plus the Flipout forms of the same layer.
//
// Tiling.  A workgroup of 4 waves owns a 128-row x BN-column output tile
// (BN = 128, 64 or 32) and walks K in steps of 32 through double-buffered LDS
// tiles (one barrier per step; the global loads of step t+1 are in flight while
// step t multiplies).  Each wave owns TM x TN accumulator tiles of 32x32 and
// feeds v_mfma_f32_32x32x2_f32 from LDS:
//   A operand  = activations  X[row = lane&31][k = 2s + (lane>>5)]
//                LDS image [128][33] (row stride 33 words: the 32 rows of a
//                half-wave read fall on 32 different banks)
//   B operand  = weights      W[k = 2s + (lane>>5)][col = lane&31]
//                LDS image [32][BN] (a half-wave reads 32 consecutive words)
//   C/D        = lane holds output column lane&31; register r holds row
//                (r&3) + 8*(r>>2) + 4*(lane>>5)
// so per-column epilogue constants (scale, shift) are one value per lane, the
// max over points is a max over a lane's 16 registers plus one cross-half
// exchange, and stores are 128-byte runs per half-wave.
//
// fp32 MFMA is an exact fp32 FMA chain in k order: results differ from a CPU
// fp32 GEMM only by summation order.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <string>

#include "fpn_common.h"

namespace fpn {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kMlpThreads = 256;
constexpr int kBM = 128;  // rows per workgroup (all configurations)
constexpr int kBK = 32;   // K per LDS tile
constexpr int kXS = kBK + 1;

enum { kPlain = 0, kFlipoutFold = 1, kFlipoutTwoPass = 2 };

struct MlpParams {
    long rows;
    int cin, cout;
    const float *x;
    int ldx;
    const float *w;        // (cin,cout); the mean weights in the Flipout modes
    const float *w_delta;  // (cin,cout) perturbation of this launch's sample
    const float *sign_in;  // (groups,cin)  +-1
    const float *sign_out; // (groups,cout) +-1
    const float *scale, *shift;
    const float *group_bias;
    int rows_per_group;
    int relu;
    int pool_rows;
    float *y;
    int ldy;
    float *partial;  // (row blocks, cout) when pool_rows > kBM
    // Flipout: gridDim.y Monte-Carlo samples in one launch; sample s works on these
    // arrays advanced by s times the stride (elements)
    long sx, swd, ssi, sso, sgb, sy, spartial;
};

// MODE kPlain:        acc = x.w
// MODE kFlipoutFold:  every row of the workgroup belongs to one sign group g, so
//                     x.w + ((x*si).wd)*so == x.(w + si^T * wd * so): the
//                     perturbation is folded into the weight tile while it is
//                     staged (sign flips are exact) and the layer costs one
//                     product instead of two.
// MODE kFlipoutTwoPass: any grouping (fully connected layers: one group per
//                     row).  K is walked twice: first (x*si).wd, then the
//                     accumulators are multiplied by so, then x.w is added on
//                     top - no second accumulator set.
template <int WM, int WN, int TM, int TN, int MODE>
__global__ __launch_bounds__(kMlpThreads) void shared_mlp_kernel(const MlpParams pin)
{
    MlpParams p = pin;
    if (MODE != kPlain) {
        const long s = blockIdx.y;
        p.x += s * pin.sx;
        p.w_delta += s * pin.swd;
        p.sign_in += s * pin.ssi;
        p.sign_out += s * pin.sso;
        if (p.group_bias) p.group_bias += s * pin.sgb;
        p.y += s * pin.sy;
        if (p.partial) p.partial += s * pin.spartial;
    }
    constexpr int BN = WN * TN * 32;
    constexpr int BM = WM * TM * 32;  // rows per workgroup: 128, or 32 in the small-batch configuration
    static_assert(WM * WN * kWave == kMlpThreads, "4 waves");
    constexpr int XREG = BM * kBK / kMlpThreads;  // 16 floats of X per thread per tile
    constexpr int WREG = kBK * BN / kMlpThreads;   // 16 / 8 / 4 floats of W
    constexpr int NQ4 = BN / 4;                    // float4 per W tile row
    constexpr int WROWS = kMlpThreads / NQ4;       // W tile rows per staging pass

    __shared__ float Xs[2][BM * kXS];
    __shared__ __attribute__((aligned(16))) float Ws[2][kBK * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int ncb = (p.cout + BN - 1) / BN;
    const long row0 = (long)(blockIdx.x / ncb) * BM;
    const int col0 = (int)(blockIdx.x % ncb) * BN;

    const bool xvec = (p.ldx % 4 == 0) && (p.cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    const bool wvec = (p.cout % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.w) & 15) == 0) &&
                      (MODE == kPlain || (reinterpret_cast<uintptr_t>(p.w_delta) & 15) == 0);

    const int ktiles = (p.cin + kBK - 1) / kBK;
    const int passes = MODE == kFlipoutTwoPass ? 2 : 1;
    const int total_tiles = ktiles * passes;

    // sign group of the whole workgroup (fold mode)
    const long gblock = MODE == kFlipoutFold ? row0 / p.rows_per_group : 0;

    float xr[XREG];
    float wr[WREG];

    // ---- global -> register staging of tile t ---------------------------------
    auto load_tile = [&](int t) {
        const int pass = t / ktiles;                 // two-pass mode: 0 = perturbation, 1 = mean
        const int k0 = (t - pass * ktiles) * kBK;
        const bool perturb = MODE == kFlipoutTwoPass && pass == 0;
        // X
        if (xvec) {
            const int kq = tid & 7, r = tid >> 3;
#pragma unroll
            for (int i = 0; i < XREG / 4; ++i) {
                const long row = row0 + r + 32 * i;
                const int k = k0 + 4 * kq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < p.rows && k < p.cin) {
                    v = *reinterpret_cast<const float4 *>(p.x + row * p.ldx + k);
                    if (perturb) {
                        const float *si = p.sign_in + (row / p.rows_per_group) * p.cin + k;
                        v.x *= si[0]; v.y *= si[1]; v.z *= si[2]; v.w *= si[3];
                    }
                }
                xr[4 * i + 0] = v.x; xr[4 * i + 1] = v.y; xr[4 * i + 2] = v.z; xr[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < XREG; ++i) {
                const int e = tid + kMlpThreads * i;
                const long row = row0 + (e >> 5);
                const int k = k0 + (e & 31);
                float v = 0.f;
                if (row < p.rows && k < p.cin) {
                    v = p.x[row * p.ldx + k];
                    if (perturb) v *= p.sign_in[(row / p.rows_per_group) * p.cin + k];
                }
                xr[i] = v;
            }
        }
        // W
        const float *wsrc = perturb ? p.w_delta : p.w;
        if (wvec) {
            const int c4 = tid % NQ4, kr = tid / NQ4;
            const int col = col0 + 4 * c4;
#pragma unroll
            for (int i = 0; i < WREG / 4; ++i) {
                const int k = k0 + kr + WROWS * i;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.cin && col < p.cout) {
                    v = *reinterpret_cast<const float4 *>(wsrc + (long)k * p.cout + col);
                    if (MODE == kFlipoutFold) {
                        const float4 d = *reinterpret_cast<const float4 *>(p.w_delta + (long)k * p.cout + col);
                        const float si = p.sign_in[gblock * p.cin + k];
                        const float *so = p.sign_out + gblock * p.cout + col;
                        v.x += si * d.x * so[0];
                        v.y += si * d.y * so[1];
                        v.z += si * d.z * so[2];
                        v.w += si * d.w * so[3];
                    }
                }
                wr[4 * i + 0] = v.x; wr[4 * i + 1] = v.y; wr[4 * i + 2] = v.z; wr[4 * i + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < WREG; ++i) {
                const int e = tid + kMlpThreads * i;
                const int k = k0 + e / BN;
                const int col = col0 + e % BN;
                float v = 0.f;
                if (k < p.cin && col < p.cout) {
                    v = wsrc[(long)k * p.cout + col];
                    if (MODE == kFlipoutFold)
                        v += p.sign_in[gblock * p.cin + k] * p.w_delta[(long)k * p.cout + col] *
                             p.sign_out[gblock * p.cout + col];
                }
                wr[i] = v;
            }
        }
    };

    // ---- register -> LDS ------------------------------------------------------
    auto store_tile = [&](int buf) {
        float *xs = Xs[buf];
        float *ws = Ws[buf];
        if (xvec) {
            const int kq = tid & 7, r = tid >> 3;
#pragma unroll
            for (int i = 0; i < XREG / 4; ++i) {
                float *dst = xs + (r + 32 * i) * kXS + 4 * kq;
                dst[0] = xr[4 * i + 0]; dst[1] = xr[4 * i + 1]; dst[2] = xr[4 * i + 2]; dst[3] = xr[4 * i + 3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < XREG; ++i) {
                const int e = tid + kMlpThreads * i;
                xs[(e >> 5) * kXS + (e & 31)] = xr[i];
            }
        }
        if (wvec) {
            const int c4 = tid % NQ4, kr = tid / NQ4;
#pragma unroll
            for (int i = 0; i < WREG / 4; ++i)
                *reinterpret_cast<float4 *>(ws + (kr + WROWS * i) * BN + 4 * c4) =
                    make_float4(wr[4 * i + 0], wr[4 * i + 1], wr[4 * i + 2], wr[4 * i + 3]);
        } else {
#pragma unroll
            for (int i = 0; i < WREG; ++i) {
                const int e = tid + kMlpThreads * i;
                ws[e] = wr[i];
            }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < total_tiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < total_tiles) load_tile(t + 1);

        if (MODE == kFlipoutTwoPass && t == ktiles) {
            // perturbation product finished: multiply by the output signs of each
            // row's group, then keep accumulating the mean product on top
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int col = col0 + (wn * TN + b) * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long row = row0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float so = 1.f;
                        if (row < p.rows && col < p.cout)
                            so = p.sign_out[(row / p.rows_per_group) * p.cout + col];
                        acc[a][b][r] *= so;
                    }
                }
        }

        const float *xs = Xs[buf] + (wm * TM * 32 + l31) * kXS + half;
        const float *ws = Ws[buf] + half * BN + wn * TN * 32 + l31;
#pragma unroll
        for (int s = 0; s < kBK / 2; ++s) {
            float av[TM], bv[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) av[a] = xs[a * 32 * kXS + 2 * s];
#pragma unroll
            for (int b = 0; b < TN; ++b) bv[b] = ws[2 * s * BN + b * 32];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
        }

        if (t + 1 < total_tiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------
    float *red = Xs[0];  // (BM/32, BN) tile maxima; the K loop ended on a barrier
    const bool pooled = p.pool_rows > 0;
    const bool gb_tile_uniform = p.group_bias && (p.rows_per_group % 32 == 0);
    const bool gb_per_row = p.rows_per_group == 1;       // (a bias row per row: no 64-bit division per element)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = col0 + (wn * TN + b) * 32 + l31;
        const bool colok = col < p.cout;
        const float sc = (colok && p.scale) ? p.scale[col] : 1.f;
        const float sh = (colok && p.shift) ? p.shift[col] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int trow = (wm * TM + a) * 32;  // tile's first row inside the block
            float gbu = 0.f;
            if (gb_tile_uniform && colok && row0 + trow < p.rows)
                gbu = p.group_bias[((row0 + trow) / p.rows_per_group) * p.cout + col];
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + trow + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[a][b][r];
                if (p.group_bias) {
                    if (gb_tile_uniform) v += gbu;
                    else if (colok && row < p.rows) v += p.group_bias[(gb_per_row ? row : row / p.rows_per_group) * p.cout + col];
                }
                v = v * sc + sh;
                if (p.relu) v = fmaxf(v, 0.f);
                if (pooled) m = fmaxf(m, v);
                else if (colok && row < p.rows) p.y[row * p.ldy + col] = v;
            }
            if (pooled) {
                m = fmaxf(m, __shfl_xor(m, 32));
                if (half == 0) red[(wm * TM + a) * BN + (wn * TN + b) * 32 + l31] = m;
            }
        }
    }
    if (!pooled) return;
    __syncthreads();
    // rows is a multiple of pool_rows and pool_rows of 32: a 32-row tile is
    // either all real rows or all padding
    const int col = col0 + tid;
    if (tid < BN && col < p.cout) {
        if (p.pool_rows <= BM) {
            const int tiles_per_group = p.pool_rows / 32;
            for (int g = 0; g < BM / p.pool_rows; ++g) {
                const long grow = row0 + (long)g * p.pool_rows;
                if (grow >= p.rows) break;
                float m = red[g * tiles_per_group * BN + tid];
                for (int u = 1; u < tiles_per_group; ++u) m = fmaxf(m, red[(g * tiles_per_group + u) * BN + tid]);
                p.y[(grow / p.pool_rows) * p.ldy + col] = m;
            }
        } else {
            float m = red[tid];
            for (int u = 1; u < BM / 32; ++u) m = fmaxf(m, red[u * BN + tid]);
            p.partial[(row0 / BM) * p.cout + col] = m;
        }
    }
}

// ---------------------------------------------------------------------------
// The same layer on the bf16 matrix instruction (v_mfma_f32_32x32x16_bf16, fp32
// accumulation) for the wide per-point layers of the PointNet++ nets whose inputs are
// features, not coordinates: x is converted to bf16 while it is staged, the weights
// arrive transposed and converted (fpn_mlp_pack_bf16: wt (cout, cin rounded up to 64),
// zero beyond cin), so that both LDS images hold a row's / a column's K values
// contiguously and an operand fragment (8 consecutive k of one row or column: lane
// (i = lane&31, g = lane>>5) supplies k = 16 s + 8 g .. + 7) is one 16-byte read.  Rows
// of both images are 72 bf16 = 144 bytes apart: 16 consecutive rows then start in 16
// different groups of four banks.  A workgroup owns 128 rows x 128 columns (2 x 2 waves,
// 2 x 2 tiles each) and walks K 64 at a time through double-buffered images, one barrier
// per step.  (Requesting tile t + 2 into a second register set while tile t multiplies was
// measured and made the kernel a third slower - 175 -> 259 us on 131 k rows x 320 -> 320.)  Epilogue as shared_mlp_kernel (group bias, scale, shift, ReLU, max over
// groups of rows), all in fp32.
// ---------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int kHK = 64;        // K per LDS tile
constexpr int kHS = kHK + 8;   // image row stride (bf16)

__global__ void mlp_pack_bf16_kernel(int cin, int cout, int ldw, const float *__restrict__ w, __bf16 *__restrict__ wt)
{
    const long total = (long)cout * ldw;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int col = (int)(e / ldw), k = (int)(e % ldw);
        wt[e] = (__bf16)(k < cin ? w[(long)k * cout + col] : 0.f);
    }
}

__global__ __launch_bounds__(kMlpThreads) void shared_mlp_bf16_kernel(const MlpParams p)
{
    constexpr int BM = 128, BN = 128;
    __shared__ __attribute__((aligned(16))) __bf16 Xs[2][BM * kHS];
    __shared__ __attribute__((aligned(16))) __bf16 Ws[2][BN * kHS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const int ncb = (p.cout + BN - 1) / BN;
    // The column tiles of one block of rows read the same rows of x: consecutive workgroup ids go round the eight
    // XCDs, each with its own L2, so id i takes logical block (i % 8) * (grid / 8) + i / 8 - the tiles of a row
    // block then share an XCD (measured on 131 k rows x 320 -> 320: 180 -> 169 us).
    const unsigned g = gridDim.x, bi = blockIdx.x;
    const unsigned blk = (ncb > 1 && g % 8 == 0) ? (bi % 8) * (g / 8) + bi / 8 : bi;
    const long row0 = (long)(blk / ncb) * BM;
    const int col0 = (int)(blk % ncb) * BN;
    const __bf16 *wt = reinterpret_cast<const __bf16 *>(p.w);
    const int ldw = (p.cin + kHK - 1) / kHK * kHK;
    const int ktiles = ldw / kHK;
    const bool xvec = (p.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);

    float4 xr[8];
    bf16x8 wr[4];
    auto load_tile = [&](int t) {
        const int k0 = t * kHK;
        const int kq = tid & 15, r = tid >> 4;
        const int k = k0 + 4 * kq;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long row = row0 + r + 16 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.rows) {
                const float *src = p.x + row * p.ldx + k;
                if (xvec && k + 3 < p.cin) {
                    v = *reinterpret_cast<const float4 *>(src);
                } else {
                    if (k + 0 < p.cin) v.x = src[0];
                    if (k + 1 < p.cin) v.y = src[1];
                    if (k + 2 < p.cin) v.z = src[2];
                    if (k + 3 < p.cin) v.w = src[3];
                }
            }
            xr[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + kMlpThreads * i, col = col0 + (id >> 3), kv = id & 7;
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            if (col < p.cout) v = *reinterpret_cast<const bf16x8 *>(wt + (long)col * ldw + k0 + 8 * kv);
            wr[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        const int kq = tid & 15, r = tid >> 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bf16x4 v;
            v[0] = (__bf16)xr[i].x; v[1] = (__bf16)xr[i].y; v[2] = (__bf16)xr[i].z; v[3] = (__bf16)xr[i].w;
            *reinterpret_cast<bf16x4 *>(&Xs[buf][(r + 16 * i) * kHS + 4 * kq]) = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + kMlpThreads * i;
            *reinterpret_cast<bf16x8 *>(&Ws[buf][(id >> 3) * kHS + 8 * (id & 7)]) = wr[i];
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ktiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ktiles) load_tile(t + 1);
        const __bf16 *xs = &Xs[buf][(wm * 64 + l31) * kHS + 8 * half];
        const __bf16 *ws = &Ws[buf][(wn * 64 + l31) * kHS + 8 * half];
#pragma unroll
        for (int s = 0; s < kHK / 16; ++s) {
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = *reinterpret_cast<const bf16x8 *>(xs + a * 32 * kHS + 16 * s);
#pragma unroll
            for (int b = 0; b < 2; ++b) bv[b] = *reinterpret_cast<const bf16x8 *>(ws + b * 32 * kHS + 16 * s);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (t + 1 < ktiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue (as shared_mlp_kernel) ------------------------------------------
    float *red = reinterpret_cast<float *>(Xs[0]);  // (BM/32, BN) tile maxima; the K loop ended on a barrier
    const bool pooled = p.pool_rows > 0;
    const bool gb_tile_uniform = p.group_bias && (p.rows_per_group % 32 == 0);
    const bool gb_per_row = p.rows_per_group == 1;       // (a bias row per row: no 64-bit division per element)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + (wn * 2 + b) * 32 + l31;
        const bool colok = col < p.cout;
        const float sc = (colok && p.scale) ? p.scale[col] : 1.f;
        const float sh = (colok && p.shift) ? p.shift[col] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int trow = (wm * 2 + a) * 32;
            float gbu = 0.f;
            if (gb_tile_uniform && colok && row0 + trow < p.rows)
                gbu = p.group_bias[((row0 + trow) / p.rows_per_group) * p.cout + col];
            float m = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long row = row0 + trow + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[a][b][r];
                if (p.group_bias) {
                    if (gb_tile_uniform) v += gbu;
                    else if (colok && row < p.rows) v += p.group_bias[(gb_per_row ? row : row / p.rows_per_group) * p.cout + col];
                }
                v = v * sc + sh;
                if (p.relu) v = fmaxf(v, 0.f);
                if (pooled) m = fmaxf(m, v);
                else if (colok && row < p.rows) p.y[row * p.ldy + col] = v;
            }
            if (pooled) {
                m = fmaxf(m, __shfl_xor(m, 32));
                if (half == 0) red[(wm * 2 + a) * BN + (wn * 2 + b) * 32 + l31] = m;
            }
        }
    }
    if (!pooled) return;
    __syncthreads();
    const int col = col0 + tid;
    if (tid < BN && col < p.cout) {
        if (p.pool_rows <= BM) {
            const int tiles_per_group = p.pool_rows / 32;
            for (int g = 0; g < BM / p.pool_rows; ++g) {
                const long grow = row0 + (long)g * p.pool_rows;
                if (grow >= p.rows) break;
                float m = red[g * tiles_per_group * BN + tid];
                for (int u = 1; u < tiles_per_group; ++u) m = fmaxf(m, red[(g * tiles_per_group + u) * BN + tid]);
                p.y[(grow / p.pool_rows) * p.ldy + col] = m;
            }
        } else {
            float m = red[tid];
            for (int u = 1; u < BM / 32; ++u) m = fmaxf(m, red[u * BN + tid]);
            p.partial[(row0 / BM) * p.cout + col] = m;
        }
    }
}

// ---------------------------------------------------------------------------
// The same bf16 layer for MANY rows and FEW input channels (the per-point products and
// feature-propagation layers of the PointNet++ nets: 131 k rows, 128 .. 320 channels in):
// such a layer is bound by reading x and writing y once, and the kernel above spends most
// of a workgroup's life waiting - its K loop is two to five steps long, the loads of one
// step ahead are all it has in flight, and filling and draining that pipeline per 128 rows
// leaves HBM a quarter used.  Here the workgroups are persistent: a workgroup keeps its 128
// columns of the weights in LDS for the whole launch (all of K: cout tile x (ldw + 8) bf16)
// and walks a strided stream of 128-row blocks, whose 64-wide K tiles form ONE sequence of
// steps across block boundaries.  The loads of steps s + 1 and s + 2 are in flight (two
// register sets, the loop unrolled by two so that each is addressed statically) while step
// s multiplies, so the queue never drains between blocks; x tiles go through the same
// double-buffered bf16 image, one barrier per step.  Same operand slots, same k order and
// the same epilogue arithmetic as shared_mlp_bf16_kernel: the results are bit-identical.
// No pooling here (the pooled layers belong to the fused set-abstraction kernels).
// Workgroup ids are renumbered so that the column tiles of one stream of row blocks run on
// one XCD and share its L2 copy of x.
// ---------------------------------------------------------------------------
// The epilogue of one 128 x 128 block of the persistent kernel, specialised by the kind of group bias (GB: 0 none,
// 1 one bias row per 32-row tile, 2 one per row, 3 any grouping).  Every block lies wholly inside the rows (the
// launcher gives a ragged tail to the per-tile kernel), so the only predicate is the lane's column, and the body is
// straight-line code whose memory-counter bookkeeping the compiler keeps exact.  Loads return in issue order, so a
// bias value requested here would wait for every x load already in flight: kinds 1 and 2 are requested by
// stream_bias() at the top of the block's last K step, BEFORE that step's x prefetch, and only read here.
template <int GB>
struct StreamBias {
    static constexpr int N = GB == 2 ? 64 : GB == 1 ? 4 : 1;
    float v[N];
};

template <int GB>
__device__ __forceinline__ void stream_bias(const MlpParams &p, StreamBias<GB> &gb, long row0, int col0, int wm, int wn, int l31,
                                            int half)
{
    if (GB != 1 && GB != 2) return;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = min(col0 + (wn * 2 + b) * 32 + l31, p.cout - 1);      // (a column past the end is never stored)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const long trow = row0 + (wm * 2 + a) * 32 + 4 * half;
            if (GB == 1) {
                gb.v[b * 2 + a] = p.group_bias[(trow / p.rows_per_group) * p.cout + col];
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    gb.v[(b * 2 + a) * 16 + q] = p.group_bias[(trow + (q & 3) + 8 * (q >> 2)) * p.cout + col];
            }
        }
    }
}

template <int GB>
__device__ __forceinline__ void stream_finish(const MlpParams &p, const floatx16 (&acc)[2][2], const StreamBias<GB> &gb,
                                              const float (&sc)[2], const float (&sh)[2], long row0, int col0, int wm, int wn,
                                              int l31, int half)
{
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + (wn * 2 + b) * 32 + l31;
        if (col < p.cout) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const long trow = row0 + (wm * 2 + a) * 32 + 4 * half;
                float g3[16];
                if (GB == 3) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        g3[q] = p.group_bias[((trow + (q & 3) + 8 * (q >> 2)) / p.rows_per_group) * p.cout + col];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const long row = trow + (q & 3) + 8 * (q >> 2);
                    float v = acc[a][b][q];
                    if (GB == 1) v += gb.v[b * 2 + a];
                    if (GB == 2) v += gb.v[(b * 2 + a) * 16 + q];
                    if (GB == 3) v += g3[q];
                    v = v * sc[b] + sh[b];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.y[row * p.ldy + col] = v;
                }
            }
        }
    }
}

template <int GB>
__global__ __launch_bounds__(kMlpThreads) void shared_mlp_bf16_stream_kernel(const MlpParams p, const int ncb, const int streams,
                                                                             const int nrb)
{
    constexpr int BM = 128, BN = 128;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds_stream[];
    const int ldw = (p.cin + kHK - 1) / kHK * kHK;
    const int ktiles = ldw / kHK;
    const int wstride = ldw + 8;
    __bf16 *Wl = lds_stream;                       // [BN][ldw + 8]
    __bf16 *Xs = lds_stream + BN * wstride;        // [2][BM * kHS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;
    const unsigned g = gridDim.x, bi = blockIdx.x;
    const unsigned blk = (g % 8 == 0) ? (bi % 8) * (g / 8) + bi / 8 : bi;
    const int col0 = (int)(blk % ncb) * BN;
    const int stream = (int)(blk / ncb);
    const __bf16 *wt = reinterpret_cast<const __bf16 *>(p.w);

    // this workgroup's columns of the weights, all of K, once
    {
        const int per_col = ldw / 8;
        for (int e = tid; e < BN * per_col; e += kMlpThreads) {
            const int c = e / per_col, kv = e - c * per_col;
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (__bf16)0.f;
            if (col0 + c < p.cout) v = *reinterpret_cast<const bf16x8 *>(wt + (long)(col0 + c) * ldw + 8 * kv);
            *reinterpret_cast<bf16x8 *>(&Wl[c * wstride + 8 * kv]) = v;
        }
    }

    const int mine = stream < nrb ? (nrb - stream + streams - 1) / streams : 0;   // row blocks of this stream
    const int total = mine * ktiles;
    const int kq = tid & 15, r = tid >> 4;

    // the cursor of the loads runs two steps ahead of the cursor of the products
    int ib = stream, ik = 0;
    // Loads are unconditional so that nothing but the counter orders them (a load under a condition makes the compiler
    // wait for both sets): channels past cin read the row's last four (and are staged as zeros), the two steps past
    // the end of the stream read the last row block again (and are never staged).
    auto issue = [&](float4 (&xr)[8]) {
        const float *src = p.x + ((long)min(ib, nrb - 1) * BM + r) * p.ldx + min(ik * kHK + 4 * kq, p.cin - 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) xr[i] = *reinterpret_cast<const float4 *>(src + (long)(16 * i) * p.ldx);
        if (++ik == ktiles) { ik = 0; ib += streams; }
    };
    auto stage = [&](const float4 (&xr)[8], int buf, int kt) {
        __bf16 *xs = Xs + buf * (BM * kHS);
        const bool live = kt * kHK + 4 * kq < p.cin;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bf16x4 v;
            v[0] = (__bf16)(live ? xr[i].x : 0.f); v[1] = (__bf16)(live ? xr[i].y : 0.f);
            v[2] = (__bf16)(live ? xr[i].z : 0.f); v[3] = (__bf16)(live ? xr[i].w : 0.f);
            *reinterpret_cast<bf16x4 *>(&xs[(r + 16 * i) * kHS + 4 * kq]) = v;
        }
    };

    floatx16 acc[2][2];
    auto clear = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    };
    clear();

    int cb = stream, ck = 0;
    // per-column constants of this workgroup's tile, once
    float sc[2], sh[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int col = col0 + (wn * 2 + b) * 32 + l31;
        sc[b] = (col < p.cout && p.scale) ? p.scale[col] : 1.f;
        sh[b] = (col < p.cout && p.shift) ? p.shift[col] : 0.f;
    }
    StreamBias<GB> gb;
    auto finish = [&]() { stream_finish<GB>(p, acc, gb, sc, sh, (long)cb * BM, col0, wm, wn, l31, half); };
    auto step = [&](float4 (&xr)[8], int s) {
        const int buf = s & 1;
        stage(xr, buf, ck);
        if (ck + 1 == ktiles) stream_bias<GB>(p, gb, (long)cb * BM, col0, wm, wn, l31, half);
        issue(xr);                       // step s + 2 into the registers just emptied
        __syncthreads();                 // image `buf` complete; every wave is done with the products of step s - 1
        const __bf16 *xs = Xs + buf * (BM * kHS) + (wm * 64 + l31) * kHS + 8 * half;
        const __bf16 *ws = Wl + (wn * 64 + l31) * wstride + ck * kHK + 8 * half;
#pragma unroll
        for (int u = 0; u < kHK / 16; ++u) {
            bf16x8 av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = *reinterpret_cast<const bf16x8 *>(xs + a * 32 * kHS + 16 * u);
#pragma unroll
            for (int b = 0; b < 2; ++b) bv[b] = *reinterpret_cast<const bf16x8 *>(ws + b * 32 * wstride + 16 * u);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (++ck == ktiles) {
            finish();
            clear();
            ck = 0;
            cb += streams;
        }
    };

    float4 xa[8], xb[8];
    issue(xa);
    issue(xb);
    // (whole pairs in the loop and the odd step after it: a conditional second step would make the order of the two
    // sets' loads unknown at the top of the loop, and the compiler would then wait for both)
    int s = 0;
    for (; s + 1 < total; s += 2) {
        step(xa, s);
        step(xb, s + 1);
    }
    if (s < total) step(xa, s);
}

// ---------------------------------------------------------------------------
// Few rows (fully connected layers: one row per frustum; the small feature-propagation
// and group-all layers of the PointNet++ variant): 128-row tiles would leave most of
// the chip idle and a K loop through LDS with a barrier per step leaves each
// workgroup waiting on memory latency.  Here a workgroup owns a 32-row x 64-column
// output tile and its four waves SPLIT K: wave w multiplies the w-th quarter of the
// input channels, with both MFMA operands loaded straight from memory into registers
// (activations: 16 bytes per lane, 4 consecutive channels of its row; weights: one dword
// per lane and MFMA, a row of 32 consecutive columns) through a register ring three or four
// K blocks deep - no LDS, no barrier inside the K loop.  The four partial tiles are then
// summed through LDS in a fixed order (deterministic) and the usual epilogue runs on the
// sum.  32 x 64 tiles give 256 workgroups for a 1024 x 512 layer.
//   k order inside a block of 8 channels: lane half h takes channels 8j + 4h + i in
//   step i, so a lane's four activation values are contiguous.
// MODE kFlipoutTwoPass: a second accumulator takes (x * s_in) . dW; the epilogue adds it
// times s_out.
// ---------------------------------------------------------------------------
constexpr int kFrRows = 32, kFrCols = 64;

template <int MODE>
__global__ __launch_bounds__(kMlpThreads) void fewrow_mlp_kernel(const MlpParams pin)
{
    static_assert(MODE == kPlain || MODE == kFlipoutTwoPass, "plain or two-product Flipout");
    constexpr bool FLIP = MODE == kFlipoutTwoPass;
    // K blocks in flight per wave: four (of 12 registers) for the plain product - whole blocks per wave come in
    // multiples of four for the layer widths met here, what is left over runs one by one - three (of 24) with
    // Flipout's second product (measured: 62 / 67 / 72 us for v1's layers with 4 / 3 / 5, 298 / 290 / 310 with Flipout)
    constexpr int kFrRing = FLIP ? 3 : 4;
    MlpParams p = pin;
    if (FLIP) {
        const long s = blockIdx.y;
        p.x += s * pin.sx;
        p.w_delta += s * pin.swd;
        p.sign_in += s * pin.ssi;
        p.sign_out += s * pin.sso;
        if (p.group_bias) p.group_bias += s * pin.sgb;
        p.y += s * pin.sy;
    }
#ifdef FPN_FEWROW_STAMPS     // timing-only build: cycle stamps of one wave into the output (wrong results)
    long long stamp[6];
    stamp[0] = clock64();
#define FPN_FRSTAMP(i) stamp[i] = clock64()
#else
#define FPN_FRSTAMP(i)
#endif
    __shared__ float part[4 * kFrRows * (kFrCols + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (a scalar: the K blocks' offsets are)
    const int l31 = lane & 31, h = lane >> 5;
    const int ncb = (p.cout + kFrCols - 1) / kFrCols;
    const long row0 = (long)(blockIdx.x / ncb) * kFrRows;
    const int col0 = (int)(blockIdx.x % ncb) * kFrCols;
    const int cin = p.cin, cout = p.cout;

    // this wave's share of the input channels, in blocks of 8
    const int nblk = (cin + 7) >> 3;
    const int per = (nblk + 3) >> 2;
    const int b0 = wave * per, b1 = min(nblk, b0 + per);

    const long row = min(row0 + l31, p.rows - 1);             // (rows beyond the end repeat the last one, never stored)
    const float *xrow = p.x + row * p.ldx;
    const float *srow = FLIP ? p.sign_in + (row / p.rows_per_group) * cin : nullptr;
    const int ca = min(col0 + l31, cout - 1), cb = min(col0 + 32 + l31, cout - 1);
    const bool two = col0 + 32 < cout;                         // (uniform: the tile's second 32 columns exist)
    const bool xvec = (p.ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    const bool svec = FLIP && (cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.sign_in) & 15) == 0);

    // buffer descriptors (raw, 32-bit elements, the window as long as an offset can be) and the lane's byte
    // offsets inside a block; the host takes this kernel for layers whose weights span less than 2 GiB only
    const unsigned kRawFlags = 0x00020000;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.w), 0, 0x7fffffff, kRawFlags);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(FLIP ? p.w_delta : p.w), 0, 0x7fffffff, kRawFlags);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.x + row0 * p.ldx), 0, 0x7fffffff, kRawFlags);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(FLIP ? p.sign_in : p.x), 0, 0x7fffffff, kRawFlags);
    int va[4], vb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        va[i] = ((4 * h + i) * cout + ca) * 4;
        vb[i] = ((4 * h + i) * cout + cb) * 4;
    }
    const int vx = (int)((row - row0) * p.ldx + 4 * h) * 4;
    const int vs = FLIP ? (int)((row / p.rows_per_group) * cin + 4 * h) * 4 : 0;
    // the epilogue's per-column constants, requested now (unconditional loads - a weight stands in where a layer has
    // no scale or shift): read where they are used, they were a trip to memory at the kernel's end
    const int col_early = min(col0 + (tid & 63), cout - 1);
    const float sc_early = (p.scale ? p.scale : p.w)[col_early], sh_early = (p.shift ? p.shift : p.w)[col_early];
    struct Stage { float x[4], s[4], wa[4], wb[4], da[4], db[4]; };
    Stage ring[kFrRing];
    // (weight indices in 32 bits - the host takes this kernel for layers below 2^31 weights only: as 64-bit products
    // the addresses were most of the kernel's vector instructions, and beside an fp32 MFMA every one of them costs)
    // a whole block (8 j + 8 <= cin, wave-uniform): no address arithmetic on the vector ALU - per-lane byte offsets
    // made once, the block's offset a scalar; beside an fp32 MFMA every vector instruction costs.  Kept apart from
    // the ragged block's loads: with both behind one branch inside the ring, the compiler put the ragged path's
    // wait for its loads where the paths join, and every stage waited for the loads it had just requested
    auto load_full = [&](Stage &st, int j) {
        const int sw = 8 * j * cout * 4, sx = 8 * j * 4;
        if (xvec) {
            const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, vx, sx, 0));
            st.x[0] = v.x; st.x[1] = v.y; st.x[2] = v.z; st.x[3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) st.x[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vx + 4 * i, sx, 0));
        }
        if (FLIP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) st.s[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vs + 4 * i, sx, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            st.wa[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, va[i], sw, 0));
            st.wb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, vb[i], sw, 0));
            if (FLIP) {
                st.da[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, va[i], sw, 0));
                st.db[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rd, vb[i], sw, 0));
            }
        }
    };
    // the block that holds the last channel: zero activations beyond it, clamped weight rows
    auto load_ragged = [&](Stage &st, int j) {
        const int k = 8 * j + 4 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = k + i < cin;
            const int kk = min(k + i, cin - 1);
            st.x[i] = ok ? xrow[kk] : 0.f;
            if (FLIP) st.s[i] = ok ? srow[kk] : 0.f;
            st.wa[i] = p.w[(unsigned)(kk * cout + ca)];
            st.wb[i] = p.w[(unsigned)(kk * cout + cb)];
            if (FLIP) {
                st.da[i] = p.w_delta[(unsigned)(kk * cout + ca)];
                st.db[i] = p.w_delta[(unsigned)(kk * cout + cb)];
            }
        }
    };
    floatx16 acc[2], accd[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; accd[0][r] = 0.f; accd[1][r] = 0.f; }
    auto step = [&](const Stage &st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(st.x[i], st.wa[i], acc[0], 0, 0, 0);
            if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(st.x[i], st.wb[i], acc[1], 0, 0, 0);
        }
        if (FLIP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xs = st.x[i] * st.s[i];
                accd[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs, st.da[i], accd[0], 0, 0, 0);
                if (two) accd[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs, st.db[i], accd[1], 0, 0, 0);
            }
        }
    };
    // ring of kFrRing K blocks; the stages are fenced so that the scheduler keeps a refill
    // behind the use of the stage it overwrites and ahead of the next stage's use
    const int n = b1 - b0;
    const int nfull = max(0, min(b1, cin >> 3) - b0);        // the wave's whole blocks (the ring takes only these)
    const int npipe = (nfull / kFrRing) * kFrRing;
    if (npipe > 0) {
#pragma unroll
        for (int s = 0; s < kFrRing; ++s) load_full(ring[s], b0 + s);
        FPN_FRSTAMP(1);
        __builtin_amdgcn_sched_barrier(0);
        for (int j0 = 0; j0 < npipe; j0 += kFrRing) {
#pragma unroll
            for (int s = 0; s < kFrRing; ++s) {
                step(ring[s]);
                __builtin_amdgcn_sched_barrier(0);
                load_full(ring[s], b0 + min(j0 + s + kFrRing, npipe - 1));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    for (int j = npipe; j < nfull; ++j) {
        load_full(ring[0], b0 + j);
        step(ring[0]);
    }
    for (int j = nfull; j < n; ++j) {
        load_ragged(ring[0], b0 + j);
        step(ring[0]);
    }

    FPN_FRSTAMP(2);
    // partial tiles -> LDS: register r of tile u holds row (r&3) + 8(r>>2) + 4h, column 32u + l31.
    // The Flipout form's second product goes through the SAME buffer after the first has been
    // summed (two more barriers, half the LDS: 33 KB, under the 64 KiB a kernel gets by default).
    constexpr int LD = kFrCols + 1;
    static_assert(sizeof(part) <= 64 * 1024, "the partial tiles must fit the default LDS limit");
    float *mine = part + wave * kFrRows * LD;
    auto put = [&](const floatx16 (&a)[2]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + 32 * u + l31] = a[u][r];
    };
    // sum of the four waves' partial tiles in wave order; a thread owns one column and eight rows
    const int c = tid & 63, col = col0 + c;
    auto sum = [&](float (&v)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rr = (tid >> 6) + 4 * i;
            float t = part[rr * LD + c];
#pragma unroll
            for (int w = 1; w < 4; ++w) t += part[(w * kFrRows + rr) * LD + c];
            v[i] = t;
        }
    };
    float v[8], d[8];
    put(acc);
    FPN_FRSTAMP(3);
    __syncthreads();
    sum(v);
    FPN_FRSTAMP(4);
    if (FLIP) {
        __syncthreads();
        put(accd);
        __syncthreads();
        sum(d);
    }
    if (col >= cout) return;
    const float sc = p.scale ? sc_early : 1.f;
    const float sh = p.shift ? sh_early : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int rr = (tid >> 6) + 4 * i;
        const long rw = row0 + rr;
        if (rw >= p.rows) break;
        float y = v[i];
        if (FLIP) y += d[i] * p.sign_out[(rw / p.rows_per_group) * cout + col];
        if (p.group_bias) y += p.group_bias[(rw / p.rows_per_group) * cout + col];
        y = y * sc + sh;
        if (p.relu) y = fmaxf(y, 0.f);
        p.y[rw * p.ldy + col] = y;
    }
#ifdef FPN_FEWROW_STAMPS
    stamp[5] = clock64();
    __syncthreads();        // (the workgroup's own results are out: the stamps go over the first of them)
    if (blockIdx.x == 0 && tid == 64)
        for (int i = 0; i < 6; ++i) reinterpret_cast<long long *>(p.y)[i] = stamp[i];
#endif
}

// y[g,c] = max over the blocks_per_group partial rows of group g
__global__ void pool_finish_kernel(long groups, int cout, int blocks_per_group,
                                   const float *__restrict__ partial, float *__restrict__ y, int ldy)
{
    const long total = groups * cout;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / cout;
        const int c = (int)(e % cout);
        const float *src = partial + g * blocks_per_group * cout + c;
        float m = src[0];
        for (int u = 1; u < blocks_per_group; ++u) m = fmaxf(m, src[(long)u * cout]);
        y[g * ldy + c] = m;
    }
}

// stand-alone max over groups of pool_rows consecutive rows (any pool_rows):
// one thread per (group, column), columns along the lanes
__global__ void max_pool_rows_kernel(long groups, int c, int pool_rows, const float *__restrict__ x,
                                     int ldx, float *__restrict__ y, int ldy)
{
    const long total = groups * c;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long g = e / c;
        const int col = (int)(e % c);
        const float *src = x + g * pool_rows * ldx + col;
        float m = src[0];
        for (int r = 1; r < pool_rows; ++r) m = fmaxf(m, src[(long)r * ldx]);
        y[g * ldy + col] = m;
    }
}

// samples: Monte-Carlo samples stacked in one launch (grid.y; p.rows is per sample)
template <int MODE>
int launch_mlp(const MlpParams &p, hipStream_t st, unsigned samples = 1)
{
    const long rb = (p.rows + kBM - 1) / kBM;
    const long fr_blocks = ((p.rows + kFrRows - 1) / kFrRows) * ((p.cout + kFrCols - 1) / kFrCols);
    if constexpr (MODE != kFlipoutFold) {
        if (p.pool_rows == 0 && fr_blocks <= 1024 && p.cin >= 32 && (long)p.cin * p.cout < (1L << 28) && p.ldx < (1 << 23) &&
            (MODE != kFlipoutTwoPass || (p.rows / p.rows_per_group + 1) * (long)p.cin < (1L << 28))) {
            // few rows: 32 x 64 tiles, K split over the workgroup's waves
            hipLaunchKernelGGL((fewrow_mlp_kernel<MODE>), dim3((unsigned)fr_blocks, samples), dim3(kMlpThreads), 0, st, p);
            return check_launch("fpn_shared_mlp(few rows)");
        }
    }
    if (p.pool_rows == 0 && p.cout > 64 && rb * ((p.cout + 127) / 128) < 256) {
        // few rows (the fully connected layers: one row per frustum): 128-row tiles
        // would leave most of the chip idle, so use 32-row x 128-column tiles
        const long nblk = ((p.rows + 31) / 32) * ((p.cout + 127) / 128);
        hipLaunchKernelGGL((shared_mlp_kernel<1, 4, 1, 1, MODE>), dim3((unsigned)nblk, samples), dim3(kMlpThreads), 0, st, p);
    } else if (p.cout > 64) {
        const long nblk = rb * ((p.cout + 127) / 128);
        FPN_REQUIRE(nblk <= INT32_MAX, "shared_mlp: grid too large");
        hipLaunchKernelGGL((shared_mlp_kernel<2, 2, 2, 2, MODE>), dim3((unsigned)nblk, samples), dim3(kMlpThreads), 0, st, p);
    } else if (p.cout > 32) {
        hipLaunchKernelGGL((shared_mlp_kernel<4, 1, 1, 2, MODE>), dim3((unsigned)rb, samples), dim3(kMlpThreads), 0, st, p);
    } else {
        hipLaunchKernelGGL((shared_mlp_kernel<4, 1, 1, 1, MODE>), dim3((unsigned)rb, samples), dim3(kMlpThreads), 0, st, p);
    }
    int rc = check_launch("fpn_shared_mlp");
    if (rc != FPN_OK) return rc;
    if (p.pool_rows > kBM) {
        // the samples' partial maxima and outputs lie back to back: one pass over all
        const long groups = p.rows / p.pool_rows * samples;
        const int blocks = (int)std::min<long>((groups * p.cout + 255) / 256, 65536);
        hipLaunchKernelGGL(pool_finish_kernel, dim3(blocks), dim3(256), 0, st, groups, p.cout,
                           p.pool_rows / kBM, p.partial, p.y, p.ldy);
        rc = check_launch("fpn_shared_mlp(pool)");
    }
    return rc;
}

// Scratch of fpn_shared_mlp_static_f32: the per-row-block partial maxima of a layer that
// pools more than 128 rows.  One copy per GPU, held until the process ends.  The array's
// NAME is a host-side handle, not a device address: host code uses the per-device
// address hipGetSymbolAddress returns (static_scratch()), never the name itself.
static __device__ __align__(256) float scratch[FPN_STATIC_SCRATCH_BYTES / sizeof(float)];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_lock;
float *g_scratch_ptr[kMaxDevices] = {nullptr};

// This device's copy of `scratch` (looked up once per device, with the device current).
int static_scratch(float **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(FPN_E_LAUNCH, "shared_mlp_static: hipGetDevice: %s", hipGetErrorString(e));
    FPN_REQUIRE(dev >= 0 && dev < kMaxDevices, "shared_mlp_static: device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(FPN_E_LAUNCH, "shared_mlp_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<float *>(ptr);
    }
    *out = g_scratch_ptr[dev];
    return FPN_OK;
}

// FPN_BF16_MLP=tile keeps the per-tile kernel for every bf16 layer (a measurement switch; the twin of the tests)
inline bool mlp_bf16_stream()
{
    const char *e = std::getenv("FPN_BF16_MLP");
    return !(e && std::string(e) == "tile");
}

int check_common(const char *who, long rows, int cin, int cout, const float *x, int ldx, const float *w,
                 int pool_rows, const float *y, int ldy, const void *workspace, size_t workspace_bytes)
{
    FPN_REQUIRE(rows >= 0 && cin >= 1 && cout >= 1, "%s: bad size rows=%ld cin=%d cout=%d", who, rows, cin, cout);
    FPN_REQUIRE(ldx >= cin && ldy >= cout, "%s: row stride smaller than the channel count", who);
    FPN_REQUIRE(rows == 0 || (x && w && y), "%s: null pointer", who);     // (an empty batch has no buffers)
    if (pool_rows) {
        FPN_REQUIRE(pool_rows > 0 && pool_rows % 32 == 0, "%s: pool_rows=%d must be a positive multiple of 32", who, pool_rows);
        FPN_REQUIRE(rows % pool_rows == 0, "%s: rows=%ld is not a multiple of pool_rows=%d", who, rows, pool_rows);
        if (pool_rows <= kBM) FPN_REQUIRE(kBM % pool_rows == 0, "%s: pool_rows=%d must be 32, 64 or a multiple of 128", who, pool_rows);
        else {
            FPN_REQUIRE(pool_rows % kBM == 0, "%s: pool_rows=%d must be 32, 64 or a multiple of 128", who, pool_rows);
            FPN_REQUIRE(workspace && workspace_bytes >= fpn_shared_mlp_workspace_bytes(rows, cout, pool_rows),
                        "%s: workspace too small (%zu bytes)", who, workspace_bytes);
        }
    }
    return FPN_OK;
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" {

size_t fpn_shared_mlp_workspace_bytes(long rows, int cout, int pool_rows)
{
    if (pool_rows <= kBM || rows <= 0 || cout <= 0) return 0;
    return (size_t)((rows + kBM - 1) / kBM) * (size_t)cout * sizeof(float);
}

int fpn_max_pool_rows(long rows, int c, int pool_rows, const float *x, int ldx, float *y, int ldy,
                      void *stream)
{
    FPN_REQUIRE(rows >= 0 && c >= 1 && pool_rows >= 1, "max_pool_rows: bad size");
    FPN_REQUIRE(rows % pool_rows == 0, "max_pool_rows: rows=%ld is not a multiple of pool_rows=%d", rows, pool_rows);
    FPN_REQUIRE(ldx >= c && ldy >= c, "max_pool_rows: row stride smaller than the channel count");
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(x && y, "max_pool_rows: null pointer");
    const long groups = rows / pool_rows;
    const int blocks = (int)std::min<long>((groups * c + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(max_pool_rows_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), groups, c,
                       pool_rows, x, ldx, y, ldy);
    return check_launch("fpn_max_pool_rows");
}

int fpn_shared_mlp_f32(long rows, int cin, int cout, const float *x, int ldx, const float *w,
                       const float *scale, const float *shift, const float *group_bias,
                       int rows_per_group, int relu, int pool_rows, float *y, int ldy, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    int rc = check_common("shared_mlp", rows, cin, cout, x, ldx, w, pool_rows, y, ldy, workspace, workspace_bytes);
    if (rc != FPN_OK) return rc;
    if (group_bias) FPN_REQUIRE(rows_per_group >= 1, "shared_mlp: group_bias needs rows_per_group >= 1");
    if (rows == 0) return FPN_OK;
    MlpParams p{};
    p.rows = rows; p.cin = cin; p.cout = cout; p.x = x; p.ldx = ldx; p.w = w;
    p.scale = scale; p.shift = shift; p.group_bias = group_bias;
    p.rows_per_group = group_bias ? rows_per_group : 1;
    p.relu = relu; p.pool_rows = pool_rows; p.y = y; p.ldy = ldy;
    p.partial = static_cast<float *>(workspace);
    return launch_mlp<kPlain>(p, as_stream(stream));
}

int fpn_mlp_pack_bf16(int cin, int cout, const float *w, void *wt, void *stream)
{
    FPN_REQUIRE(cin >= 1 && cout >= 1, "mlp_pack_bf16: bad size cin=%d cout=%d", cin, cout);
    FPN_REQUIRE(w && wt, "mlp_pack_bf16: null pointer");
    const int ldw = (cin + kHK - 1) / kHK * kHK;
    const long total = (long)cout * ldw;
    hipLaunchKernelGGL(mlp_pack_bf16_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
                       as_stream(stream), cin, cout, ldw, w, static_cast<__bf16 *>(wt));
    return check_launch("fpn_mlp_pack_bf16");
}

int fpn_shared_mlp_bf16(long rows, int cin, int cout, const float *x, int ldx, const void *wt,
                        const float *scale, const float *shift, const float *group_bias,
                        int rows_per_group, int relu, int pool_rows, float *y, int ldy, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    int rc = check_common("shared_mlp_bf16", rows, cin, cout, x, ldx, static_cast<const float *>(wt), pool_rows, y, ldy,
                          workspace, workspace_bytes);
    if (rc != FPN_OK) return rc;
    if (group_bias) FPN_REQUIRE(rows_per_group >= 1, "shared_mlp_bf16: group_bias needs rows_per_group >= 1");
    FPN_REQUIRE((reinterpret_cast<uintptr_t>(wt) & 15) == 0, "shared_mlp_bf16: wt must be 16-byte aligned");
    if (rows == 0) return FPN_OK;
    MlpParams p{};
    p.rows = rows; p.cin = cin; p.cout = cout; p.x = x; p.ldx = ldx; p.w = static_cast<const float *>(wt);
    p.scale = scale; p.shift = shift; p.group_bias = group_bias;
    p.rows_per_group = group_bias ? rows_per_group : 1;
    p.relu = relu; p.pool_rows = pool_rows; p.y = y; p.ldy = ldy;
    p.partial = static_cast<float *>(workspace);
    hipStream_t st = as_stream(stream);
    const long nblk = ((rows + kBM - 1) / kBM) * ((cout + 127) / 128);
    FPN_REQUIRE(nblk <= INT32_MAX, "shared_mlp_bf16: grid too large");
    {
        // many rows, few input channels, no pooling: the persistent kernel with the weights resident in LDS takes the
        // whole 128-row blocks, the per-tile kernel below a ragged tail
        const int ldw = (cin + kHK - 1) / kHK * kHK, ncb = (cout + 127) / 128;
        const long nrb = rows / kBM, tail = rows - nrb * kBM;
        const size_t lds = ((size_t)128 * (ldw + 8) + 2 * (size_t)kBM * kHS) * sizeof(__bf16);
        const int gbmode = !group_bias ? 0 : (p.rows_per_group % 32 == 0) ? 1 : (p.rows_per_group == 1) ? 2 : 3;
        const bool tail_ok = tail == 0 || gbmode == 0 || (nrb * kBM) % p.rows_per_group == 0;
        int cus = 0;
        if (pool_rows == 0 && nrb >= 1024 && nrb <= INT32_MAX && lds <= 160 * 1024 && cin % 4 == 0 && ldx % 4 == 0 && tail_ok &&
            (reinterpret_cast<uintptr_t>(x) & 15) == 0 && mlp_bf16_stream() && (cus = current_device_cus()) >= 8) {
            // two workgroups a CU when LDS allows (at most 256 registers), one with a bias row per row (its 64 bias
            // registers make 318)
            int streams = cus * (gbmode == 2 ? 1 : (int)std::min<size_t>(2, (160 * 1024) / lds)) / ncb;
            streams -= streams % 8;                        // (whole rounds over the XCDs: the kernel's renumbering)
            if (streams >= 8) {
                const void *fn = gbmode == 0 ? reinterpret_cast<const void *>(&shared_mlp_bf16_stream_kernel<0>)
                               : gbmode == 1 ? reinterpret_cast<const void *>(&shared_mlp_bf16_stream_kernel<1>)
                               : gbmode == 2 ? reinterpret_cast<const void *>(&shared_mlp_bf16_stream_kernel<2>)
                                             : reinterpret_cast<const void *>(&shared_mlp_bf16_stream_kernel<3>);
                if (!allow_dynamic_lds(fn, lds))
                    return fail(FPN_E_LAUNCH, "shared_mlp_bf16: the device refuses %zu bytes of LDS per workgroup", lds);
                MlpParams q = p;
                q.rows = nrb * kBM;
                const dim3 grid((unsigned)(streams * ncb)), block(kMlpThreads);
                switch (gbmode) {
                case 0: hipLaunchKernelGGL(shared_mlp_bf16_stream_kernel<0>, grid, block, lds, st, q, ncb, streams, (int)nrb); break;
                case 1: hipLaunchKernelGGL(shared_mlp_bf16_stream_kernel<1>, grid, block, lds, st, q, ncb, streams, (int)nrb); break;
                case 2: hipLaunchKernelGGL(shared_mlp_bf16_stream_kernel<2>, grid, block, lds, st, q, ncb, streams, (int)nrb); break;
                default: hipLaunchKernelGGL(shared_mlp_bf16_stream_kernel<3>, grid, block, lds, st, q, ncb, streams, (int)nrb); break;
                }
                rc = check_launch("fpn_shared_mlp_bf16(stream)");
                if (rc != FPN_OK || tail == 0) return rc;
                p.rows = tail;
                p.x = x + nrb * kBM * (long)ldx;
                p.y = y + nrb * kBM * (long)ldy;
                if (group_bias) p.group_bias = group_bias + (nrb * kBM / p.rows_per_group) * (long)cout;
                hipLaunchKernelGGL(shared_mlp_bf16_kernel, dim3((unsigned)ncb), dim3(kMlpThreads), 0, st, p);
                return check_launch("fpn_shared_mlp_bf16(tail)");
            }
        }
    }
    hipLaunchKernelGGL(shared_mlp_bf16_kernel, dim3((unsigned)nblk), dim3(kMlpThreads), 0, st, p);
    rc = check_launch("fpn_shared_mlp_bf16");
    if (rc != FPN_OK) return rc;
    if (pool_rows > kBM) {
        const long groups = rows / pool_rows;
        const int blocks = (int)std::min<long>((groups * cout + 255) / 256, 65536);
        hipLaunchKernelGGL(pool_finish_kernel, dim3(blocks), dim3(256), 0, st, groups, cout, pool_rows / kBM, p.partial, y, ldy);
        rc = check_launch("fpn_shared_mlp_bf16(pool)");
    }
    return rc;
}

int fpn_shared_mlp_static_f32(long rows, int cin, int cout, const float *x, int ldx, const float *w,
                              const float *scale, const float *shift, const float *group_bias,
                              int rows_per_group, int relu, int pool_rows, float *y, int ldy, void *stream)
{
    const size_t need = fpn_shared_mlp_workspace_bytes(rows, cout, pool_rows);
    FPN_REQUIRE(need <= (size_t)FPN_STATIC_SCRATCH_BYTES,
                "shared_mlp_static: rows=%ld cout=%d pool_rows=%d needs %zu bytes of scratch, the static buffer holds %zu "
                "(use fpn_shared_mlp_f32 with a workspace)", rows, cout, pool_rows, need, (size_t)FPN_STATIC_SCRATCH_BYTES);
    float *ws = nullptr;
    if (need) {
        int rc = static_scratch(&ws);
        if (rc != FPN_OK) return rc;
    }
    // every slot of the partial-maxima table that the second kernel reads is written by
    // the first one of the same call: nothing has to start at zero
    return fpn_shared_mlp_f32(rows, cin, cout, x, ldx, w, scale, shift, group_bias, rows_per_group, relu,
                              pool_rows, y, ldy, ws, need, stream);
}

int fpn_shared_mlp_flipout_f32(long rows, int cin, int cout, const float *x, int ldx, const float *w_mean,
                               const float *w_delta, long rows_per_sample, const float *sign_in,
                               const float *sign_out, int rows_per_group, const float *group_bias,
                               const float *scale, const float *shift, int relu, int pool_rows, float *y,
                               int ldy, void *workspace, size_t workspace_bytes, void *stream)
{
    int rc = check_common("shared_mlp_flipout", rows, cin, cout, x, ldx, w_mean, pool_rows, y, ldy, workspace, workspace_bytes);
    if (rc != FPN_OK) return rc;
    FPN_REQUIRE(rows == 0 || (w_delta && sign_in && sign_out), "shared_mlp_flipout: null pointer");
    FPN_REQUIRE(rows_per_group >= 1 && rows_per_sample >= 1, "shared_mlp_flipout: rows_per_group / rows_per_sample must be >= 1");
    FPN_REQUIRE(rows % rows_per_sample == 0 && rows_per_sample % rows_per_group == 0,
                "shared_mlp_flipout: rows (%ld) must split into samples (%ld) and samples into groups (%d)",
                rows, rows_per_sample, rows_per_group);
    if (pool_rows) FPN_REQUIRE(rows_per_sample % pool_rows == 0, "shared_mlp_flipout: pool groups straddle samples");
    if (rows == 0) return FPN_OK;
    // the Monte-Carlo samples (each with its own weight perturbation) share launches:
    // grid.y walks the samples, at most 65535 at a time
    const long nsamples = rows / rows_per_sample;
    const bool fold = rows_per_group % kBM == 0;
    const long groups_per_sample = rows_per_sample / rows_per_group;
    const size_t ws_per_sample = fpn_shared_mlp_workspace_bytes(rows_per_sample, cout, pool_rows);
    for (long s = 0; s < nsamples; s += 65535) {
        MlpParams p{};
        p.rows = rows_per_sample; p.cin = cin; p.cout = cout;
        p.x = x + s * rows_per_sample * ldx; p.ldx = ldx;
        p.w = w_mean; p.w_delta = w_delta + s * (long)cin * cout;
        p.sign_in = sign_in + s * groups_per_sample * cin;
        p.sign_out = sign_out + s * groups_per_sample * cout;
        p.scale = scale; p.shift = shift; p.rows_per_group = rows_per_group;
        p.group_bias = group_bias ? group_bias + s * groups_per_sample * cout : nullptr;
        p.relu = relu; p.pool_rows = pool_rows; p.ldy = ldy;
        p.y = y + (pool_rows ? (s * rows_per_sample / pool_rows) : s * rows_per_sample) * (long)ldy;
        p.partial = workspace ? reinterpret_cast<float *>(static_cast<char *>(workspace) + s * ws_per_sample) : nullptr;
        p.sx = rows_per_sample * (long)ldx; p.swd = (long)cin * cout;
        p.ssi = groups_per_sample * cin; p.sso = groups_per_sample * cout; p.sgb = groups_per_sample * cout;
        p.sy = (pool_rows ? rows_per_sample / pool_rows : rows_per_sample) * (long)ldy;
        p.spartial = (long)(ws_per_sample / sizeof(float));
        const unsigned count = (unsigned)std::min<long>(nsamples - s, 65535);
        rc = fold ? launch_mlp<kFlipoutFold>(p, as_stream(stream), count) : launch_mlp<kFlipoutTwoPass>(p, as_stream(stream), count);
        if (rc != FPN_OK) return rc;
    }
    return FPN_OK;
}

}  // extern "C"
