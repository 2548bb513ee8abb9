// Point-set operators of the Frustum-PointNet hot path for gfx950 (MI355X):
// farthest point sampling, gather, ball query, grouping, three-NN, three-point
// interpolation (separate and fused), partial selection sort, and the
// mask -> centroid -> object-point gather step.
//
// This translation unit is compiled with -ffp-contract=off: the squared
// distance is three rounded products summed left to right and the interpolation
// is three rounded products summed left to right, the arithmetic a C compiler
// without FMA contraction gives the reference's CPU loops.  sqrt and division
// are the correctly rounded forms (hipcc default, also passed explicitly).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>

#include "fpn_common.h"

namespace fpn {
namespace {

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sqdist3(float x2, float y2, float z2, float x1, float y1, float z1)
{
    float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    float s = dx * dx + dy * dy;
    return s + dz * dz;
}

// max over the 64 lanes of a wave for values that order as unsigned integers
// (non-negative floats do).  Six DPP steps leave the result in lane 63:
// row_shr 1/2/4/8 fold each 16-lane row into its last lane, row_bcast15 folds
// rows 0->1 and 2->3, row_bcast31 folds the lower half into the upper.  Lanes a
// step does not reach see their own value again (old == src), which is harmless
// for an idempotent max.
__device__ __forceinline__ unsigned wave_umax(unsigned v)
{
#define FPN_DPP_MAX(ctrl, rows)                                                      \
    {                                                                                \
        unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, rows, 0xf, false); \
        v = v > t ? v : t;                                                           \
    }
    FPN_DPP_MAX(0x111, 0xf)  // row_shr:1
    FPN_DPP_MAX(0x112, 0xf)  // row_shr:2
    FPN_DPP_MAX(0x114, 0xf)  // row_shr:4
    FPN_DPP_MAX(0x118, 0xf)  // row_shr:8
    FPN_DPP_MAX(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
    FPN_DPP_MAX(0x143, 0xc)  // row_bcast:31 into rows 2 and 3
#undef FPN_DPP_MAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---------------------------------------------------------------------------
// Farthest point sampling.
//
// One workgroup per frustum.  The frustum's xyz is staged once in LDS (the
// picked point's coordinates are re-read from there every iteration as an LDS
// broadcast); each thread owns PPT CONSECUTIVE points and keeps their
// coordinates and running min-distance in registers for the whole m-1
// iterations.  Because ownership is blocked, "lower lane" and "lower wave" both
// mean "lower point index", so the lowest-index tie-break is: strict > inside a
// thread, the first lane of the ballot inside a wave, strict > over waves in
// wave order.  One barrier per iteration (the wave results are double-buffered
// by iteration parity); single-wave blocks need none.
// ---------------------------------------------------------------------------
template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_kernel(int n, int m, const float *__restrict__ inp,
                                                      int *__restrict__ out)
{
    extern __shared__ float smem[];
    constexpr int W = THREADS / kWave;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *pts = inp + (size_t)blockIdx.x * n * 3;
    int *o = out + (size_t)blockIdx.x * m;

    float *sxyz = smem;
    unsigned *xval = reinterpret_cast<unsigned *>(smem + ((n * 3 + 3) & ~3));
    int *xidx = reinterpret_cast<int *>(xval + 2 * W);

    for (int i = tid; i < n * 3; i += THREADS) sxyz[i] = pts[i];
    __syncthreads();

    // The thread's points are kept as PAIRS: the subtract / multiply / add chain of the
    // squared distance then runs on the packed fp32 instructions (v_pk_add_f32,
    // v_pk_mul_f32: two IEEE-rounded results per instruction, same values as the scalar
    // forms), which is what bounds this kernel once several frustums share a CU.
    constexpr int NP = (PPT + 1) / 2;   // with an odd PPT the last pair's second half is unused
    float2v px[NP], py[NP], pz[NP];
#pragma unroll
    for (int s2 = 0; s2 < NP; ++s2) px[s2] = py[s2] = pz[s2] = float2v{0.f, 0.f};
    float md[PPT];
    const int base = tid * PPT;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        int k = base + s;
        bool valid = k < n;
        px[s >> 1][s & 1] = valid ? sxyz[k * 3 + 0] : 0.f;
        py[s >> 1][s & 1] = valid ? sxyz[k * 3 + 1] : 0.f;
        pz[s >> 1][s & 1] = valid ? sxyz[k * 3 + 2] : 0.f;
        // slots past the end start at distance 0 with an index above every real
        // point: they can tie only when every real point is at 0 too, and then
        // point 0 (lower index) wins.
        md[s] = valid ? 1e38f : 0.f;
    }

    int old = 0;
    if (tid == 0) o[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = sxyz[old * 3 + 0], y1 = sxyz[old * 3 + 1], z1 = sxyz[old * 3 + 2];
        float best = -1.f;
        int besti = 0;
#pragma unroll
        for (int s2 = 0; s2 < NP; ++s2) {
            // (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1), left to right, for two points
            const float2v dx = px[s2] - x1, dy = py[s2] - y1, dz = pz[s2] - z1;
            float2v d = dx * dx + dy * dy;
            d = d + dz * dz;
#pragma unroll
            for (int e = 0; e < 2 && 2 * s2 + e < PPT; ++e) {
                const int s = 2 * s2 + e;
                float d2 = d[e] < md[s] ? d[e] : md[s];
                md[s] = d2;
                if (d2 > best) {
                    best = d2;
                    besti = base + s;
                }
            }
        }
        const unsigned key = __float_as_uint(best);
        const unsigned wmax = wave_umax(key);
        const unsigned long long tie = __ballot(key == wmax);
        const int widx = __builtin_amdgcn_readlane(besti, (int)__builtin_ctzll(tie));
        if (W == 1) {
            old = widx;
        } else {
            const int buf = (j & 1) * W;
            if (lane == 0) {
                xval[buf + wave] = wmax;
                xidx[buf + wave] = widx;
            }
            __syncthreads();
            unsigned bv = xval[buf];
            int bi = xidx[buf];
#pragma unroll
            for (int w = 1; w < W; ++w) {
                unsigned v = xval[buf + w];
                int vi = xidx[buf + w];
                if (v > bv) {
                    bv = v;
                    bi = vi;
                }
            }
            old = bi;
        }
        if (tid == 0) o[j] = old;
    }
}

// FPS for frustums too large for the register-resident kernel: running
// distances live in LDS (n floats), coordinates are re-read from global memory
// (L2-resident) every iteration.  Same ownership and tie rules as above.
constexpr int kFpsLargeThreads = 1024;
__global__ __launch_bounds__(kFpsLargeThreads) void fps_large_kernel(int n, int m,
                                                                     const float *__restrict__ inp,
                                                                     int *__restrict__ out)
{
    extern __shared__ float smem[];
    constexpr int W = kFpsLargeThreads / kWave;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *pts = inp + (size_t)blockIdx.x * n * 3;
    int *o = out + (size_t)blockIdx.x * m;
    float *smd = smem;
    unsigned *xval = reinterpret_cast<unsigned *>(smem + ((n + 3) & ~3));
    int *xidx = reinterpret_cast<int *>(xval + 2 * W);

    const int chunk = (n + kFpsLargeThreads - 1) / kFpsLargeThreads;
    const int k0 = tid * chunk, k1 = min(n, k0 + chunk);
    for (int k = k0; k < k1; ++k) smd[k] = 1e38f;

    int old = 0;
    if (tid == 0) o[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
        float best = 0.f;  // a thread with no points reports (0, index >= n)
        int besti = k0 < n ? k0 : n + tid;
        bool first = true;
        for (int k = k0; k < k1; ++k) {
            float d = sqdist3(pts[k * 3 + 0], pts[k * 3 + 1], pts[k * 3 + 2], x1, y1, z1);
            float t = smd[k];
            float d2 = d < t ? d : t;
            smd[k] = d2;
            if (first || d2 > best) {
                best = d2;
                besti = k;
                first = false;
            }
        }
        const unsigned key = __float_as_uint(best);
        const unsigned wmax = wave_umax(key);
        const unsigned long long tie = __ballot(key == wmax);
        const int widx = __builtin_amdgcn_readlane(besti, (int)__builtin_ctzll(tie));
        const int buf = (j & 1) * W;
        if (lane == 0) {
            xval[buf + wave] = wmax;
            xidx[buf + wave] = widx;
        }
        __syncthreads();
        unsigned bv = xval[buf];
        int bi = xidx[buf];
        for (int w = 1; w < W; ++w) {
            unsigned v = xval[buf + w];
            if (v > bv) {
                bv = v;
                bi = xidx[buf + w];
            }
        }
        old = bi;
        if (tid == 0) o[j] = old;
    }
}

template <int THREADS, int PPT>
int launch_fps(int b, int n, int m, const float *inp, int *out, hipStream_t st)
{
    size_t lds = (size_t)((n * 3 + 3) & ~3) * 4 + (size_t)4 * (THREADS / kWave) * 4;
    hipLaunchKernelGGL((fps_kernel<THREADS, PPT>), dim3(b), dim3(THREADS), lds, st, n, m, inp, out);
    return check_launch("fpn_farthest_point_sample");
}

// ---------------------------------------------------------------------------
// FPS for MANY frustums: one WAVE per frustum, no LDS, no barrier.
//
// The workgroup kernel above is built for latency (256 threads on one frustum); with
// thousands of frustums in flight what counts is vector-ALU instructions per (point,
// iteration) and whether every SIMD always has a wave ready to issue.  Here a lane owns PPL
// consecutive points - coordinates and running minimum distances in registers for the
// whole run, PPL = 32 for 2048 points: 128 registers, three waves per SIMD - and the waves
// of a SIMD belong to different frustums, so one wave's dependent steps (the cross-lane
// maximum, the picked point's coordinates coming back from memory) are covered by the
// others' distance updates.  Per point and iteration: the packed subtract / multiply / add
// chain (4 instructions for the exactly rounded squared distance), one minimum and half a
// maximum - 5.5 against 8, and a per-iteration overhead (reduction, search) that is spread
// over 32 points per lane instead of 8.
//
// Lowest index among equal maxima (the tie rule of the workgroup kernel): the lane maximum
// is a maximum of VALUES only, formed over groups of four positions; after the wave maximum
// (DPP) the first lane L of the ballot holds the lowest-index maximum, and its position is
// found by the scalar unit on lane L's values alone: the first group whose maximum equals
// the wave's (one v_readlane per group), then the first of that group's four positions (the
// group is wave-uniform, so the register is selected through the scalar index mode).
// Measured on this chip (tools/valu_issue_probe.hip, clocks per wave instruction and SIMD
// with several waves resident): v_add_f32 / v_mul_f32 2.4, their packed forms 4.4 (two
// results), v_min / v_max3 / v_cmp of either type 4.2, any scalar ALU instruction 4.4 - and
// scalar instructions overlap the vector ones of other waves only partly.  The earlier form
// of the search (one v_cmp per position for the whole wave, four scalar instructions each to
// collect lane L's bits) cost 34 vector + 128 scalar instructions per iteration; this one
// costs about 16 + 40.
// ---------------------------------------------------------------------------
constexpr int kFpsWaveWaves = 4;      // waves (= frustums) per workgroup, independent of each other

// ZLDS (the 32-point form): x, y and the minima stay in registers (96), the z coordinates live
// in LDS as the lane's PRIVATE array - vector i of a lane at ((i * 64 + lane) * 16 bytes: a
// wave's ds_read_b128 touches consecutive 16-byte slots, no bank conflicts; each lane reads
// only what it wrote itself, so there is no barrier) - which brings the kernel from 138 to
// under 128 registers: FOUR waves per SIMD instead of three, and 4096 frustums are exactly
// one resident wave set of the chip (with three, every fourth wave ran alone at the end).
//
// The picked point's coordinates never come from memory: position s (wave-uniform) selects
// the register through the index mode of the scalar unit (what a uniform dynamic index into
// a register vector compiles to), lane L's value is read with v_readlane, z comes from the
// wave's LDS array.  (Read back from memory at a wave-uniform address - a scalar load - the
// waves spent 40 % of their time waiting for it: sixteen frustums per CU are 384 KiB, far
// more than the scalar cache.)
template <int N>
using fvec = float __attribute__((ext_vector_type(N)));
template <int N>
using uvec = unsigned __attribute__((ext_vector_type(N)));

//
// REG = false (the 8-point form, 512 points: eight waves per SIMD, little work per
// iteration): there the fixed cost of the search and of the register selection outweighs
// what it saves; the position is collected from one compare per position (lane L's bit of
// each ballot, on the scalar unit) and the picked point is read back from memory at a
// wave-uniform address, its latency covered by the other seven waves.
template <int PPL, bool ZLDS, bool REG>
__global__ __launch_bounds__(64 * kFpsWaveWaves, ZLDS ? 4 : 1) void fps_wave_kernel(int b, int n, int m,
                                                                                  const float *__restrict__ inp,
                                                                                  int *__restrict__ out)
{
    static_assert(PPL % 4 == 0 && PPL <= 32, "the position word holds 32 points per lane");
    __shared__ float4v zlds[ZLDS ? kFpsWaveWaves * (PPL / 4) * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int f = blockIdx.x * kFpsWaveWaves + wave;
    if (f >= b) return;          // whole waves leave: nothing below synchronises across waves
    const float *pts = inp + (size_t)f * n * 3;
    int *o = out + (size_t)f * m;
    float4v *zwave = zlds + wave * ((PPL / 4) * 64);       // vector i of lane l at zwave[i * 64 + l]

    // Running minimum distances are kept as BIT PATTERNS: a squared distance is >= +0 or NaN,
    // so unsigned order is float order, a NaN (any sign) or +inf lies above the 1e38 every
    // minimum starts from, and min / max / == on the patterns give what `d < md ? d : md`,
    // `>` and `==` give on the floats - without the canonicalising copies the compiler puts
    // before float minima of loop-carried values.
    fvec<PPL> px, py;
    fvec<ZLDS ? 2 : PPL> pz;
    uvec<PPL> md;
    const int base = lane * PPL;
#pragma unroll
    for (int s4 = 0; s4 < PPL / 4; ++s4) {
        float z[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s = 4 * s4 + e, k = base + s;
            const bool valid = k < n;
            const float *src = pts + (size_t)(valid ? k : 0) * 3;
            px[s] = valid ? src[0] : 0.f;
            py[s] = valid ? src[1] : 0.f;
            z[e] = valid ? src[2] : 0.f;
            if constexpr (!ZLDS) pz[s] = z[e];
            // slots past the end: distance 0 and an index above every real point - they can tie
            // only when every real point is at 0 too, and then point 0 (lower index) wins
            md[s] = valid ? __float_as_uint(1e38f) : 0u;
        }
        if constexpr (ZLDS) zwave[s4 * 64 + lane] = float4v{z[0], z[1], z[2], z[3]};
    }
    // (left to itself the compiler forwards the stored z values to the loop's reads - each lane
    // reads what it wrote - and the array is back in registers: the reads go through a pointer
    // it knows nothing about)
    typedef __attribute__((address_space(3))) const float4v *LdsVec;
    typedef __attribute__((address_space(3))) const float *LdsFloat;
    LdsVec zread = (LdsVec)zwave;
    asm volatile("" : "+v"(zread) : : "memory");

    // the first pick is point 0: lane 0, position 0
    float x1 = pts[0], y1 = pts[1], z1 = pts[2];
    int mine = 0;      // lane (j mod 64) keeps the j-th pick until 64 are stored together
    for (int j = 1; j < m; ++j) {
        if constexpr (ZLDS) asm volatile("" : : : "memory");    // (... nor hoisted out of the loop as invariant)
#pragma unroll
        for (int s4 = 0; s4 < PPL / 4; ++s4) {
            float4v zq;
            if constexpr (ZLDS) zq = zread[s4 * 64 + lane];
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int s = 4 * s4 + 2 * e2;
                float2v zz;
                if constexpr (ZLDS) zz = e2 == 0 ? float2v{zq.x, zq.y} : float2v{zq.z, zq.w};
                else zz = float2v{pz[s], pz[s + 1]};
                // (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1), left to right, for two points
                const float2v dx = float2v{px[s], px[s + 1]} - x1, dy = float2v{py[s], py[s + 1]} - y1, dz = zz - z1;
                float2v d = dx * dx + dy * dy;
                d = d + dz * dz;
                md[s] = min(__float_as_uint(d[0]), md[s]);
                md[s + 1] = min(__float_as_uint(d[1]), md[s + 1]);
            }
        }
        unsigned g[PPL / 4];
#pragma unroll
        for (int q = 0; q < PPL / 4; ++q) g[q] = max(max(md[4 * q], md[4 * q + 1]), max(md[4 * q + 2], md[4 * q + 3]));
        unsigned best = g[0];
#pragma unroll
        for (int q = 1; q < PPL / 4; ++q) best = max(best, g[q]);
        const unsigned wmax = wave_umax(best);
        const int L = (int)__builtin_ctzll(__ballot(best == wmax));
        int sp;
        if constexpr (REG) {
            int q0 = 0;
#pragma unroll
            for (int q = PPL / 4 - 1; q >= 0; --q)
                if ((unsigned)__builtin_amdgcn_readlane((int)g[q], L) == wmax) q0 = q;
            sp = 4 * q0;
#pragma unroll
            for (int e = 3; e >= 0; --e)
                if ((unsigned)__builtin_amdgcn_readlane((int)md[4 * q0 + e], L) == wmax) sp = 4 * q0 + e;
        } else {
            unsigned where = 0;
#pragma unroll
            for (int s = 0; s < PPL; ++s) where |= (unsigned)((__ballot(md[s] == wmax) >> L) & 1ull) << s;
            sp = (int)__builtin_ctz(where);
        }
        const int old = L * PPL + sp;
        if constexpr (REG) {
            x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(px[sp]), L));
            y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(py[sp]), L));
            if constexpr (ZLDS) z1 = ((LdsFloat)zread)[((sp >> 2) * 64 + L) * 4 + (sp & 3)];
            else z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pz[sp]), L));
        } else {
            x1 = pts[(size_t)old * 3 + 0];
            y1 = pts[(size_t)old * 3 + 1];
            z1 = pts[(size_t)old * 3 + 2];
        }
        if (lane == (j & 63)) mine = old;
        if ((j & 63) == 63) o[j - 63 + lane] = mine;
    }
    if ((m & 63) != 0 && lane < (m & 63)) o[(m & ~63) + lane] = mine;
}

template <int PPL>
int launch_fps_wave(int b, int n, int m, const float *inp, int *out, hipStream_t st)
{
    hipLaunchKernelGGL((fps_wave_kernel<PPL, PPL == 32, PPL >= 16>), dim3((b + kFpsWaveWaves - 1) / kFpsWaveWaves),
                       dim3(64 * kFpsWaveWaves), 0, st, b, n, m, inp, out);
    return check_launch("fpn_farthest_point_sample");
}

// Which kernel a call takes: the wave-per-frustum form from kFpsWaveMinBatch frustums on
// (at least one wave for every SIMD of the chip) for 257..2048 points; FPN_FPS=block /
// FPN_FPS=wave force one form where it exists (the two are bit-identical: tests use both).
constexpr int kFpsWaveMinBatch = 1024;
inline bool fps_use_wave(int b, int n)
{
    if (n <= 256 || n > 2048) return false;
    const char *e = std::getenv("FPN_FPS");
    if (e && e[0] == 'b') return false;
    if (e && e[0] == 'w') return true;
    return b >= kFpsWaveMinBatch;
}

// ---------------------------------------------------------------------------
// gather_point: one thread per output point.
// ---------------------------------------------------------------------------
__global__ void gather_point_kernel(long total, int n, int m, const float *__restrict__ inp,
                                    const int *__restrict__ idx, float *__restrict__ out)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long)gridDim.x * blockDim.x) {
        long i = e / m;
        int a = idx[e];
        const float *src = inp + (i * n + a) * 3;
        float *dst = out + e * 3;
        dst[0] = src[0];
        dst[1] = src[1];
        dst[2] = src[2];
    }
}

// ---------------------------------------------------------------------------
// Grids (blocks of a frustum, frustums): the workgroups of one frustum read the same
// points, and workgroup ids are handed to the eight XCDs - each with an L2 of its own -
// in turn, x fastest.  Taken as they come, the blocks of a frustum land on all eight and
// every L2 fetches the frustum's points for itself (measured on the fused interpolation:
// 4x the algorithmic read traffic at the memory side).  So launch id i works on logical
// block (i % 8) * (total / 8) + i / 8: consecutive logical blocks - a frustum's - share
// an XCD.  (x, y) = the logical block's position in the launch grid.
// ---------------------------------------------------------------------------
struct BlockXY {
    unsigned x, y;
};

__device__ __forceinline__ BlockXY frustum_major_block()
{
    const unsigned long long gx = gridDim.x, total = gx * gridDim.y;
    unsigned long long i = blockIdx.y * gx + blockIdx.x;
    if (total % 8 == 0) i = (i % 8) * (total / 8) + i / 8;
    return BlockXY{(unsigned)(i % gx), (unsigned)(i / gx)};
}

// ---------------------------------------------------------------------------
// Ball query.
//
// Grid (query blocks, frustums).  A workgroup stages a tile of the frustum's
// dataset points in LDS with coalesced loads, then each WAVE takes one query at
// a time and sweeps the tile 64 points per step: every lane tests one point, a
// ballot gives the hits of the step in index order, a hit's slot is the running
// count plus the number of hits in lower lanes, and the sweep stops (for the
// whole wave) once nsample slots are filled.  That reproduces "first nsample
// points in index order" without any per-lane serial scan, and the row is
// written in runs of consecutive slots.
//
// The accept test max(sqrtf(s),1e-20f) < radius is evaluated as s <= smax,
// where smax (computed on the host with the correctly rounded sqrtf) is the
// largest float whose square root is still below radius; sqrtf is monotonic, so
// the two tests accept exactly the same set of s.  radius <= 1e-20f accepts
// nothing (smax = -1).
// ---------------------------------------------------------------------------
constexpr int kQbThreads = 256;
constexpr int kQbTile = 4096;  // dataset points per LDS tile (48 KiB)
#ifndef FPN_QB_TARGET
#define FPN_QB_TARGET 4096
#endif
constexpr int kQbTargetBlocks = FPN_QB_TARGET;   // workgroups a launch aims for (queries per workgroup follow)

__global__ __launch_bounds__(kQbThreads) void query_ball_kernel(
    int n, int m, float smax, int nsample, int qpb, int tile_pts, const float *__restrict__ xyz1,
    const float *__restrict__ xyz2, int *__restrict__ idx, int *__restrict__ pts_cnt)
{
    extern __shared__ float smem[];
    constexpr int W = kQbThreads / kWave;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BlockXY blk = frustum_major_block();
    const int bi = blk.y;
    const float *data = xyz1 + (size_t)bi * n * 3;
    const float *query = xyz2 + (size_t)bi * m * 3;
    const int q0 = blk.x * qpb, q1 = min(m, q0 + qpb);

    // the tile as three arrays (x | y | z, tile_pts even): a lane tests the points 2 lane and
    // 2 lane + 1 of a 128-point step, read as 8-byte pairs, on the packed fp32 instructions
    // (v_pk_add_f32 / v_pk_mul_f32: two IEEE-rounded results per instruction, the values of the
    // scalar forms) - this kernel is bound by its vector instructions.  (Four points per lane
    // and step were measured too: 3 % faster on 2048-point sets, 30 % slower on 128-point ones.)
    float *tx = smem, *ty = smem + tile_pts, *tz = smem + 2 * tile_pts;
    int *s_cnt = reinterpret_cast<int *>(smem + tile_pts * 3); // qpb
    int *s_first = s_cnt + qpb;                               // qpb

    for (int q = q0 + tid; q < q1; q += kQbThreads) {
        s_cnt[q - q0] = 0;
        s_first[q - q0] = 0;
    }

    for (int t0 = 0; t0 < n; t0 += tile_pts) {
        const int tn = min(tile_pts, n - t0);
        __syncthreads();  // previous tile fully consumed (and state initialised)
        for (int i = tid; i < tn * 3; i += kQbThreads) {
            const int pnt = i / 3;
            smem[(i - 3 * pnt) * tile_pts + pnt] = data[(size_t)t0 * 3 + i];
        }
        __syncthreads();

        for (int q = q0 + wave; q < q1; q += W) {
            int cnt = s_cnt[q - q0];
            if (cnt >= nsample) continue;
            int first = s_first[q - q0];
            const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
            int *row = idx + ((size_t)bi * m + q) * nsample;
            for (int kb = 0; kb < tn && cnt < nsample; kb += 2 * kWave) {
                const int k = kb + 2 * lane;
                bool hit0 = false, hit1 = false;
                if (k < tn) {       // (k + 1 may be one past the tile's points: the arrays are even-sized, the hit is masked)
                    const float2v px = *reinterpret_cast<const float2v *>(tx + k);
                    const float2v py = *reinterpret_cast<const float2v *>(ty + k);
                    const float2v pz = *reinterpret_cast<const float2v *>(tz + k);
                    // (x2-x1)*(x2-x1) + (y2-y1)*(y2-y1) + (z2-z1)*(z2-z1), left to right, as sqdist3
                    const float2v dx = qx - px, dy = qy - py, dz = qz - pz;
                    float2v sq = dx * dx + dy * dy;
                    sq = sq + dz * dz;
                    hit0 = sq[0] <= smax;
                    hit1 = k + 1 < tn && sq[1] <= smax;
                }
                const unsigned long long h0 = __ballot(hit0), h1 = __ballot(hit1);
                if (h0 | h1) {
                    if (cnt == 0) {
                        const int f0 = h0 ? 2 * (int)__builtin_ctzll(h0) : 0x7fffffff;
                        const int f1 = h1 ? 2 * (int)__builtin_ctzll(h1) + 1 : 0x7fffffff;
                        first = t0 + kb + min(f0, f1);
                    }
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const int slot0 = cnt + __popcll(h0 & below) + __popcll(h1 & below);
                    const int slot1 = slot0 + (hit0 ? 1 : 0);
                    if (hit0 && slot0 < nsample) row[slot0] = t0 + k;
                    if (hit1 && slot1 < nsample) row[slot1] = t0 + k + 1;
                    cnt += __popcll(h0) + __popcll(h1);
                }
            }
            if (lane == 0) {
                s_cnt[q - q0] = cnt;
                s_first[q - q0] = first;
            }
        }
    }

    // pad short rows with the first hit (zeros for an empty ball), write counts
    for (int q = q0 + wave; q < q1; q += W) {
        // s_cnt/s_first of query q were written by lane 0 of this same wave:
        // LDS operations of one wave execute in order, no workgroup barrier needed
        __builtin_amdgcn_wave_barrier();
        int cnt = min(s_cnt[q - q0], nsample);
        const int first = s_first[q - q0];
        int *row = idx + ((size_t)bi * m + q) * nsample;
        for (int l = cnt + lane; l < nsample; l += kWave) row[l] = first;
        if (pts_cnt && lane == 0) pts_cnt[(size_t)bi * m + q] = cnt;
    }
}

// Ball query for the NR radii of a multi-scale set abstraction (models/pointnet_util.py:
// 176-178 calls query_ball_point once per radius on the same xyz / new_xyz) in ONE sweep:
// the tile is staged once and every point pair's squared distance is formed once, then
// tested against each radius that still has free slots; per radius the slot order, the
// early stop at nsample hits and the padding are those of query_ball_kernel, so each
// output equals the single-radius launch bit for bit.
constexpr int kQbMaxRadii = 4;
struct QbMulti {
    float smax[kQbMaxRadii];
    int nsample[kQbMaxRadii];
    int *idx[kQbMaxRadii];
    int *pts_cnt[kQbMaxRadii];
};

template <int NR>
__global__ __launch_bounds__(kQbThreads) void query_ball_multi_kernel(
    int n, int m, int qpb, int tile_pts, const float *__restrict__ xyz1, const float *__restrict__ xyz2, const QbMulti p)
{
    extern __shared__ float smem[];
    constexpr int W = kQbThreads / kWave;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const BlockXY blk = frustum_major_block();
    const int bi = blk.y;
    const float *data = xyz1 + (size_t)bi * n * 3;
    const float *query = xyz2 + (size_t)bi * m * 3;
    const int q0 = blk.x * qpb, q1 = min(m, q0 + qpb);
    float *tx = smem, *ty = smem + tile_pts, *tz = smem + 2 * tile_pts;
    int *s_cnt = reinterpret_cast<int *>(smem + tile_pts * 3);   // [NR][qpb]
    int *s_first = s_cnt + NR * qpb;                             // [NR][qpb]

    for (int i = tid; i < NR * qpb; i += kQbThreads) {
        s_cnt[i] = 0;
        s_first[i] = 0;
    }
    for (int t0 = 0; t0 < n; t0 += tile_pts) {
        const int tn = min(tile_pts, n - t0);
        __syncthreads();
        for (int i = tid; i < tn * 3; i += kQbThreads) {
            const int pnt = i / 3;
            smem[(i - 3 * pnt) * tile_pts + pnt] = data[(size_t)t0 * 3 + i];
        }
        __syncthreads();

        for (int q = q0 + wave; q < q1; q += W) {
            int cnt[NR], first[NR];
            bool open = false;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                cnt[r] = s_cnt[r * qpb + q - q0];
                first[r] = s_first[r * qpb + q - q0];
                open = open || cnt[r] < p.nsample[r];
            }
            if (!open) continue;
            const float qx = query[q * 3 + 0], qy = query[q * 3 + 1], qz = query[q * 3 + 2];
            for (int kb = 0; kb < tn && open; kb += 2 * kWave) {
                const int k = kb + 2 * lane;
                float2v sq = {INFINITY, INFINITY};      // beyond the tile: no radius accepts it
                if (k < tn) {
                    const float2v px = *reinterpret_cast<const float2v *>(tx + k);
                    const float2v py = *reinterpret_cast<const float2v *>(ty + k);
                    const float2v pz = *reinterpret_cast<const float2v *>(tz + k);
                    const float2v dx = qx - px, dy = qy - py, dz = qz - pz;
                    sq = dx * dx + dy * dy;
                    sq = sq + dz * dz;
                    if (k + 1 >= tn) sq[1] = INFINITY;
                }
                open = false;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    if (cnt[r] < p.nsample[r]) {        // (wave-uniform)
                        const bool hit0 = sq[0] <= p.smax[r], hit1 = sq[1] <= p.smax[r];
                        const unsigned long long h0 = __ballot(hit0), h1 = __ballot(hit1);
                        if (h0 | h1) {
                            if (cnt[r] == 0) {
                                const int f0 = h0 ? 2 * (int)__builtin_ctzll(h0) : 0x7fffffff;
                                const int f1 = h1 ? 2 * (int)__builtin_ctzll(h1) + 1 : 0x7fffffff;
                                first[r] = t0 + kb + min(f0, f1);
                            }
                            const unsigned long long below = (1ull << lane) - 1ull;
                            const int slot0 = cnt[r] + __popcll(h0 & below) + __popcll(h1 & below);
                            const int slot1 = slot0 + (hit0 ? 1 : 0);
                            int *row = p.idx[r] + ((size_t)bi * m + q) * p.nsample[r];
                            if (hit0 && slot0 < p.nsample[r]) row[slot0] = t0 + k;
                            if (hit1 && slot1 < p.nsample[r]) row[slot1] = t0 + k + 1;
                            cnt[r] += __popcll(h0) + __popcll(h1);
                        }
                        open = open || cnt[r] < p.nsample[r];
                    }
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    s_cnt[r * qpb + q - q0] = cnt[r];
                    s_first[r * qpb + q - q0] = first[r];
                }
            }
        }
    }

    for (int q = q0 + wave; q < q1; q += W) {
        __builtin_amdgcn_wave_barrier();       // (written by lane 0 of this same wave: LDS operations of a wave are in order)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int cnt = min(s_cnt[r * qpb + q - q0], p.nsample[r]);
            const int first = s_first[r * qpb + q - q0];
            int *row = p.idx[r] + ((size_t)bi * m + q) * p.nsample[r];
            for (int l = cnt + lane; l < p.nsample[r]; l += kWave) row[l] = first;
            if (p.pts_cnt[r] && lane == 0) p.pts_cnt[r][(size_t)bi * m + q] = cnt;
        }
    }
}

// ---------------------------------------------------------------------------
// group_point / three_interpolate share one thread mapping: LW lanes walk the
// channels of a row (LW = power of two covering min(c,64)), 256/LW rows per
// workgroup, so stores are contiguous runs of c floats per row and consecutive
// rows are adjacent in memory.  No integer division by c.
// ---------------------------------------------------------------------------
constexpr int kRowThreads = 256;

template <int LW>
__global__ __launch_bounds__(kRowThreads) void group_point_kernel(
    int b, int n, int c, int rows_per_batch, const float *__restrict__ points,
    const int *__restrict__ idx, float *__restrict__ out)
{
    constexpr int RPB = kRowThreads / LW;
    const int lx = threadIdx.x % LW, ly = threadIdx.x / LW;
    const BlockXY blk = frustum_major_block();
    for (int bi = blk.y; bi < b; bi += gridDim.y) {
        const float *p = points + (size_t)bi * n * c;
        for (int r = blk.x * RPB + ly; r < rows_per_batch; r += gridDim.x * RPB) {
            const size_t row = (size_t)bi * rows_per_batch + r;
            const int ii = idx[row];
            const float *src = p + (size_t)ii * c;
            float *dst = out + row * c;
            for (int l = lx; l < c; l += LW) dst[l] = src[l];
        }
    }
}

// the same copy in 16-byte groups (c a multiple of 4, 16-byte aligned arrays): c4 = c / 4
// groups per row, LW lanes along them
template <int LW>
__global__ __launch_bounds__(kRowThreads) void group_point_vec_kernel(
    int b, int n, int c4, int rows_per_batch, const float4 *__restrict__ points,
    const int *__restrict__ idx, float4 *__restrict__ out)
{
    constexpr int RPB = kRowThreads / LW;
    const int lx = threadIdx.x % LW, ly = threadIdx.x / LW;
    const BlockXY blk = frustum_major_block();
    for (int bi = blk.y; bi < b; bi += gridDim.y) {
        const float4 *p = points + (size_t)bi * n * c4;
        for (int r = blk.x * RPB + ly; r < rows_per_batch; r += gridDim.x * RPB) {
            const size_t row = (size_t)bi * rows_per_batch + r;
            const int ii = idx[row];
            const float4 *src = p + (size_t)ii * c4;
            float4 *dst = out + row * c4;
            for (int l = lx; l < c4; l += LW) dst[l] = src[l];
        }
    }
}

template <int LW>
__global__ __launch_bounds__(kRowThreads) void three_interpolate_kernel(
    int b, int m, int c, int n, int vec, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out)
{
    constexpr int RPB = kRowThreads / LW;
    const int lx = threadIdx.x % LW, ly = threadIdx.x / LW;
    const BlockXY blk = frustum_major_block();
    for (int bi = blk.y; bi < b; bi += gridDim.y) {
        const float *p = points + (size_t)bi * m * c;
        for (int r = blk.x * RPB + ly; r < n; r += gridDim.x * RPB) {
            const size_t row = (size_t)bi * n + r;
            const int i1 = idx[row * 3 + 0], i2 = idx[row * 3 + 1], i3 = idx[row * 3 + 2];
            const float w1 = weight[row * 3 + 0], w2 = weight[row * 3 + 1], w3 = weight[row * 3 + 2];
            const float *p1 = p + (size_t)i1 * c, *p2 = p + (size_t)i2 * c, *p3 = p + (size_t)i3 * c;
            float *dst = out + row * c;
            if (vec) {   // c % 4 == 0 and 16-byte aligned arrays: the same sums, four channels per lane
                const float4 *q1 = reinterpret_cast<const float4 *>(p1), *q2 = reinterpret_cast<const float4 *>(p2),
                             *q3 = reinterpret_cast<const float4 *>(p3);
                float4 *d4 = reinterpret_cast<float4 *>(dst);
                for (int l = lx; l < c / 4; l += LW) {
                    const float4 a = q1[l], bq = q2[l], cq = q3[l];
                    float4 t;
                    t.x = a.x * w1 + bq.x * w2; t.y = a.y * w1 + bq.y * w2;
                    t.z = a.z * w1 + bq.z * w2; t.w = a.w * w1 + bq.w * w2;
                    t.x = t.x + cq.x * w3; t.y = t.y + cq.y * w3; t.z = t.z + cq.z * w3; t.w = t.w + cq.w * w3;
                    d4[l] = t;
                }
            } else {
                for (int l = lx; l < c; l += LW) {
                    float s = p1[l] * w1 + p2[l] * w2;
                    dst[l] = s + p3[l] * w3;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// three_nn: one lane per query point; the known points are staged in LDS as
// float4 tiles and read by every lane as a broadcast, in index order, through
// the reference's strict-"<" insertion cascade (branch-free).  Running bests
// start at +inf: a float distance is never "<" +inf only when it is +inf or
// NaN, the same inputs the reference's double 1e40 start rejects.
// ---------------------------------------------------------------------------
constexpr int kNnThreads = 256;
constexpr int kNnTile = 2048;  // known points per LDS tile (32 KiB as float4)

struct Best3 {
    float d1, d2, d3;
    int i1, i2, i3;
};

__device__ __forceinline__ void best3_init(Best3 &s)
{
    s.d1 = s.d2 = s.d3 = INFINITY;
    s.i1 = s.i2 = s.i3 = 0;
}

__device__ __forceinline__ void best3_push(Best3 &s, float d, int k)
{
    const bool c1 = d < s.d1, c2 = d < s.d2, c3 = d < s.d3;
    s.d3 = c2 ? s.d2 : (c3 ? d : s.d3);
    s.i3 = c2 ? s.i2 : (c3 ? k : s.i3);
    s.d2 = c1 ? s.d1 : (c2 ? d : s.d2);
    s.i2 = c1 ? s.i1 : (c2 ? k : s.i2);
    s.d1 = c1 ? d : s.d1;
    s.i1 = c1 ? k : s.i1;
}

__device__ __forceinline__ void three_nn_scan(Best3 &s, bool active, float x1, float y1, float z1,
                                              const float *__restrict__ known, int m, float4 *tile)
{
    for (int t0 = 0; t0 < m; t0 += kNnTile) {
        const int tn = min(kNnTile, m - t0);
        __syncthreads();
        for (int i = threadIdx.x; i < tn; i += kNnThreads) {
            const float *p = known + (size_t)(t0 + i) * 3;
            tile[i] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        if (active) {
#pragma unroll 4
            for (int k = 0; k < tn; ++k) {
                const float4 p = tile[k];
                best3_push(s, sqdist3(p.x, p.y, p.z, x1, y1, z1), t0 + k);
            }
        }
    }
}

__global__ __launch_bounds__(kNnThreads) void three_nn_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    float *__restrict__ dist, int *__restrict__ idx)
{
    __shared__ float4 tile[kNnTile];
    const BlockXY blk = frustum_major_block();
    const int bi = blk.y;
    const int j = blk.x * kNnThreads + threadIdx.x;
    const bool active = j < n;
    const size_t row = (size_t)bi * n + (active ? j : 0);
    const float x1 = xyz1[row * 3 + 0], y1 = xyz1[row * 3 + 1], z1 = xyz1[row * 3 + 2];
    Best3 s;
    best3_init(s);
    three_nn_scan(s, active, x1, y1, z1, xyz2 + (size_t)bi * m * 3, m, tile);
    if (active) {
        dist[row * 3 + 0] = s.d1;
        dist[row * 3 + 1] = s.d2;
        dist[row * 3 + 2] = s.d3;
        idx[row * 3 + 0] = s.i1;
        idx[row * 3 + 1] = s.i2;
        idx[row * 3 + 2] = s.i3;
    }
}

// Fused three_nn + inverse-distance weights + three_interpolate.  Phase 1 is
// the kernel above; each lane then turns its three distances into weights
// (max(d,1e-10), reciprocal, left-to-right sum, divide: the op sequence of
// pointnet_fp_module) and parks (indices, weights) in LDS; phase 2 has the
// whole workgroup write the 256 output rows with lanes along the channel axis.
// OutT = float, or __bf16: the same sums rounded to bf16 on the way out (round to nearest even,
// the conversion a bf16 consumer would apply to the fp32 result itself).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store4(float *dst, int l, const float4 &t) { reinterpret_cast<float4 *>(dst)[l] = t; }
__device__ __forceinline__ void store4(__bf16 *dst, int l, const float4 &t)
{
    bf16x4 v;
    v[0] = (__bf16)t.x; v[1] = (__bf16)t.y; v[2] = (__bf16)t.z; v[3] = (__bf16)t.w;
    reinterpret_cast<bf16x4 *>(dst)[l] = v;
}

template <int LW, class OutT>
__device__ __forceinline__ void three_nn_interpolate_body(
    int n, int m, int c, int vec, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ points2, OutT *__restrict__ out, float *__restrict__ dist,
    int *__restrict__ idx)
{
    __shared__ float4 tile[kNnTile];
    __shared__ int s_i[kNnThreads * 3];
    __shared__ float s_w[kNnThreads * 3];
    const BlockXY blk = frustum_major_block();
    const int bi = blk.y;
    const int j0 = blk.x * kNnThreads;
    const int j = j0 + threadIdx.x;
    const bool active = j < n;
    const size_t row = (size_t)bi * n + (active ? j : 0);
    const float x1 = xyz1[row * 3 + 0], y1 = xyz1[row * 3 + 1], z1 = xyz1[row * 3 + 2];
    Best3 s;
    best3_init(s);
    three_nn_scan(s, active, x1, y1, z1, xyz2 + (size_t)bi * m * 3, m, tile);

    if (active) {
        if (dist) {
            dist[row * 3 + 0] = s.d1;
            dist[row * 3 + 1] = s.d2;
            dist[row * 3 + 2] = s.d3;
        }
        if (idx) {
            idx[row * 3 + 0] = s.i1;
            idx[row * 3 + 1] = s.i2;
            idx[row * 3 + 2] = s.i3;
        }
    }
    const float r1 = 1.0f / fmaxf(s.d1, 1e-10f);
    const float r2 = 1.0f / fmaxf(s.d2, 1e-10f);
    const float r3 = 1.0f / fmaxf(s.d3, 1e-10f);
    float norm = r1 + r2;
    norm = norm + r3;
    s_i[threadIdx.x * 3 + 0] = s.i1;
    s_i[threadIdx.x * 3 + 1] = s.i2;
    s_i[threadIdx.x * 3 + 2] = s.i3;
    s_w[threadIdx.x * 3 + 0] = r1 / norm;
    s_w[threadIdx.x * 3 + 1] = r2 / norm;
    s_w[threadIdx.x * 3 + 2] = r3 / norm;
    __syncthreads();

    constexpr int RPB = kNnThreads / LW;
    const int lx = threadIdx.x % LW, ly = threadIdx.x / LW;
    const float *p = points2 + (size_t)bi * m * c;
    const int rows = min(kNnThreads, n - j0);
    for (int r = ly; r < rows; r += RPB) {
        const int i1 = s_i[r * 3 + 0], i2 = s_i[r * 3 + 1], i3 = s_i[r * 3 + 2];
        const float w1 = s_w[r * 3 + 0], w2 = s_w[r * 3 + 1], w3 = s_w[r * 3 + 2];
        const float *p1 = p + (size_t)i1 * c, *p2 = p + (size_t)i2 * c, *p3 = p + (size_t)i3 * c;
        OutT *dst = out + ((size_t)bi * n + j0 + r) * c;
        if (vec) {       // c % 4 == 0 and 16-byte aligned arrays: the same sums, four channels per lane
            const float4 *q1 = reinterpret_cast<const float4 *>(p1), *q2 = reinterpret_cast<const float4 *>(p2),
                         *q3 = reinterpret_cast<const float4 *>(p3);
            for (int l = lx; l < c / 4; l += LW) {
                const float4 a = q1[l], bq = q2[l], cq = q3[l];
                float4 t;
                t.x = a.x * w1 + bq.x * w2; t.y = a.y * w1 + bq.y * w2;
                t.z = a.z * w1 + bq.z * w2; t.w = a.w * w1 + bq.w * w2;
                t.x = t.x + cq.x * w3; t.y = t.y + cq.y * w3; t.z = t.z + cq.z * w3; t.w = t.w + cq.w * w3;
                store4(dst, l, t);
            }
        } else {
            for (int l = lx; l < c; l += LW) {
                float t = p1[l] * w1 + p2[l] * w2;
                dst[l] = (OutT)(t + p3[l] * w3);
            }
        }
    }
}

template <int LW>
__global__ __launch_bounds__(kNnThreads) void three_nn_interpolate_kernel(
    int n, int m, int c, int vec, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ points2, float *__restrict__ out, float *__restrict__ dist,
    int *__restrict__ idx)
{
    three_nn_interpolate_body<LW>(n, m, c, vec, xyz1, xyz2, points2, out, dist, idx);
}

template <int LW>
__global__ __launch_bounds__(kNnThreads) void three_nn_interpolate_bf16_kernel(
    int n, int m, int c, int vec, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ points2, __bf16 *__restrict__ out)
{
    three_nn_interpolate_body<LW>(n, m, c, vec, xyz1, xyz2, points2, out, static_cast<float *>(nullptr),
                                  static_cast<int *>(nullptr));
}

// ---------------------------------------------------------------------------
// Partial selection sort: one wave per (b,m) row, the row (values + indices)
// resident in LDS.  Each of the k steps is a wave-wide arg-min over the tail
// (lowest position on ties, like the reference's strict "<" scan) followed by
// the reference's swap, so the WHOLE output row - not only its first k columns
// - equals the reference's.
//
// KNN = true is knn_point (models/tf_ops/grouping/tf_grouping.py:49-74) as ONE kernel: the
// row is not read from a (b,m,n) distance matrix but formed in LDS from the coordinates -
// sum over the c channels, left to right, of the rounded squares of (xyz1 - xyz2), the
// float32 value of tf.reduce_sum((xyz1 - xyz2)**2, -1) for the three-channel case - the same
// k swap steps run on it, and only the first k columns leave (:70-71 slices them): neither
// the matrix nor the two (b,m,n) output arrays exist.
// ---------------------------------------------------------------------------
template <bool KNN>
__global__ __launch_bounds__(kWave) void selection_sort_kernel(int n, int k,
                                                               const float *__restrict__ dist,
                                                               int *__restrict__ outi,
                                                               float *__restrict__ out, int m, int c,
                                                               const float *__restrict__ xyz1,
                                                               const float *__restrict__ xyz2)
{
    extern __shared__ float smem[];
    float *v = smem;
    int *ix = reinterpret_cast<int *>(smem + n);
    const int lane = threadIdx.x;
    const size_t row = blockIdx.x;
    if constexpr (KNN) {
        const float *q = xyz2 + row * c;                       // row = frustum * m + query
        const float *pts = xyz1 + (row / m) * (size_t)n * c;
        for (int s = lane; s < n; s += kWave) {
            float acc = 0.f;
            for (int ch = 0; ch < c; ++ch) {
                const float d = pts[(size_t)s * c + ch] - q[ch];
                const float sq = d * d;
                acc = ch == 0 ? sq : acc + sq;
            }
            v[s] = acc;
            ix[s] = s;
        }
    } else {
        for (int s = lane; s < n; s += kWave) {
            v[s] = dist[row * n + s];
            ix[s] = s;
        }
    }
    __syncthreads();
    const int steps = min(k, n);
    for (int s = 0; s < steps; ++s) {
        // lane-local minimum over t = s+lane, s+lane+64, ...: first strict min
        float bv = INFINITY;
        int bt = n;  // n = "nothing found" (also covers NaN rows: nothing is < NaN)
        bool any = false;
        for (int t = s + lane; t < n; t += kWave) {
            float x = v[t];
            if (!any || x < bv) {
                // a later t replaces only on strict "<"; the first t seeds
                bv = x;
                bt = t;
                any = true;
            }
        }
        // wave arg-min with lowest position on ties.  The reference compares
        // p[t] < p[min] sequentially starting from min = s; a sequential scan
        // with strict "<" returns the lowest position among the smallest
        // values provided comparisons are a total order (no NaN).
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            float ov = __shfl_xor(bv, off);
            int ot = __shfl_xor(bt, off);
            bool oany = ot < n;
            bool take = oany && (!(bt < n) || ov < bv || (ov == bv && ot < bt));
            bv = take ? ov : bv;
            bt = take ? ot : bt;
        }
        if (lane == 0 && bt < n && bt != s) {
            float tv = v[bt];
            v[bt] = v[s];
            v[s] = tv;
            int ti = ix[bt];
            ix[bt] = ix[s];
            ix[s] = ti;
        }
        __syncthreads();
    }
    if constexpr (KNN) {
        // k > n: the remaining columns hold the unsorted tail, as the slice of the reference's arrays would
        for (int s = lane; s < k; s += kWave) {
            out[row * k + s] = s < n ? v[s] : 0.f;
            outi[row * k + s] = s < n ? ix[s] : 0;
        }
    } else {
        for (int s = lane; s < n; s += kWave) {
            out[row * n + s] = v[s];
            outi[row * n + s] = ix[s];
        }
    }
}

// ---------------------------------------------------------------------------
// Mask -> centroid -> object-point selection -> centred gather, one workgroup
// per frustum.  Each thread owns a contiguous chunk of points, so an exclusive
// scan of the per-thread positive counts gives every positive its rank in index
// order; ranks land in an LDS list from which the npick selections are read.
// ---------------------------------------------------------------------------
constexpr int kMgThreads = 256;
constexpr int kMgMaxN = 16384;

__global__ __launch_bounds__(kMgThreads) void mask_gather_kernel(
    int n, int cpc, int npick, int spread, const float *__restrict__ pc, const float *__restrict__ logits,
    float *__restrict__ mask, float *__restrict__ mean, int *__restrict__ sel,
    float *__restrict__ obj)
{
    extern __shared__ int s_pos[];  // n entries: positives in index order
    __shared__ int s_cnt[kMgThreads];
    __shared__ float s_sum[kMgThreads][3];
    __shared__ float s_mean[3];
    __shared__ int s_total;
    const int tid = threadIdx.x;
    const int bi = blockIdx.x;
    const float *p = pc + (size_t)bi * n * cpc;
    const float *lg = logits + (size_t)bi * n * 2;
    const int chunk = (n + kMgThreads - 1) / kMgThreads;
    const int k0 = min(n, tid * chunk), k1 = min(n, k0 + chunk);

    int cnt = 0;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = k0; k < k1; ++k) {
        const bool pos = lg[k * 2 + 0] < lg[k * 2 + 1];
        if (mask) mask[(size_t)bi * n + k] = pos ? 1.f : 0.f;
        if (pos) {
            ++cnt;
            sx += p[(size_t)k * cpc + 0];
            sy += p[(size_t)k * cpc + 1];
            sz += p[(size_t)k * cpc + 2];
        }
    }
    s_cnt[tid] = cnt;
    s_sum[tid][0] = sx;
    s_sum[tid][1] = sy;
    s_sum[tid][2] = sz;
    __syncthreads();
    // exclusive scan of the 256 counts and the coordinate sums by wave 0
    if (tid < kWave) {
        int c4[4];
        int local = 0;
        float ax = 0.f, ay = 0.f, az = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c4[u] = s_cnt[tid * 4 + u];
            local += c4[u];
            ax += s_sum[tid * 4 + u][0];
            ay += s_sum[tid * 4 + u][1];
            az += s_sum[tid * 4 + u][2];
        }
        int incl = local;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            int t = __shfl_up(incl, off);
            if (tid >= off) incl += t;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ax += __shfl_xor(ax, off);
            ay += __shfl_xor(ay, off);
            az += __shfl_xor(az, off);
        }
        int run = incl - local;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s_cnt[tid * 4 + u] = run;
            run += c4[u];
        }
        if (tid == kWave - 1) s_total = incl;
        if (tid == 0) {
            s_sum[0][0] = ax;
            s_sum[0][1] = ay;
            s_sum[0][2] = az;
        }
    }
    __syncthreads();
    const int total = s_total;
    if (tid < 3) {
        const float denom = total > 0 ? (float)total : 1.f;
        const float mu = s_sum[0][tid] / denom;
        s_mean[tid] = mu;
        mean[(size_t)bi * 3 + tid] = mu;
    }
    int rank = s_cnt[tid];
    for (int k = k0; k < k1; ++k)
        if (lg[k * 2 + 0] < lg[k * 2 + 1]) s_pos[rank++] = k;
    __syncthreads();
    const float mx = s_mean[0], my = s_mean[1], mz = s_mean[2];
    for (int t = tid; t < npick; t += kMgThreads) {
        // more positives than slots: the first npick of them, or (spread) every
        // total/npick-th one, which covers the whole object whatever the point order;
        // fewer: all of them, repeated cyclically
        int k = 0;
        if (total > 0) k = s_pos[total >= npick ? (spread ? (int)((long)t * total / npick) : t) : t % total];
        if (sel) sel[(size_t)bi * npick + t] = k;
        float *dst = obj + ((size_t)bi * npick + t) * 3;
        dst[0] = p[(size_t)k * cpc + 0] - mx;
        dst[1] = p[(size_t)k * cpc + 1] - my;
        dst[2] = p[(size_t)k * cpc + 2] - mz;
    }
}

int pow2_at_least(int c)
{
    int lw = 1;
    while (lw < c && lw < kWave) lw <<= 1;
    return lw;
}

// Largest float s with max(sqrtf(s),1e-20f) < radius, or -1 when no s passes.
float ball_threshold(float radius)
{
    if (!(radius > 1e-20f)) return -1.f;
    if (std::isinf(radius) || radius > sqrtf(FLT_MAX)) return FLT_MAX;
    float s = radius * radius;
    if (std::isinf(s)) s = FLT_MAX;
    while (s > 0.f && !(sqrtf(s) < radius)) s = nextafterf(s, -INFINITY);
    for (;;) {
        float up = nextafterf(s, INFINITY);
        if (std::isinf(up) || !(sqrtf(up) < radius)) break;
        s = up;
    }
    return s;
}

}  // namespace
}  // namespace fpn

using namespace fpn;

extern "C" {

// Exposed for the host-side test of the sqrt-free accept test (not part of the
// operator surface; declared in tests only).
float fpn_ball_threshold(float radius) { return ball_threshold(radius); }

int fpn_farthest_point_sample(int b, int n, int m, const float *inp, int *out, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "farthest_point_sample: negative size (b=%d n=%d m=%d)", b, n, m);
    if (b == 0 || m == 0) return FPN_OK;
    FPN_REQUIRE(n >= 1, "farthest_point_sample: npoint > 0 needs at least one input point");
    FPN_REQUIRE(inp && out, "farthest_point_sample: null pointer");
    hipStream_t st = as_stream(stream);
    if (fps_use_wave(b, n)) {
        if (n <= 512) return launch_fps_wave<8>(b, n, m, inp, out, st);
        if (n <= 1024) return launch_fps_wave<16>(b, n, m, inp, out, st);
        return launch_fps_wave<32>(b, n, m, inp, out, st);
    }
    if (n <= 64) return launch_fps<64, 1>(b, n, m, inp, out, st);
    if (n <= 128) return launch_fps<64, 2>(b, n, m, inp, out, st);
    if (n <= 256) return launch_fps<64, 4>(b, n, m, inp, out, st);
    if (n <= 512) return launch_fps<128, 4>(b, n, m, inp, out, st);
    if (n <= 1024) return launch_fps<256, 4>(b, n, m, inp, out, st);
    if (n <= 2048) return launch_fps<256, 8>(b, n, m, inp, out, st);
    if (n <= 4096) return launch_fps<512, 8>(b, n, m, inp, out, st);
    if (n <= 8192) return launch_fps<1024, 8>(b, n, m, inp, out, st);
    if (n > 32768)
        return fail(FPN_E_UNSUPPORTED, "farthest_point_sample: n=%d exceeds the 32768-point limit", n);
    size_t lds = (size_t)((n + 3) & ~3) * 4 + (size_t)4 * (kFpsLargeThreads / kWave) * 4;
    hipLaunchKernelGGL(fps_large_kernel, dim3(b), dim3(kFpsLargeThreads), lds, st, n, m, inp, out);
    return check_launch("fpn_farthest_point_sample");
}

int fpn_gather_point(int b, int n, int m, const float *inp, const int *idx, float *out, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "gather_point: negative size");
    long total = (long)b * m;
    if (total == 0) return FPN_OK;
    FPN_REQUIRE(inp && idx && out, "gather_point: null pointer");
    int blocks = (int)std::min<long>((total + 255) / 256, 65536);
    hipLaunchKernelGGL(gather_point_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), total, n, m,
                       inp, idx, out);
    return check_launch("fpn_gather_point");
}

int fpn_query_ball_point(int b, int n, int m, float radius, int nsample, const float *xyz1,
                         const float *xyz2, int *idx, int *pts_cnt, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "query_ball_point: negative size");
    FPN_REQUIRE(radius > 0.f, "QueryBallPoint expects positive radius");
    FPN_REQUIRE(nsample > 0, "QueryBallPoint expects positive nsample");
    if (b == 0 || m == 0) return FPN_OK;
    FPN_REQUIRE(xyz1 && xyz2 && idx, "query_ball_point: null pointer");
    FPN_REQUIRE(b <= 65535, "query_ball_point: batch %d exceeds 65535", b);
    // queries per workgroup: enough workgroups to cover the chip several times
    // over, but not so few queries that the tile staging dominates
    long want = ((long)b * m + kQbTargetBlocks - 1) / kQbTargetBlocks;
    int qpb = (int)std::min<long>(std::max<long>(want, 4), 64);
    qpb = (qpb + 3) & ~3;
    // LDS for the points actually staged (a 2048-point frustum takes 24 KiB, not the 48 KiB of
    // a full tile): twice the workgroups per CU to cover each other's staging and stores
    const int tile_pts = (std::max(1, std::min(n, kQbTile)) + 1) & ~1;       // (even: the kernel reads point pairs)
    size_t lds = (size_t)tile_pts * 3 * 4 + (size_t)qpb * 2 * 4;
    dim3 grid((m + qpb - 1) / qpb, b);
    hipLaunchKernelGGL(query_ball_kernel, grid, dim3(kQbThreads), lds, as_stream(stream), n, m,
                       ball_threshold(radius), nsample, qpb, tile_pts, xyz1, xyz2, idx, pts_cnt);
    return check_launch("fpn_query_ball_point");
}

int fpn_query_ball_point_multi(int b, int n, int m, int nr, const float *radii, const int *nsamples,
                               const float *xyz1, const float *xyz2, int *const *idx, int *const *pts_cnt,
                               void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "query_ball_point_multi: negative size");
    FPN_REQUIRE(nr >= 1 && nr <= kQbMaxRadii, "query_ball_point_multi: 1 to %d radii (%d)", kQbMaxRadii, nr);
    FPN_REQUIRE(radii && nsamples && idx, "query_ball_point_multi: null pointer");
    QbMulti p{};
    for (int r = 0; r < nr; ++r) {
        FPN_REQUIRE(radii[r] > 0.f, "QueryBallPoint expects positive radius");
        FPN_REQUIRE(nsamples[r] > 0, "QueryBallPoint expects positive nsample");
        p.smax[r] = ball_threshold(radii[r]);
        p.nsample[r] = nsamples[r];
        p.idx[r] = idx[r];
        p.pts_cnt[r] = pts_cnt ? pts_cnt[r] : nullptr;
    }
    if (b == 0 || m == 0) return FPN_OK;
    FPN_REQUIRE(xyz1 && xyz2, "query_ball_point_multi: null pointer");
    for (int r = 0; r < nr; ++r) FPN_REQUIRE(idx[r], "query_ball_point_multi: idx[%d] is null", r);
    FPN_REQUIRE(b <= 65535, "query_ball_point_multi: batch %d exceeds 65535", b);
    long want = ((long)b * m + kQbTargetBlocks - 1) / kQbTargetBlocks;
    int qpb = (int)std::min<long>(std::max<long>(want, 4), 64);
    qpb = (qpb + 3) & ~3;
    const int tile_pts = (std::max(1, std::min(n, kQbTile)) + 1) & ~1;
    size_t lds = (size_t)tile_pts * 3 * 4 + (size_t)qpb * 2 * 4 * nr;
    dim3 grid((m + qpb - 1) / qpb, b);
    hipStream_t st = as_stream(stream);
    switch (nr) {
    case 1: hipLaunchKernelGGL(query_ball_multi_kernel<1>, grid, dim3(kQbThreads), lds, st, n, m, qpb, tile_pts, xyz1, xyz2, p); break;
    case 2: hipLaunchKernelGGL(query_ball_multi_kernel<2>, grid, dim3(kQbThreads), lds, st, n, m, qpb, tile_pts, xyz1, xyz2, p); break;
    case 3: hipLaunchKernelGGL(query_ball_multi_kernel<3>, grid, dim3(kQbThreads), lds, st, n, m, qpb, tile_pts, xyz1, xyz2, p); break;
    default: hipLaunchKernelGGL(query_ball_multi_kernel<4>, grid, dim3(kQbThreads), lds, st, n, m, qpb, tile_pts, xyz1, xyz2, p); break;
    }
    return check_launch("fpn_query_ball_point_multi");
}

#define FPN_DISPATCH_LW(lw, KERNEL, ...)                                             \
    switch (lw) {                                                                    \
    case 1: hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__); break;                     \
    case 2: hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__); break;                     \
    case 4: hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__); break;                     \
    case 8: hipLaunchKernelGGL((KERNEL<8>), __VA_ARGS__); break;                     \
    case 16: hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__); break;                   \
    case 32: hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__); break;                   \
    default: hipLaunchKernelGGL((KERNEL<64>), __VA_ARGS__); break;                   \
    }

int fpn_group_point(int b, int n, int c, int m, int nsample, const float *points, const int *idx,
                    float *out, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0 && nsample >= 0, "group_point: negative size");
    long rows = (long)m * nsample;
    if (b == 0 || rows == 0 || c == 0) return FPN_OK;
    FPN_REQUIRE(points && idx && out, "group_point: null pointer");
    FPN_REQUIRE(rows <= INT32_MAX, "group_point: m*nsample too large");
    hipStream_t st = as_stream(stream);
    if (c % 4 == 0 && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {
        const int lw = pow2_at_least(c / 4);
        const int rpb = kRowThreads / lw;
        dim3 grid((unsigned)std::min<long>((rows + rpb - 1) / rpb, 1 << 20), (unsigned)std::min(b, 65535));
        FPN_DISPATCH_LW(lw, group_point_vec_kernel, grid, dim3(kRowThreads), 0, st, b, n, c / 4, (int)rows,
                        reinterpret_cast<const float4 *>(points), idx, reinterpret_cast<float4 *>(out))
        return check_launch("fpn_group_point");
    }
    const int lw = pow2_at_least(c);
    const int rpb = kRowThreads / lw;
    dim3 grid((unsigned)std::min<long>((rows + rpb - 1) / rpb, 1 << 20), (unsigned)std::min(b, 65535));
    FPN_DISPATCH_LW(lw, group_point_kernel, grid, dim3(kRowThreads), 0, st, b, n, c, (int)rows, points,
                    idx, out)
    return check_launch("fpn_group_point");
}

int fpn_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist, int *idx,
                 void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "three_nn: negative size");
    if (b == 0 || n == 0) return FPN_OK;
    FPN_REQUIRE(xyz1 && dist && idx && (xyz2 || m == 0), "three_nn: null pointer");
    FPN_REQUIRE(b <= 65535, "three_nn: batch %d exceeds 65535", b);
    dim3 grid((n + kNnThreads - 1) / kNnThreads, b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(kNnThreads), 0, as_stream(stream), n, m, xyz1, xyz2,
                       dist, idx);
    return check_launch("fpn_three_nn");
}

int fpn_three_interpolate(int b, int m, int c, int n, const float *points, const int *idx,
                          const float *weight, float *out, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 0, "three_interpolate: negative size");
    if (b == 0 || n == 0 || c == 0) return FPN_OK;
    FPN_REQUIRE(points && idx && weight && out, "three_interpolate: null pointer");
    const int vec = c % 4 == 0 && ((reinterpret_cast<uintptr_t>(points) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int lw = pow2_at_least(vec ? c / 4 : c);
    const int rpb = kRowThreads / lw;
    dim3 grid((unsigned)std::min<long>(((long)n + rpb - 1) / rpb, 1 << 20), (unsigned)std::min(b, 65535));
    hipStream_t st = as_stream(stream);
    FPN_DISPATCH_LW(lw, three_interpolate_kernel, grid, dim3(kRowThreads), 0, st, b, m, c, n, vec, points, idx,
                    weight, out)
    return check_launch("fpn_three_interpolate");
}

int fpn_three_nn_interpolate(int b, int n, int m, int c, const float *xyz1, const float *xyz2,
                             const float *points2, float *out, float *dist, int *idx, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 1, "three_nn_interpolate: bad size (m must be >= 1)");
    if (b == 0 || n == 0) return FPN_OK;
    FPN_REQUIRE(xyz1 && xyz2 && points2 && out, "three_nn_interpolate: null pointer");
    FPN_REQUIRE(b <= 65535, "three_nn_interpolate: batch %d exceeds 65535", b);
    const int vec = c > 0 && c % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(points2) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int lw = pow2_at_least(c > 0 ? (vec ? c / 4 : c) : 1);
    dim3 grid((n + kNnThreads - 1) / kNnThreads, b);
    hipStream_t st = as_stream(stream);
    FPN_DISPATCH_LW(lw, three_nn_interpolate_kernel, grid, dim3(kNnThreads), 0, st, n, m, c, vec, xyz1, xyz2,
                    points2, out, dist, idx)
    return check_launch("fpn_three_nn_interpolate");
}

int fpn_three_nn_interpolate_bf16(int b, int n, int m, int c, const float *xyz1, const float *xyz2,
                                  const float *points2, void *out, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && c >= 0 && m >= 1, "three_nn_interpolate_bf16: bad size (m must be >= 1)");
    if (b == 0 || n == 0) return FPN_OK;
    FPN_REQUIRE(xyz1 && xyz2 && points2 && out, "three_nn_interpolate_bf16: null pointer");
    FPN_REQUIRE(b <= 65535, "three_nn_interpolate_bf16: batch %d exceeds 65535", b);
    const int vec = c > 0 && c % 4 == 0 && (reinterpret_cast<uintptr_t>(points2) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 7) == 0;
    const int lw = pow2_at_least(c > 0 ? (vec ? c / 4 : c) : 1);
    dim3 grid((n + kNnThreads - 1) / kNnThreads, b);
    hipStream_t st = as_stream(stream);
    FPN_DISPATCH_LW(lw, three_nn_interpolate_bf16_kernel, grid, dim3(kNnThreads), 0, st, n, m, c, vec, xyz1, xyz2,
                    points2, static_cast<__bf16 *>(out))
    return check_launch("fpn_three_nn_interpolate_bf16");
}

int fpn_select_top_k(int b, int n, int m, int k, const float *dist, int *outi, float *out, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && m >= 0, "select_top_k: negative size");
    FPN_REQUIRE(k > 0, "SelectionSort expects positive k");
    long rows = (long)b * m;
    if (rows == 0 || n == 0) return FPN_OK;
    FPN_REQUIRE(dist && outi && out, "select_top_k: null pointer");
    if (n > 16384) return fail(FPN_E_UNSUPPORTED, "select_top_k: n=%d exceeds the 16384-column limit", n);
    FPN_REQUIRE(rows <= INT32_MAX, "select_top_k: too many rows");
    hipLaunchKernelGGL(selection_sort_kernel<false>, dim3((unsigned)rows), dim3(kWave), (size_t)n * 8,
                       as_stream(stream), n, k, dist, outi, out, m, 0, static_cast<const float *>(nullptr),
                       static_cast<const float *>(nullptr));
    return check_launch("fpn_select_top_k");
}

int fpn_knn_point(int b, int n, int c, int m, int k, const float *xyz1, const float *xyz2, float *val, int *idx,
                  void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && c >= 1, "knn_point: bad size (b=%d n=%d m=%d c=%d)", b, n, m, c);
    FPN_REQUIRE(k > 0, "SelectionSort expects positive k");
    FPN_REQUIRE(k <= n, "knn_point: k=%d exceeds the %d dataset points", k, n);
    long rows = (long)b * m;
    if (rows == 0) return FPN_OK;
    FPN_REQUIRE(xyz1 && xyz2 && val && idx, "knn_point: null pointer");
    if (n > 16384) return fail(FPN_E_UNSUPPORTED, "knn_point: n=%d exceeds the 16384-point limit", n);
    FPN_REQUIRE(rows <= INT32_MAX, "knn_point: too many rows");
    hipLaunchKernelGGL(selection_sort_kernel<true>, dim3((unsigned)rows), dim3(kWave), (size_t)n * 8, as_stream(stream), n,
                       k, static_cast<const float *>(nullptr), idx, val, m, c, xyz1, xyz2);
    return check_launch("fpn_knn_point");
}

int fpn_mask_gather(int b, int n, int cpc, int npick, int policy, const float *point_cloud, const float *logits,
                    float *mask, float *mean, int *sel, float *obj, void *stream)
{
    FPN_REQUIRE(b >= 0 && n >= 1 && cpc >= 3 && npick >= 1, "mask_gather: bad size (n=%d cpc=%d npick=%d)", n, cpc, npick);
    FPN_REQUIRE(policy == FPN_PICK_FIRST || policy == FPN_PICK_SPREAD, "mask_gather: unknown pick policy %d", policy);
    if (b == 0) return FPN_OK;
    FPN_REQUIRE(point_cloud && logits && mean && obj, "mask_gather: null pointer");
    if (n > kMgMaxN) return fail(FPN_E_UNSUPPORTED, "mask_gather: n=%d exceeds %d", n, kMgMaxN);
    hipLaunchKernelGGL(mask_gather_kernel, dim3(b), dim3(kMgThreads), (size_t)n * 4, as_stream(stream), n,
                       cpc, npick, policy == FPN_PICK_SPREAD ? 1 : 0, point_cloud, logits, mask, mean, sel, obj);
    return check_launch("fpn_mask_gather");
}

}  // extern "C"
