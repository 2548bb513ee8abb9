// Probe: rate of the chain kernels' inner product loop in isolation.  Each wave repeats
// a TM x 1 tile product (A from an LDS image with odd row stride, B from global memory)
// and variants drop one operand path to show which one limits the MFMA rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int kKB = 16, kHB = 8;

// MODE 0: A from LDS, B from global (compiler-scheduled, reads next to their use)
// MODE 1: A from LDS, B in registers (no global traffic)
// MODE 2: A in registers, B from global (no LDS traffic)
// MODE 3: as 0 with explicit half-block register pipeline + sched fences
template <int TM, int MODE, int XS, int LDW, int KSTEPS>
__global__ void probe(const float *__restrict__ w, float *out, int reps)
{
    extern __shared__ float X[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < 32 * TM * XS; i += blockDim.x) X[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    floatx16 acc[TM];
    for (int a = 0; a < TM; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float *xs = X + l31 * XS + half;
    const float *wg = w + half * LDW + wv * 32 + l31;
    constexpr int NB = KSTEPS / kKB;
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 3) {
            float b0[kKB], b1[kKB], a0[TM][kHB], a1[TM][kHB];
            for (int s = 0; s < kKB; ++s) b0[s] = wg[(long)2 * s * LDW];
            for (int a = 0; a < TM; ++a) for (int s = 0; s < kHB; ++s) a0[a][s] = xs[a * 32 * XS + 2 * s];
#pragma unroll 1
            for (int kb = 0; kb < NB; kb += 2) {
                const float *x0 = xs + 2 * kb * kKB;
#pragma unroll
                for (int s = 0; s < kKB; ++s) b1[s] = wg[(long)2 * ((kb + 1) * kKB + s) * LDW];
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int s = 0; s < kHB; ++s) a1[a][s] = x0[a * 32 * XS + 2 * (kHB + s)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < kHB; ++s)
#pragma unroll
                    for (int a = 0; a < TM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[a][s], b0[s], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int s = 0; s < kHB; ++s) a0[a][s] = x0[a * 32 * XS + 2 * (kKB + s)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < kHB; ++s)
#pragma unroll
                    for (int a = 0; a < TM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[a][s], b0[kHB + s], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kb + 2 < NB) {
#pragma unroll
                    for (int s = 0; s < kKB; ++s) b0[s] = wg[(long)2 * ((kb + 2) * kKB + s) * LDW];
                }
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int s = 0; s < kHB; ++s) a1[a][s] = x0[a * 32 * XS + 2 * (kKB + kHB + s)];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < kHB; ++s)
#pragma unroll
                    for (int a = 0; a < TM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[a][s], b1[s], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kb + 2 < NB) {
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int s = 0; s < kHB; ++s) a0[a][s] = x0[a * 32 * XS + 2 * (2 * kKB + s)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < kHB; ++s)
#pragma unroll
                    for (int a = 0; a < TM; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[a][s], b1[kHB + s], acc[a], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float b0[kKB], b1[kKB];
#pragma unroll
            for (int s = 0; s < kKB; ++s) b0[s] = MODE == 1 ? 0.5f + s : wg[(long)2 * s * LDW];
#pragma unroll 1
            for (int kb = 0; kb < NB; kb += 2) {
#pragma unroll
                for (int s = 0; s < kKB; ++s) b1[s] = MODE == 1 ? 0.25f + s : wg[(long)2 * ((kb + 1) * kKB + s) * LDW];
#pragma unroll
                for (int s = 0; s < kKB; ++s)
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        float av = MODE == 2 ? (float)(a + s) : xs[a * 32 * XS + 2 * (kb * kKB + s)];
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[s], acc[a], 0, 0, 0);
                    }
                if (kb + 2 < NB) {
#pragma unroll
                    for (int s = 0; s < kKB; ++s) b0[s] = MODE == 1 ? 0.5f + s : wg[(long)2 * ((kb + 2) * kKB + s) * LDW];
                }
#pragma unroll
                for (int s = 0; s < kKB; ++s)
#pragma unroll
                    for (int a = 0; a < TM; ++a) {
                        float av = MODE == 2 ? (float)(a - s) : xs[a * 32 * XS + 2 * ((kb + 1) * kKB + s)];
                        acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1[s], acc[a], 0, 0, 0);
                    }
            }
        }
    }
    float s = 0;
    for (int a = 0; a < TM; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int TM, int MODE>
void run(int threads, const float *w, float *out)
{
    constexpr int XS = 129, LDW = 256, KSTEPS = 64;
    const int reps = 256;
    auto k = probe<TM, MODE, XS, LDW, KSTEPS>;
    size_t lds = (size_t)32 * TM * XS * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, threads, lds>>>(w, out, reps);
    hipEventRecord(e0);
    k<<<256, threads, lds>>>(w, out, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfmas = 256.0 * (threads / 64) * reps * KSTEPS * TM;
    printf("TM=%d MODE=%d waves/SIMD=%d: %7.3f ms  %6.1f TFLOP/s\n", TM, MODE, threads / 256, ms, mfmas * 4096 / ms / 1e9);
}

int main()
{
    float *w, *out;
    hipMalloc(&w, 128 * 256 * 4 + 4096);
    hipMemset(w, 0, 128 * 256 * 4 + 4096);
    hipMalloc(&out, 256 * 1024 * 4);
    for (int thr : {256, 512}) {
        if (thr == 256) { run<1,0>(256,w,out); run<1,1>(256,w,out); run<1,2>(256,w,out); run<1,3>(256,w,out);
                          run<2,0>(256,w,out); run<2,1>(256,w,out); run<2,2>(256,w,out); run<2,3>(256,w,out);
                          run<4,0>(256,w,out); run<4,1>(256,w,out); run<4,2>(256,w,out); run<4,3>(256,w,out); }
        else { run<1,0>(512,w,out); run<1,1>(512,w,out); run<1,2>(512,w,out); run<1,3>(512,w,out);
               run<2,0>(512,w,out); run<2,1>(512,w,out); run<2,2>(512,w,out); run<2,3>(512,w,out);
               run<4,0>(512,w,out); run<4,1>(512,w,out); run<4,2>(512,w,out); run<4,3>(512,w,out); }
    }
    return 0;
}
