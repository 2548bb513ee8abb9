// Probe: how many vector-ALU instructions fit beside a bf16 MFMA (32x32x16, 32 clocks of the
// matrix pipe) before they cost time - K independent packed 16-bit maxima / conversions
// after every MFMA, accumulators in VGPRs, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int K, int KIND>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a0)
{
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = a0 * r;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x); b[e] = (__bf16)a0; }
    int v[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i; f[i] = a0 + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v[k & 7]) : "v"(v[(k + 1) & 7]));
                else if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[k & 7]) : "v"(f[k & 7]), "v"(f[(k + 1) & 7]));
                else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int KIND>
void run(int blocks_per_cu)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int iters = 2048;     // 16384 MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<K, KIND><<<256 * blocks_per_cu, 256>>>(out, iters, 1.f);
    hipEventRecord(e0);
    probe<K, KIND><<<256 * blocks_per_cu, 256>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = 16384.0 * blocks_per_cu;
    printf("kind=%d K=%d waves/SIMD=%d  %7.3f ms  %6.1f ns per MFMA and SIMD (%.1f clocks at 2.4 GHz)\n", KIND, K, blocks_per_cu, ms,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
    hipFree(out);
}

int main()
{
    run<0, 0>(1); run<2, 0>(1); run<4, 0>(1); run<5, 0>(1); run<6, 0>(1); run<7, 0>(1); run<8, 0>(1);
    run<4, 1>(1); run<6, 1>(1); run<4, 2>(1); run<6, 2>(1);
    run<0, 0>(2); run<4, 0>(2); run<6, 0>(2); run<8, 0>(2); run<6, 1>(2); run<6, 2>(2);
    return 0;
}
