// Probe: fp32 MFMA (32x32x2) issue rate from registers only, by waves per SIMD and by
// the number of independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int ACC>
__global__ void probe(float *out, int iters, float a0, float b0)
{
    floatx16 acc[ACC];
    for (int i = 0; i < ACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < ACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ACC>
void run(int threads, int blocks_per_cu)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int iters = 4096 / ACC;   // 8*ACC*iters = 32768 MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<ACC><<<256 * blocks_per_cu, threads>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e0);
    probe<ACC><<<256 * blocks_per_cu, threads>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double waves = 256.0 * blocks_per_cu * threads / 64;
    double flops = waves * 32768.0 * 32 * 32 * 2 * 2;
    printf("ACC=%d threads=%4d blocks/CU=%d  waves/SIMD=%.1f  %8.3f ms  %7.1f TFLOP/s\n", ACC, threads, blocks_per_cu,
           blocks_per_cu * threads / 256.0, ms, flops / ms / 1e9);
    hipFree(out);
}

int main()
{
    run<1>(256, 1); run<2>(256, 1); run<4>(256, 1);
    run<1>(512, 1); run<2>(512, 1); run<4>(512, 1);
    run<1>(256, 3); run<2>(256, 3); run<4>(256, 3);
    run<2>(1024, 1);
    return 0;
}
