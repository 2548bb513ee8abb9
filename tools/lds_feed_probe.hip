// Probe: bf16 MFMAs (32x32x16) whose A operand is re-read from LDS (ds_read_b128, 1 KiB per
// wave and fragment) every NREAD-th MFMA, one wave per SIMD: cycles per MFMA by the number
// of fragment reads per four MFMAs and by how the reads are placed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// READS: fragment reads per group of four MFMAs (0..4).  FRONT: all of a group's reads
// behind its first MFMA, else one behind each MFMA.
template <int READS, bool FRONT>
__global__ __launch_bounds__(256) void probe(float *out, int iters, const bf16x8 *src)
{
    __shared__ bf16x8 lds[4 * 256];
    for (int i = threadIdx.x; i < 4 * 256; i += 256) lds[i] = src[i];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    bf16x8 frag[2][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) frag[0][f] = frag[1][f] = lds[f * 64 + lane];
    bf16x8 b = lds[lane];
    floatx16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {        // 8 groups of 4 MFMAs
            const int cur = u & 1, nxt = cur ^ 1;
            const int slot = (u & 3) * 256;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag[cur][m], b, acc[m & 1], 0, 0, 0);
                if (FRONT) {
                    if (m == 0)
#pragma unroll
                        for (int f = 0; f < READS; ++f) frag[nxt][f] = lds[slot + f * 64 + lane];
                } else if (m < READS) {
                    frag[nxt][m] = lds[slot + m * 64 + lane];
                }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (FRONT ? m == 0 : m < READS) __builtin_amdgcn_sched_group_barrier(0x100, FRONT ? READS : 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int READS, bool FRONT>
void run(const bf16x8 *src, float *out)
{
    const int iters = 512;     // 32 * iters MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<READS, FRONT><<<256, 256>>>(out, iters, src);
    hipEventRecord(e0);
    probe<READS, FRONT><<<256, 256>>>(out, iters, src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = 32.0 * iters;
    printf("reads per 4 MFMAs %d, %s: %7.3f ms = %5.1f cycles per MFMA at 2.4 GHz\n", READS,
           FRONT ? "behind the first MFMA" : "one per MFMA        ", ms, ms * 1e-3 * 2.4e9 / mfmas);
}

int main()
{
    bf16x8 *src; float *out;
    hipMalloc(&src, 4 * 256 * 16); hipMemset(src, 0, 4 * 256 * 16);
    hipMalloc(&out, 256 * 256 * 4);
    run<0, false>(src, out);
    run<1, false>(src, out); run<2, false>(src, out); run<4, false>(src, out);
    run<2, true>(src, out); run<4, true>(src, out);
    return 0;
}
