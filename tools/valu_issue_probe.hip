// Probe: issue cost per instruction kind on one SIMD, by waves per SIMD - independent
// v_add_f32, v_pk_add_f32, v_min_u32, v_cmp (writing vcc), scalar ALU, and mixes of vector
// and scalar instructions (do they issue side by side from different waves?).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_issue_probe.hip -o tools/bin/valu_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a0)
{
    float f[8];
    float2v p[8];
    unsigned u[8];
    unsigned sacc = 1;
    for (int i = 0; i < 8; ++i) { f[i] = a0 + i + threadIdx.x; p[i] = float2v{a0 + i, a0 - i}; u[i] = threadIdx.x * 7 + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]));
            else if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k & 7]) : "v"(p[(k + 1) & 7]));
            else if (KIND == 2) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7]));
            else if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k & 7]) : "v"(p[(k + 1) & 7]));
            else if (KIND == 4) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) : : "scc");
            else if (KIND == 5) {     // one vector + one scalar instruction
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]));
                asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc) : : "scc");
            } else if (KIND == 6) {   // compare into vcc, then the scalar unit reads vcc
                asm volatile("v_cmp_eq_u32 vcc, %1, %2\n s_lshr_b64 s[20:21], vcc, 3\n s_or_b32 %0, %0, s20" : "+s"(sacc) : "v"(u[k & 7]), "v"(u[(k + 1) & 7]) : "vcc", "s20", "s21", "scc");
            } else if (KIND == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]));
            else if (KIND == 8) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(u[k & 7]) : "v"(u[(k + 1) & 7]), "v"(u[(k + 2) & 7]));
            else if (KIND == 9) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]));
            else if (KIND == 10) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]), "v"(f[(k + 2) & 7]));
            else if (KIND == 11) asm volatile("v_cmp_eq_f32 s[20:21], %0, %1" : : "v"(f[k & 7]), "v"(f[(k + 1) & 7]) : "s20", "s21");
            else if (KIND == 12) asm volatile("v_cmp_eq_u32 s[20:21], %0, %1" : : "v"(u[k & 7]), "v"(u[(k + 1) & 7]) : "s20", "s21");
            else if (KIND == 13) asm volatile("s_bitcmp1_b64 s[22:23], 5\n s_cmov_b32 %0, 7" : "+s"(sacc) : : "scc");
            else if (KIND == 14) {    // the search step: compare into an SGPR pair, test one lane's bit, conditional move
                asm volatile("v_cmp_eq_f32 s[20:21], %1, %2\n s_bitcmp1_b64 s[20:21], 5\n s_cmov_b32 %0, 7" : "+s"(sacc) : "v"(f[k & 7]), "v"(f[(k + 1) & 7]) : "s20", "s21", "scc");
            }
        }
    }
    float s = sacc;
    for (int i = 0; i < 8; ++i) s += f[i] + p[i][0] + p[i][1] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char *name, int per_iter)
{
    float *out;
    hipMalloc(&out, 256 * 8 * 1024 * 4);
    const int iters = 4096;
    for (int wps : {1, 2, 4, 8}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        probe<KIND><<<256 * wps, 256>>>(out, iters, 1.f);
        hipEventRecord(e0);
        probe<KIND><<<256 * wps, 256>>>(out, iters, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_simd = (double)iters * 64 * per_iter * wps;
        printf("%-28s waves/SIMD=%d  %7.3f ms  %5.2f clocks per instruction and SIMD (2.4 GHz)\n", name, wps, ms,
               ms * 1e-3 * 2.4e9 / inst_per_simd);
    }
    hipFree(out);
}

int main()
{
    run<0>("v_add_f32", 1);
    run<7>("v_mul_f32", 1);
    run<1>("v_pk_add_f32", 1);
    run<3>("v_pk_mul_f32", 1);
    run<2>("v_min_u32", 1);
    run<8>("v_max3_u32", 1);
    run<4>("s_add_u32", 1);
    run<5>("v_add_f32 + s_add_u32", 2);
    run<6>("v_cmp->vcc + 2 scalar", 3);
    run<9>("v_min_f32", 1);
    run<10>("v_max3_f32", 1);
    run<11>("v_cmp_eq_f32 -> sgpr", 1);
    run<12>("v_cmp_eq_u32 -> sgpr", 1);
    run<13>("s_bitcmp1 + s_cmov", 2);
    run<14>("v_cmp + s_bitcmp1 + s_cmov", 3);
    return 0;
}
