/*
 * The C ABI used from plain C, with no Python and no torch in the process: farthest point
 * sampling and a ball query on one synthetic frustum, the results checked against the same
 * definitions written as loops (fps: greedy pick of the largest running minimum, lowest
 * index on ties; ball query: first nsample indices inside the radius, short rows padded with
 * the first hit).  What a C or C++ host of the reference's operators would write against
 * include/fpn.h.
 *
 *   gcc -std=c99 -O1 examples/c_abi_client.c -Iinclude -I/opt/rocm/include -L3d-object-detection_amd/fpn -lfpn_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,'$ORIGIN/../3d-object-detection_amd/fpn' -Wl,-rpath,/opt/rocm/lib \
 *       -o examples/c_abi_client        (__graft_entry__.build() does this)
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "fpn.h"

#define N 1000
#define M 37
#define NS 16

static int check(hipError_t e, const char *what)
{
    if (e != hipSuccess) fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e));
    return e != hipSuccess;
}

int main(void)
{
    static float pts[N * 3], centres[M * 3], mind[N];
    static int picks[M], want[M], idx[M * NS], cnt[M];
    unsigned s = 12345u;
    for (int i = 0; i < N * 3; ++i) {
        s = s * 1664525u + 1013904223u;
        pts[i] = (float)(s >> 8) / 16777216.0f;
    }
    float *d_pts, *d_centres;
    int *d_picks, *d_idx, *d_cnt;
    if (check(hipMalloc((void **)&d_pts, sizeof pts), "hipMalloc") ||
        check(hipMalloc((void **)&d_centres, sizeof centres), "hipMalloc") ||
        check(hipMalloc((void **)&d_picks, sizeof picks), "hipMalloc") ||
        check(hipMalloc((void **)&d_idx, sizeof idx), "hipMalloc") || check(hipMalloc((void **)&d_cnt, sizeof cnt), "hipMalloc") ||
        check(hipMemcpy(d_pts, pts, sizeof pts, hipMemcpyHostToDevice), "copy"))
        return 2;

    if (fpn_farthest_point_sample(1, N, M, d_pts, d_picks, NULL) != FPN_OK) {
        fprintf(stderr, "fps: %s\n", fpn_last_error_string());
        return 3;
    }
    if (check(hipMemcpy(picks, d_picks, sizeof picks, hipMemcpyDeviceToHost), "copy")) return 2;
    /* the definition, as loops (products and sums rounded one by one, like the kernel) */
    for (int i = 0; i < N; ++i) mind[i] = 1e10f;
    int cur = 0;
    want[0] = 0;
    for (int j = 1; j < M; ++j) {
        int best = 0;
        float bestd = -1.f;
        for (int k = 0; k < N; ++k) {
            const float dx = pts[k * 3] - pts[cur * 3], dy = pts[k * 3 + 1] - pts[cur * 3 + 1], dz = pts[k * 3 + 2] - pts[cur * 3 + 2];
            volatile float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            volatile float sxy = xx + yy;
            const float d = sxy + zz;
            if (d < mind[k]) mind[k] = d;
            if (mind[k] > bestd) { bestd = mind[k]; best = k; }
        }
        want[j] = cur = best;
    }
    for (int j = 0; j < M; ++j)
        if (picks[j] != want[j]) {
            fprintf(stderr, "fps pick %d: got %d, want %d\n", j, picks[j], want[j]);
            return 4;
        }

    for (int j = 0; j < M; ++j)
        for (int c = 0; c < 3; ++c) centres[j * 3 + c] = pts[picks[j] * 3 + c];
    const float radius = 0.15f;
    if (check(hipMemcpy(d_centres, centres, sizeof centres, hipMemcpyHostToDevice), "copy")) return 2;
    if (fpn_query_ball_point(1, N, M, radius, NS, d_pts, d_centres, d_idx, d_cnt, NULL) != FPN_OK) {
        fprintf(stderr, "ball query: %s\n", fpn_last_error_string());
        return 3;
    }
    if (check(hipMemcpy(idx, d_idx, sizeof idx, hipMemcpyDeviceToHost), "copy") ||
        check(hipMemcpy(cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost), "copy"))
        return 2;
    for (int j = 0; j < M; ++j) {
        int row[NS], c = 0;
        for (int k = 0; k < N && c < NS; ++k) {
            const float dx = centres[j * 3] - pts[k * 3], dy = centres[j * 3 + 1] - pts[k * 3 + 1], dz = centres[j * 3 + 2] - pts[k * 3 + 2];
            volatile float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            volatile float sxy = xx + yy;
            const float d = sqrtf(sxy + zz);
            if ((d > 1e-20f ? d : 1e-20f) < radius) row[c++] = k;
        }
        for (int l = c; l < NS; ++l) row[l] = c ? row[0] : 0;
        if (cnt[j] != c) {
            fprintf(stderr, "ball %d: %d hits, want %d\n", j, cnt[j], c);
            return 4;
        }
        for (int l = 0; l < NS; ++l)
            if (idx[j * NS + l] != row[l]) {
                fprintf(stderr, "ball %d slot %d: got %d, want %d\n", j, l, idx[j * NS + l], row[l]);
                return 4;
            }
    }
    printf("c_abi_client: ABI %d, %d picks and %d balls agree with the definitions\n", fpn_abi_version(), M, M);
    hipFree(d_pts); hipFree(d_centres); hipFree(d_picks); hipFree(d_idx); hipFree(d_cnt);
    return 0;
}
