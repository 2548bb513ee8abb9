// Diagnostic: where does a chain-kernel workgroup spend its cycles?  Builds the chain
// kernels with in-kernel s_memtime stamps around every barrier, runs seg_head
// on random data, and prints the median cycles of each phase (compute
// segment between two barriers, and the barrier wait itself) for lane 0's wave.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I3d-object-detection_amd/csrc \
//         -DFPN_CHAIN_STAMPS tools/chain_stamps.hip 3d-object-detection_amd/csrc/api.cpp -x hip -o gpurun_out/chain_stamps
#include <algorithm>
#include <cstdio>
#include <vector>

#include "chain.hip"

static float *dev_random(size_t n, float scale)
{
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX * 2.f - 1.f);
    float *d;
    hipMalloc(&d, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    return d;
}

static void report(const char *name, long blocks, int nstamps, int waves)
{
    std::vector<unsigned long long> h((size_t)blocks * 8 * 64);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fpn::g_stamps), h.size() * 8);
    printf("%s: %ld workgroups; median cycles per segment and wave (c = compute, b = barrier wait)\n", name, blocks);
    for (int i = 1; i < nstamps; ++i) {
        printf("  seg %2d (%c):", i, (i & 1) ? 'c' : 'b');
        for (int w = 0; w < waves; ++w) {
            std::vector<unsigned long long> d;
            for (long b = 0; b < blocks; ++b) d.push_back(h[(b * 8 + w) * 64 + i] - h[(b * 8 + w) * 64 + i - 1]);
            std::sort(d.begin(), d.end());
            printf(" %6llu", d[d.size() / 2]);
        }
        printf("\n");
    }
    std::vector<unsigned long long> life;
    for (long b = 0; b < blocks; ++b) life.push_back(h[b * 8 * 64 + nstamps - 1] - h[b * 8 * 64]);
    std::sort(life.begin(), life.end());
    printf("  wave 0 first stamp -> last stamp: median %llu cycles\n", life[life.size() / 2]);
}

int main(int argc, char **argv)
{
    const int frustums = argc > 1 ? atoi(argv[1]) : 256, n = 1024;
    const long rows = (long)frustums * n;
    float *pc = dev_random(rows * 4, 10.f), *pfo, *part;
    hipMalloc(&pfo, rows * 64 * 4);
    hipMalloc(&part, rows / 64 * 1024 * 4);
    const float *head[15] = {dev_random(4 * 64, .3f),     dev_random(64, 1.f),   dev_random(64, .1f),
                             dev_random(64 * 64, .1f),    dev_random(64, 1.f),   dev_random(64, .1f),
                             dev_random(64 * 64, .1f),    dev_random(64, 1.f),   dev_random(64, .1f),
                             dev_random(64 * 128, .1f),   dev_random(128, 1.f),  dev_random(128, .1f),
                             dev_random(128 * 1024, .1f), dev_random(1024, 1.f), dev_random(1024, .1f)};
    for (int it = 0; it < 3; ++it) fpn_v1_seg_head_f32(rows, n, 4, pc, head, pfo, part, nullptr);
    hipDeviceSynchronize();
    report("seg_head (8 waves, 128 rows)", rows / 128, 10, 8);  // begin + 4 barriers x 2 + end
    return 0;
}
