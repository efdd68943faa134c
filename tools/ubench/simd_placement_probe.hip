// Where do the four wavefronts of a 256-thread workgroup land?  Launch shape of k_nfm_fwd (1024 workgroups x 256 threads,
// 37 KB dynamic LDS, 4 workgroups resident per CU); every wavefront records HW_ID (SIMD / CU / SE) and XCC_ID.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/simd_placement_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>

__global__ __launch_bounds__(256, 4) void probe(unsigned *out, long spin)
{
    extern __shared__ unsigned char smem[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long t0 = clock64();
    while (clock64() - t0 < spin) { smem[threadIdx.x] = (unsigned char)t0; }   // stay resident so that the machine fills up
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
    }
}

int main(int argc, char **argv)
{
    // simd_probe [workgroups] [dynamic LDS bytes]   (k_am_grp's shape: 683 workgroups x 256 threads, 25152 bytes of LDS)
    const int wgs = argc > 1 ? atoi(argv[1]) : 1024;
    const int lds = argc > 2 ? atoi(argv[2]) : 37 * 1024;
    unsigned *d;
    hipMalloc(&d, wgs * 4 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), lds, 0, d, 2000000L);
    hipDeviceSynchronize();
    std::vector<unsigned> h(wgs * 8);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    int hist[4][4] = {};
    std::map<unsigned, std::vector<int>> cu_wgs;   // (xcc, se, cu) -> workgroups
    int same = 0;
    for (int b = 0; b < wgs; b++) {
        unsigned seen = 0;
        for (int w = 0; w < 4; w++) {
            unsigned hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            int simd = (hw >> 4) & 3;
            hist[w][simd]++;
            seen |= 1u << simd;
            if (w == 0) cu_wgs[(xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15)].push_back(b);
        }
        same += seen == 0xf;
    }
    printf("wave index -> SIMD histogram (rows: wave 0..3 of the workgroup, columns: SIMD 0..3)\n");
    for (int w = 0; w < 4; w++) printf("  wave %d: %5d %5d %5d %5d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("workgroups whose four wavefronts sit on four different SIMDs: %d of %d\n", same, wgs);
    printf("distinct CUs used: %zu\n", cu_wgs.size());
    int shown = 0;
    for (auto &kv : cu_wgs) {
        if (shown++ >= 6) break;
        printf("  xcc %u se %u cu %2u: workgroups", kv.first >> 16, (kv.first >> 8) & 7, kv.first & 15);
        for (int b : kv.second) {
            printf(" %d(", b);
            for (int w = 0; w < 4; w++) printf("%u", (h[(b * 4 + w) * 2] >> 4) & 3);
            printf(")");
        }
        printf("\n");
    }
    return 0;
}
