// store_rate.hip — what the spectrum kernels' memory pattern costs on its own (no transform): persistent wavefronts, one
// 1024-point frame per wavefront and turn — read 8 KB as 16 coalesced 8-byte loads per lane (512 B per instruction), write the
// 4 KB dB row as 16 dword stores per lane (256 B per instruction) or 4 dwordx4 stores (1 KB per instruction), optionally with
// a block of dependent float64 arithmetic per frame in between (FMAS per lane) standing in for the transform.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_rate.hip -o /tmp/store_rate && /tmp/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE, int FMAS>
__global__ __launch_bounds__(256) void k(const float2 *__restrict__ in, float *__restrict__ out, long rows, double a)
{
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    float2 nx[16];
    constexpr bool RD = MODE & 1, WR = MODE & 2, W4 = MODE & 4, PF = MODE & 8;
    auto fetch = [&](long r) {
        const float2 *x = in + (size_t)r * 1024;
#pragma unroll
        for (int q = 0; q < 16; q++) nx[q] = RD ? x[lane + 64 * q] : make_float2(1.0f + q, lane);
    };
    if (PF && wave < rows) fetch(wave);
    for (long r = wave; r < rows; r += nw) {
        if (!PF) fetch(r);
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = (double)nx[q].x + (double)nx[q].y;
        if (PF && r + nw < rows) fetch(r + nw);
#pragma unroll 1
        for (int i = 0; i < FMAS / 16; i++)
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = __fma_rn(v[q], a, 0.5);
        float *o = out + (size_t)r * 1024;
        if (WR) {
            if (W4) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    reinterpret_cast<float4 *>(o)[lane + 64 * q] = make_float4((float)v[4 * q], (float)v[4 * q + 1], (float)v[4 * q + 2], (float)v[4 * q + 3]);
            } else {
#pragma unroll
                for (int q = 0; q < 16; q++) o[lane + 64 * q] = (float)v[q];
            }
        } else {
            double s = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) s += v[q];
            if (s == 1.2345e300) o[lane] = (float)s;
        }
    }
}

template <int MODE, int FMAS>
void run(const char *name, const float2 *in, float *out, long rows, int wgs)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, FMAS>), dim3(wgs), dim3(256), 0, 0, in, out, rows, 0.999999);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)rows * 1024 * ((MODE & 1 ? 8 : 0) + (MODE & 2 ? 4 : 0));
    printf("%-58s wgs %5d fmas %5d  mean %.4f min %.4f ms  %.2f TB/s\n", name, wgs, FMAS, sum / 5, best, bytes / (sum / 5) / 1e9);
}

int main()
{
    const long rows = 65536;
    float2 *in;
    float *out;
    hipMalloc(&in, rows * 1024 * sizeof(float2));
    hipMalloc(&out, rows * 1024 * sizeof(float));
    hipMemset(in, 0, rows * 1024 * sizeof(float2));
    for (int wgs : {512, 1024, 2048}) {
        run<1, 0>("read 8 KB / row (dwordx2 x16)", in, out, rows, wgs);
        run<9, 0>("read, prefetched a row ahead", in, out, rows, wgs);
        run<2, 0>("write 4 KB / row (dword x16)", in, out, rows, wgs);
        run<6, 0>("write 4 KB / row (dwordx4 x4)", in, out, rows, wgs);
        run<3, 0>("read + write (dword stores)", in, out, rows, wgs);
        run<11, 0>("read prefetched + write (dword stores)", in, out, rows, wgs);
        run<15, 0>("read prefetched + write (dwordx4 stores)", in, out, rows, wgs);
        run<0, 800>("800 dependent-by-16 fma only", in, out, rows, wgs);
        run<9, 800>("read prefetched + 800 fma", in, out, rows, wgs);
        run<2, 800>("800 fma + write dword", in, out, rows, wgs);
        run<11, 800>("read prefetched + 800 fma + write dword", in, out, rows, wgs);
        run<15, 800>("read prefetched + 800 fma + write dwordx4", in, out, rows, wgs);
    }
    return 0;
}
