// f64lat.hip — micro-benchmark: issue rate and dependent latency of float64 VALU ops on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/f64lat.hip -o /tmp/f64lat && /tmp/f64lat
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)

template <int MODE, int CHAINS>
__global__ void k(double *out, double a, double b, int iters, unsigned long long *cyc)
{
    double v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = a + threadIdx.x * 1e-9 + c;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (MODE == 0) v[c] = __fma_rn(v[c], b, a);
                if (MODE == 1) v[c] = v[c] + a;
                if (MODE == 2) v[c] = v[c] * b;
                if (MODE == 3) v[c] = (double)(float)v[c];   // cvt f64->f32->f64
                if (MODE == 4) { float t = __fmaf_rn((float)v[c], 0.999f, 1.0f); asm volatile("" : "+v"(t)); v[c] = t; }
                if (MODE == 5) { float t = (float)v[c]; asm volatile("" : "+v"(t)); v[c] = (double)t; }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE, int CHAINS>
__global__ void kf(float *out, float a, float b, int iters, unsigned long long *cyc)
{
    float v[CHAINS]; double d[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) { v[c] = a + threadIdx.x * 1e-6f + c; d[c] = 0.0; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (MODE == 0) v[c] = __fmaf_rn(v[c], b, a);
                if (MODE == 1) v[c] = v[c] + a;
                if (MODE == 2) { d[c] = (double)v[c]; asm volatile("" : "+v"(d[c])); v[c] += 1.0f; }
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c] + (float)d[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// the serial biquad step of the systolic IIR kernels (k_am_sys, k_iir4_sys): nine float64 operations per sample, four of them on the
// recurrence's critical path; ORDER 0 as the source states it, ORDER 1 with the three input products of the NEXT sample issued early
template <int ORDER>
__global__ void kbq(double *out, double b0, double b1, double b2, double a1, double a2, int iters, unsigned long long *cyc)
{
    double z0 = threadIdx.x * 1e-9, z1 = 0.5, x = 1.0 + threadIdx.x * 1e-7, acc = 0.0;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (ORDER == 0) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const double xn = b0 * x + z0;
                z0 = (b1 * x - a1 * xn) + z1;
                z1 = b2 * x - a2 * xn;
                acc += xn; x = x * 0.999999;
            }
        }
    } else {
        double p0 = b0 * x, p1 = b1 * x, p2 = b2 * x;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const double xn = p0 + z0;
                x = x * 0.999999;
                const double q0 = b0 * x, q1 = b1 * x, q2 = b2 * x;     // next sample's products: off the critical path
                const double t1 = a1 * xn, t2 = a2 * xn;
                z0 = (p1 - t1) + z1;
                z1 = p2 - t2;
                acc += xn;
                p0 = q0; p1 = q1; p2 = q2;
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + z0 + z1;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int ORDER>
void runbq(const char *name, int waves_per_simd)
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8);
    int iters = 2000;
    hipLaunchKernelGGL((kbq<ORDER>), dim3(1), dim3(64 * 4 * waves_per_simd), 0, 0, out, 0.9, -1.7, 0.8, -1.6, 0.7, iters, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((kbq<ORDER>), dim3(1), dim3(64 * 4 * waves_per_simd), 0, 0, out, 0.9, -1.7, 0.8, -1.6, 0.7, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s waves/SIMD=%d : %.1f clk per sample (11 float64 instructions)\n", name, waves_per_simd, (double)h / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

template <int MODE, int CHAINS>
void runf(const char *name, int waves_per_simd)
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8);
    int iters = 2000;
    int threads = 64 * 4 * waves_per_simd;
    hipLaunchKernelGGL((kf<MODE, CHAINS>), dim3(1), dim3(threads), 0, 0, out, 1.000001f, 0.999999f, iters, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((kf<MODE, CHAINS>), dim3(1), dim3(threads), 0, 0, out, 1.000001f, 0.999999f, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double per = (double)h / (iters * 8.0 * CHAINS);
    printf("%-12s chains=%d : %.2f clk per loop element (s_memtime units)\n", name, CHAINS, per);
    hipFree(out); hipFree(cyc);
}

template <int MODE, int CHAINS>
void run(const char *name, int waves_per_simd)
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8);
    int iters = 2000;
    int threads = 64 * 4 * waves_per_simd;  // one block on one CU: waves spread over the 4 SIMDs
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(1), dim3(threads), 0, 0, out, 1.000001, 0.999999, iters, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, CHAINS>), dim3(1), dim3(threads), 0, 0, out, 1.000001, 0.999999, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double per = (double)h / (iters * 8.0 * CHAINS);
    printf("%-10s chains=%d waves/SIMD=%d : %.2f clk per wave-instruction (s_memtime units)\n", name, CHAINS, waves_per_simd, per);
    hipFree(out); hipFree(cyc);
}

int main()
{
    runbq<0>("biquad step, source order", 1); runbq<1>("biquad step, products early", 1); runbq<0>("biquad step, source order", 2); runbq<1>("biquad step, products early", 2);
    run<0, 1>("fma_f64", 1); run<0, 2>("fma_f64", 1); run<0, 4>("fma_f64", 1); run<0, 8>("fma_f64", 1);
    run<0, 1>("fma_f64", 2); run<0, 4>("fma_f64", 2); run<0, 4>("fma_f64", 3); run<0, 4>("fma_f64", 4);
    run<0, 1>("fma_f64", 4); run<0, 2>("fma_f64", 4);
    run<1, 4>("add_f64", 2); run<1, 4>("add_f64", 4); run<2, 4>("mul_f64", 2); run<2, 4>("mul_f64", 4);
    run<1, 1>("add_f64", 1); run<1, 4>("add_f64", 1); run<1, 8>("add_f64", 1);
    run<2, 1>("mul_f64", 1); run<2, 4>("mul_f64", 1); run<2, 8>("mul_f64", 1);
    run<3, 1>("cvt_rt", 1); run<3, 8>("cvt_rt", 1);
    run<5, 1>("cvt_rt_noopt", 1); run<5, 8>("cvt_rt_noopt", 1);
    run<4, 8>("cvt+fmaf+cvt", 1);
    runf<0, 1>("fma_f32", 1); runf<0, 8>("fma_f32", 1); runf<1, 8>("add_f32", 1); runf<2, 8>("cvt_f64_f32", 1);
    return 0;
}
