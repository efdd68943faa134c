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
    run<0, 1>("fma_f64", 1); run<0, 2>("fma_f64", 1); run<0, 4>("fma_f64", 1); run<0, 8>("fma_f64", 1);
    run<0, 1>("fma_f64", 2); run<0, 4>("fma_f64", 2); run<0, 4>("fma_f64", 3); run<0, 4>("fma_f64", 4);
    run<0, 1>("fma_f64", 4); run<0, 2>("fma_f64", 4);
    run<1, 4>("add_f64", 2); run<1, 4>("add_f64", 4); run<2, 4>("mul_f64", 2); run<2, 4>("mul_f64", 4);
    run<1, 1>("add_f64", 1); run<1, 4>("add_f64", 1); run<1, 8>("add_f64", 1);
    run<2, 1>("mul_f64", 1); run<2, 4>("mul_f64", 1); run<2, 8>("mul_f64", 1);
    run<3, 1>("cvt_rt", 1); run<3, 8>("cvt_rt", 1);
    return 0;
}
