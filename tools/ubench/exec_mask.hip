// exec_mask.hip — does a float64 VALU instruction cost fewer clocks when only part of the wavefront is active?
// One wavefront, the serial biquad step of the systolic IIR kernels (nine float64 operations per sample, four on the critical path) and
// plain dependent / independent v_fma_f64 chains, with lanes >= NACT switched off for the whole loop.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/exec_mask.hip -o /tmp/exec_mask && /tmp/exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)

__global__ void kbq(double *out, double b0, double b1, double b2, double a1, double a2, int iters, int nact, unsigned long long *cyc)
{
    if ((int)threadIdx.x >= nact) return;
    double z0 = threadIdx.x * 1e-9, z1 = 0.5, x = 1.0 + threadIdx.x * 1e-7, acc = 0.0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const double xn = b0 * x + z0;
            z0 = (b1 * x - a1 * xn) + z1;
            z1 = b2 * x - a2 * xn;
            acc += xn; x = x * 0.999999;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc + z0 + z1;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS>
__global__ void kfma(double *out, double a, double b, int iters, int nact, unsigned long long *cyc)
{
    if ((int)threadIdx.x >= nact) return;
    double v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = a + threadIdx.x * 1e-9 + c;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) v[c] = __fma_rn(v[c], b, a);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int iters = 4000;
    const int acts[] = {64, 48, 32, 17, 16, 8, 1};
    for (int nact : acts) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kbq, dim3(1), dim3(64), 0, 0, out, 0.9, -1.7, 0.8, -1.6, 0.7, iters, nact, cyc);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("biquad step       active lanes %2d : %6.1f clk per sample\n", nact, (double)h / (iters * 8.0));
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kfma<1>, dim3(1), dim3(64), 0, 0, out, 1.000001, 0.999999, iters, nact, cyc);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("fma_f64 dependent active lanes %2d : %6.2f clk per instruction\n", nact, (double)h / (iters * 8.0));
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kfma<8>, dim3(1), dim3(64), 0, 0, out, 1.000001, 0.999999, iters, nact, cyc);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("fma_f64 8 chains  active lanes %2d : %6.2f clk per instruction\n", nact, (double)h / (iters * 8.0 * 8));
    }
    return 0;
}
