// Probe of v_mfma_f64_16x16x4_f64 on gfx950: operand / result lane layout and the accumulation order inside one instruction.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, const double *C, double *D, int la, int lb)
{
    // layout hypothesis: lane l holds A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16];
    // D: lane l, register r holds D[i = 4 * (l / 16) + r][j = l % 16]
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    v4d c;
    for (int r = 0; r < 4; r++) c[r] = C[(la ? (4 * r + l / 16) : (4 * (l / 16) + r)) * 16 + l % 16];
    v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(la ? (4 * r + l / 16) : (4 * (l / 16) + r)) * 16 + l % 16] = d[r];
}
int main()
{
    std::vector<double> A(64), B(64), C(256), D(256);
    srand(1);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * pow(2.0, rand() % 8 - 4); };
    int layout_ok = 1;
    long n_seq = 0, n_rev = 0, n_tree = 0, n_tot = 0;
    for (int la = 0; la < 2; la++) {
    layout_ok = 1; n_seq = n_rev = n_tree = n_tot = 0;
    for (int it = 0; it < 200; it++) {
        for (auto &x : A) x = rnd();
        for (auto &x : B) x = rnd();
        for (auto &x : C) x = rnd();
        double *dA, *dB, *dC, *dD;
        hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048);
        hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, la, 0);
        hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                double s = C[i * 16 + j], t = C[i * 16 + j];
                for (int kk = 0; kk < 4; kk++) s = fma(A[i * 4 + kk], B[kk * 16 + j], s);
                for (int kk = 3; kk >= 0; kk--) t = fma(A[i * 4 + kk], B[kk * 16 + j], t);
                double p0 = fma(A[i * 4 + 1], B[16 + j], A[i * 4] * B[j]), p1 = fma(A[i * 4 + 3], B[48 + j], A[i * 4 + 2] * B[32 + j]);
                double u = C[i * 16 + j] + (p0 + p1);
                const double d = D[i * 16 + j];
                if (fabs(d - s) > 1e-9 * (fabs(s) + 1e-30)) layout_ok = 0;
                n_tot++;
                n_seq += d == s; n_rev += d == t; n_tree += d == u;
            }
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    }
    printf("D layout %s: ", la ? "i = 4 r + lane / 16" : "i = 4 (lane / 16) + r");
    printf("layout hypothesis %s; bitwise equal to: k-ascending fma chain %ld, k-descending chain %ld, pairwise tree %ld of %ld\n",
           layout_ok ? "OK" : "WRONG", n_seq, n_rev, n_tree, n_tot);
    }
    return 0;
}
