// valu_rate.hip — machine-wide issue rate of VALU instruction classes on gfx950 (hipEvent timing, every SIMD holding W wavefronts):
// wavefront-instructions x 64 lanes / time, and the clocks one wavefront-instruction occupies its SIMD if the shader clock is F GHz.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate [GHz]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang fp contract(off)

constexpr int CH = 8, UNR = 8;

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters)
{
    double v[CH];
    float f[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { v[c] = a + threadIdx.x * 1e-9 + c; f[c] = (float)v[c]; }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < UNR; r++) {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                if (MODE == 0) v[c] = __fma_rn(v[c], b, a);                                    // v_fma_f64
                if (MODE == 1) v[c] = v[c] + a;                                                // v_add_f64
                if (MODE == 2) v[c] = v[c] * b;                                                // v_mul_f64
                if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"((float)b), "v"((float)a));   // not packed
                if (MODE == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v[c]) : "v"(f[c]));  // v_cvt_f64_f32
                if (MODE == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(reinterpret_cast<unsigned &>(f[c])) : "v"(i));
                // operand forms of the fused NFM kernel's FIR: a tap in an SGPR pair, the accumulator started with an inline zero
                if (MODE == 6) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[c]) : "v"(v[(c + 1) % CH]), "s"(b));          // acc += x * s[tap]
                if (MODE == 7) asm volatile("v_fma_f64 %0, %1, %2, 0" : "=v"(v[c]) : "v"(v[(c + 1) % CH]), "s"(b));           // acc = x * s[tap] + 0
                if (MODE == 8) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[c]) : "v"(v[(c + 1) % CH]), "v"(v[(c + 2) % CH]));   // three VGPR pairs
                // round 6: the classes the counting select (pss_post.h) and the float32 scanner epilogue are made of
                if (MODE == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(v[(c + 1) % CH]), "v"(v[(c + 2) % CH]));   // packed: two float32 FMAs per lane
                if (MODE == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(v[(c + 1) % CH]));
                if (MODE == 12) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(f[c]), "v"(f[(c + 1) % CH]) : "vcc");                  // compare into VCC
                if (MODE == 13) { asm volatile("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(reinterpret_cast<unsigned &>(f[c])) : "v"(f[(c + 1) % CH]), "v"(i) : "vcc"); }   // the counting pair (x2)
                if (MODE == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[c]) : "v"(f[(c + 1) % CH]) : "vcc");
                if (MODE == 15) asm volatile("v_min_u32 %0, %0, %1" : "+v"(reinterpret_cast<unsigned &>(f[c])) : "v"(i));
                if (MODE == 16) asm volatile("v_cmp_gt_u64 vcc, %0, %1" : : "v"(v[c]), "v"(v[(c + 1) % CH]) : "vcc");                  // 64-bit compare
                if (MODE == 17) asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(reinterpret_cast<unsigned &>(f[c])));   // a DPP step of a wave reduction (with its hazard nop)
                if (MODE == 18) { unsigned s_; asm volatile("v_readlane_b32 %0, %1, 16" : "=s"(s_) : "v"(f[c])); asm volatile("" :: "s"(s_)); }
                if (MODE == 19) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[c]) : "v"(f[(c + 1) % CH]), "s"((unsigned long long)blockIdx.x * 0x9e3779b97f4a7c15ull));   // mask in an SGPR pair
                if (MODE == 20) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[c]) : "v"(f[(c + 1) % CH]) : "vcc");   // compare + select (x2)
                if (MODE == 21) { float t_; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(t_) : "v"(f[c]), "v"(f[(c + 1) % CH]) : "vcc"); asm volatile("" :: "v"(t_)); }   // result never read
                if (MODE == 9) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v[c]) : "v"(f[c]));                                // cvt feeding an fma
                                 asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[(c + 3) % CH]) : "v"(v[c]), "s"(b)); }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) s += v[c] + f[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int wg_per_cu, double ghz)
{
    const int blocks = 256 * wg_per_cu, iters = 20000;
    double *d;
    hipMalloc(&d, (size_t)blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 0.9999999, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * UNR * CH;   // wavefront-instructions of the measured class
    const double per_simd = winstr / 1024.0;                        // issued by one SIMD
    printf("%-14s %d wavefronts/SIMD: %8.3f ms  %7.2f T lane-ops/s  -> %5.2f clk per wavefront-instruction at %.2f GHz\n", name, wg_per_cu, ms,
           winstr * 64 / (ms * 1e-3) / 1e12, ms * 1e-3 * ghz * 1e9 / per_simd, ghz);
    hipFree(d);
}

int main(int argc, char **argv)
{
    const double ghz = argc > 1 ? atof(argv[1]) : 2.4;
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, ghz);
        run<1>("v_add_f64", w, ghz);
        run<2>("v_mul_f64", w, ghz);
        run<3>("v_fma_f32", w, ghz);
        run<4>("v_cvt_f64_f32", w, ghz);
        run<5>("v_add_u32", w, ghz);
        run<6>("fma v,s,acc", w, ghz);
        run<7>("fma v,s,0", w, ghz);
        run<8>("fma v,v,acc", w, ghz);
        run<9>("cvt+fma (x2)", w, ghz);
        run<10>("v_pk_fma_f32", w, ghz);
        run<11>("v_pk_mul_f32", w, ghz);
        run<12>("v_cmp_lt_u32", w, ghz);
        run<13>("cmp+addc (x2)", w, ghz);
        run<14>("v_cndmask_b32", w, ghz);
        run<15>("v_min_u32", w, ghz);
        run<16>("v_cmp_gt_u64", w, ghz);
        run<17>("nop+add_dpp", w, ghz);
        run<18>("v_readlane", w, ghz);
        run<19>("cndmask sgpr", w, ghz);
        run<20>("cmp+cndmask(x2)", w, ghz);
        run<21>("cndmask indep", w, ghz);
    }
    return 0;
}
