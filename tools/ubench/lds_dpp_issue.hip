// lds_dpp_issue.hip — what an LDS or DPP instruction costs a LONE wavefront (one wavefront on a CU, nothing else resident): clocks per
// instruction of a long run of the same instruction, stores with the data ready and loads whose results are only summed at the end.
// The recurrence wavefront of k_am_grp (pss_demod.hip) is such a wavefront: it has a SIMD's issue slots to itself.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dpp_issue.hip -o /tmp/lds_dpp_issue && /tmp/lds_dpp_issue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d2_t __attribute__((ext_vector_type(2)));

// MODE 0 ds_write_b64, 1 ds_write_b128, 2 ds_read_b64, 3 ds_read_b128, 4 v_mov_b32 dpp row_shr:3, 5 ds_write_b128 with 12 of 64 lanes active
template <int MODE>
__global__ void k(double *out, int iters, unsigned long long *cyc)
{
    __shared__ __align__(16) double buf[64 * 34];
    const int lane = threadIdx.x;
    double *p = buf + lane * 34;            // 272 bytes apart: the 16 lanes a b128 access serves together hit 16 different bank groups
    double a = 1.0 + lane, b = 2.0 + lane, s0 = 0.0, s1 = 0.0;
    const d2_t ab = {a, b};
    int d = lane, e = lane + 1;
    for (int i = 0; i < 34; i++) p[i] = i;
    __syncthreads();
    const bool on = MODE != 5 || (lane & 15) >= 12 || lane >= 52;   // MODE 5: the twelve lanes 12-15, 28-31, 44-47 (+ some above) — a section-4 pattern
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (MODE == 0) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"((unsigned)(size_t)p), "v"(a), "n"(16 * r) : "memory");
            if (MODE == 1) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"((unsigned)(size_t)p), "v"(ab), "n"(16 * r) : "memory");
            if (MODE == 2) { double v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"(16 * r) : "memory"); s0 += 0; asm volatile("" ::"v"(v)); }
            if (MODE == 3) { d2_t v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)p), "n"(16 * r) : "memory"); asm volatile("" ::"v"(v)); }
            if (MODE == 4) { d = __builtin_amdgcn_update_dpp(e, d, 0x113, 0xf, 0xf, false); e = __builtin_amdgcn_update_dpp(d, e, 0x113, 0xf, 0xf, false); }
            if (MODE == 5) { if (on) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"((unsigned)(size_t)p), "v"(ab), "n"(16 * r) : "memory"); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    out[lane] = s0 + s1 + d + e + p[3];
    if (lane == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char *name, int per_iter)
{
    double *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    const int iters = 4000;
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE>), dim3(1), dim3(64), 0, 0, out, iters, cyc);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %6.1f clk per instruction (one wavefront, %d instructions per loop turn)\n", name, (double)h / ((double)iters * per_iter), per_iter);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("ds_write_b64  (64 lanes)", 16);
    run<1>("ds_write_b128 (64 lanes)", 16);
    run<5>("ds_write_b128 (16 of 64 lanes, under exec)", 16);
    run<2>("ds_read_b64   (64 lanes)", 16);
    run<3>("ds_read_b128  (64 lanes)", 16);
    run<4>("v_mov_b32_dpp row_shr:3 (dependent pair)", 32);
    return 0;
}
