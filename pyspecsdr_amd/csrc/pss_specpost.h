// pss_specpost.h — compute_fft and the caller's post-process fused for 1024-point frames (BASELINE cfg 2): the dB row a
// wavefront has just produced (pss_fft_r16.h: 64 threads x 16 bins) goes through the frame's own LDS exchange buffer into
// the post-process of pss_post.h (one wavefront per row, 16 consecutive elements per lane) without a round trip through
// HBM.  Outputs: the dB row (compute_fft's result, signal_processing.py:243-264), the smoothed / clamped row
// (pyspecsdr.py:2278-2283) and its finite extremes.  Saves the 4 KB re-read of every dB row and a launch.
#pragma once
#include <hip/hip_runtime.h>

#include "pss_fft_r16.h"
#include "pss_post.h"

// the spectrum half of this kernel must produce k_spectrum_r16's bits: same contraction mode as pss_fft_r16.h (the post-process
// half lives in pss_post.h's functions, which were defined under contract(off))
#pragma clang fp contract(fast)

namespace pss_sp {

template <bool PREFETCH>
__global__ __launch_bounds__(256) void k_spectrum_post_1024(const float2 *__restrict__ iq, float *__restrict__ db,
                                                            float *__restrict__ post, float *__restrict__ row_lo,
                                                            float *__restrict__ row_hi, const double2 *__restrict__ tw,
                                                            const double *__restrict__ win, long n_frames, int flags)
{
    constexpr int LOG_R3 = 2;
    using C = pss_r16::Cfg<LOG_R3>;
    constexpr int R3 = C::R3, T = C::T, N = C::N, FPW = C::FPW;   // 4, 64, 1024, 4: a frame is one wavefront
    constexpr int EPL = 16, S = pss_post::PostCfg<EPL>::S, Q = EPL / 4;
    static_assert(T == 64 && (T + 1) * S * sizeof(float) <= C::EX * sizeof(double2), "the staged row must fit the frame's exchange buffer");
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex_all = reinterpret_cast<double2 *>(smem);
    double2 *tw2 = ex_all + (size_t)FPW * C::EX;
    const int tid = threadIdx.x;
    const int fl = __builtin_amdgcn_readfirstlane(tid / T);
    const int t = tid % T;
    double2 *ex = ex_all + (size_t)fl * C::EX;
    float *buf = reinterpret_cast<float *>(ex);
    double2 tw1[16];
    double w[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)t * k2];
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) w[n2] = win[t + T * n2];
    if (tid < R3 * 16) tw2[tid] = tw[(size_t)((tid / 16) * (tid % 16)) * 16];
    __syncthreads();
    int slot[Q];
#pragma unroll
    for (int j = 0; j < Q; j++) {
        const int e0 = 4 * (j * T + t);
        slot[j] = (e0 / EPL) * S + (e0 % EPL);
    }
    int phase = 0;
    const long groups = (n_frames + FPW - 1) / FPW;
    float2 nx[16];
    auto fetch = [&](long g) {
        const long f = g * FPW + fl;
        const float2 *x = iq + (size_t)(f < n_frames ? f : 0) * N;
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) nx[n2] = x[t + T * n2];
    };
    if (PREFETCH && (long)blockIdx.x < groups) fetch(blockIdx.x);
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long f = g * FPW + fl;
        const bool valid = f < n_frames;     // wavefront-uniform: the frame is this wavefront's
        double2 v[16];
        if constexpr (!PREFETCH) fetch(g);
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) v[n2] = make_double2((double)nx[n2].x * w[n2], (double)nx[n2].y * w[n2]);
        if (PREFETCH && g + gridDim.x < groups) fetch(g + gridDim.x);
        float *out = valid ? db + (size_t)f * N : nullptr;
        float dbv[16];
        pss_r16::r16_core<LOG_R3, true>(v, ex, tw1, tw2, t, [&](int i, int k, double2 X) {
            const float d = pss_r16::db_of_fast(X.x * X.x + X.y * X.y + 1e-10);   // ("db_exact" takes the two separate kernels)
            if (out) out[(k + N / 2) & (N - 1)] = d;  // fftshift; 64 consecutive bins per store instruction
            dbv[i] = d;
        });
        pss_r16::frame_sync<true>();         // the transform's last LDS reads are done: the exchange buffer becomes the row stage
        if (valid) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                // value i = c R3 + j1 is bin 256 j1 + t + T c; DC-centred position o; staged for the thread that owns o
                const int o = (256 * (i % R3) + t + T * (i / R3) + N / 2) & (N - 1);
                buf[(o / EPL) * S + (o % EPL)] = dbv[i];
            }
            pss_post::row_sync<true>();
            pss_post::post_row_staged<EPL, 1, true>(buf, slot, t, N - 4, nullptr, fl, t, phase,
                                                    reinterpret_cast<float4 *>(post + (size_t)f * (N - 4)), row_lo, row_hi, f);
        }
        pss_r16::frame_sync<true>();         // the next frame overwrites the exchange buffer
    }
}

}  // namespace pss_sp

#pragma clang fp contract(off)   // back to pss_post.h's mode for the rest of the unit
