// pss_spec_post.h — compute_fft AND the caller's post-process of the same frame in ONE kernel, for 1024-point frames (the headline batch):
// signal_processing.py:243-264 (window, float64 transform, fftshift, 10 log10(|X|^2 + 1e-10)) followed by pyspecsdr.py:2278-2283 (5-tap
// smoothing, median clamp) and the resampling of the clamped row to the display width (np.interp, pyspecsdr.py:1379-1383) — everything the
// display accumulators need of a row — while the frame's float64 dB values are still in the registers of the wavefront that transformed it.
//
// Why: as two kernels (k_spectrum_r16<D64> -> k_post_sel<double>) the float64 rows are written (8 bytes per bin) and read straight back: 1.07 GB of
// the step's 3.7 GB at 65 536 x 1024.  A frame of 1024 points IS one wavefront in both kernels (16 values per lane), so the row can stay where it
// is: the 16 float64 dB values of a lane go through the frame's (then idle) exchange buffer into k_post_sel's per-thread-consecutive layout and
// pss_post::post_row_staged runs on them unchanged — the same smoothing, the same select on the keys' high words, the same clamp, extremes and
// np.interp: the same bits.  What reaches HBM per frame: the dB row ONCE — float32 (db32: compute_fft's float64 value rounded once, the
// 1e-4-relative contract of the spectrum output: 4 bytes per bin) and / or float64 (db64: the reference's own row type) —, 16 bytes of
// extremes and disp_w float64 resampled values.
// Included by pss_fft.hip behind pss_fft_r16.h and pss_post.h.
#pragma once

namespace pss_sp {

using namespace pss_r16;

#ifndef PSS_EXP_FUSE_WAVES
#define PSS_EXP_FUSE_WAVES 2     // minimum workgroups per CU the register allocation has to allow: two wavefronts per SIMD = at most 256 VGPRs
#endif
template <bool ROW32, bool ROW64>
__global__ __launch_bounds__(256, PSS_EXP_FUSE_WAVES) void k_spectrum_post(const float2 *__restrict__ iq, float *__restrict__ db32, double *__restrict__ db64,
                                                       const double2 *__restrict__ tw, const double *__restrict__ win, long n_frames,
                                                       double *__restrict__ row_lo, double *__restrict__ row_hi, double *__restrict__ vals,
                                                       int disp_w)
{
    using C = Cfg<2>;
    constexpr int R3 = C::R3, T = C::T, N = C::N, FPW = C::FPW;     // 4, 64, 1024, 4
    static_assert(T == 64, "a frame is one wavefront");
    constexpr int EPL = 16;
    using PC = pss_post::PostCfg<EPL, double>;
    static_assert((size_t)(T + 1) * PC::S * sizeof(double) <= (size_t)C::EX * sizeof(double2), "the staged row fits the frame's exchange buffer");
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex_all = reinterpret_cast<double2 *>(smem);
#ifdef PSS_EXP_FUSE_LDS
    constexpr size_t EXS = PSS_EXP_FUSE_LDS / FPW / sizeof(double2);
    double2 *tw2 = ex_all;
#else
    constexpr size_t EXS = C::EX;
    double2 *tw2 = ex_all + (size_t)FPW * C::EX;
#endif
#ifdef PSS_EXP_POST_PRIO    // timing experiment: this kernel's user priority beside the backward pass (reset at the end)
    __builtin_amdgcn_s_setprio(PSS_EXP_POST_PRIO);
#endif
    const int tid = threadIdx.x;
    const int fl = __builtin_amdgcn_readfirstlane(tid / T);   // frame slot = wavefront of the workgroup
    const int t = tid % T;
    double2 *ex = ex_all + (size_t)fl * EXS;
    double *stage = reinterpret_cast<double *>(ex);
    double2 tw1[16];
    double w[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)t * k2];
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) w[n2] = win[t + T * n2];
#ifndef PSS_EXP_FUSE_LDS
    if (tid < R3 * 16) {
        const int m1 = tid / 16, j2 = tid % 16;
        tw2[C::tw2_slot(tid)] = tw[(size_t)(m1 * j2) * 16];
    }
#endif
    __syncthreads();
    const long groups = (n_frames + FPW - 1) / FPW;
    float2 nx[16];
    auto fetch = [&](long g) {
        const long f = g * FPW + fl;
        const float2 *x = iq + (size_t)(f < n_frames ? f : 0) * N;
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) nx[n2] = x[t + T * n2];
    };
    if ((long)blockIdx.x < groups) fetch(blockIdx.x);
    int phase = 0;
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long f = g * FPW + fl;
        const bool valid = f < n_frames;                      // wave-uniform
        double2 v[16];
#pragma unroll
#ifdef PSS_EXP_FUSE_NOFFT
        for (int n2 = 0; n2 < 16; n2++) v[n2] = make_double2((double)nx[n2].x, (double)nx[n2].y);
#else
        for (int n2 = 0; n2 < 16; n2++) v[n2] = make_double2((double)nx[n2].x * w[n2], (double)nx[n2].y * w[n2]);
#endif
        if (g + gridDim.x < groups) fetch(g + gridDim.x);
        // row stores through buffer resources over the workgroup's FPW rows (frames past the end fall outside and are dropped): unconditional
        // in the instruction stream, see k_spectrum_r16
        const long f_first = g * FPW;
        const long rows_here = n_frames - f_first < FPW ? n_frames - f_first : FPW;
        const __amdgpu_buffer_rsrc_t ro32 = make_rsrc((ROW32 && db32) ? db32 + (size_t)f_first * N : nullptr, (ROW32 && db32) ? (unsigned)(rows_here * N * 4) : 0u);
        const __amdgpu_buffer_rsrc_t ro64 = make_rsrc((ROW64 && db64) ? db64 + (size_t)f_first * N : nullptr, (ROW64 && db64) ? (unsigned)(rows_here * N * 8) : 0u);
        const int lane_el = fl * N + t;
        double dbv[16];
        auto emit = [&](int i, int k, double2 X) {
            const double d64 = db64_of_exact(power_of(X));
            const int e0 = ((k - t) + N / 2) & (N - 1);        // fftshift; compile-time constant per call
            if constexpr (ROW64) buf_store_f64(ro64, lane_el * 8, e0 * 8, d64);
            if constexpr (ROW32) buf_store_f32(ro32, lane_el * 4, e0 * 4, (float)d64);
            dbv[i] = d64;
        };
#ifdef PSS_EXP_FUSE_NOFFT   // timing experiment (results wrong): no transform, the dB values are the windowed samples
#pragma unroll
        for (int i = 0; i < 16; i++) emit(i, 256 * (i % R3) + t + T * (i / R3), v[i]);
#else
        r16_core<2, true>(v, ex, tw1, tw2, t, emit);
#endif
        frame_sync<true>();                                    // every lane has read its stage-3 operands: the exchange buffer is free
        // the row into k_post_sel's staging layout: element e (after fftshift) at stage[(e / EPL) * S + e % EPL]
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int c = i / R3, j1 = i % R3;
            const int e = (((256 * j1 + T * c) + N / 2) & (N - 1)) + t;     // 64 consecutive elements over the wavefront
            stage[(e / EPL) * PC::S + (e % EPL)] = dbv[i];
        }
        frame_sync<true>();
#ifndef PSS_EXP_FUSE_NOPOST  // timing experiment (results wrong): no post-process
        if (valid) {
            int slot[PC::Q];
#pragma unroll
            for (int j = 0; j < PC::Q; j++) slot[j] = 0;       // (only used for materialised post-processed rows: none here)
            pss_post::post_row_staged<EPL, 1, true, double>(stage, slot, t, N - 4, nullptr, fl, t, phase, nullptr, row_lo, row_hi, f, nullptr, vals, disp_w);
        }
#endif
        frame_sync<true>();                                    // the next frame's stage 1 overwrites the buffer
    }
#ifdef PSS_EXP_POST_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

}  // namespace pss_sp
