// pss_hilbert.h — scipy.signal.hilbert along rows (analytic signal): X = fft(x); X *= h (h[0] = h[N/2] = 1, h[1..N/2-1] = 2,
// h[N/2+1..] = 0); ifft(X) (scipy/signal/_signaltools.py:2318 ff.), which demodulate_ssb applies to the real part of its
// FIR output (signal_processing.py:205, :210).  Included by pss_fft.hip.
//
// Both transforms of a row run back to back in ONE kernel, the row never leaving the CU in between: the register FFTs
// (pss_fft_r16.h: N = 256..4096, pss_fft_xl.h: N = 8192, 16384) take their input as x[t + T q] (thread t, q < 16) and
// deliver bin t + T q' to the same thread — the forward transform's output IS the next transform's input layout, so
// the one-sided mask is a per-register constant (h depends on q' only, except for bins 0 and N/2 in thread 0) and the
// inverse transform is the same code on the conjugate (ifft(Z) = conj(fft(conj(Z))) / N).
#pragma once
#include <hip/hip_runtime.h>

#include "pss_fft_r16.h"
#include "pss_fft_xl.h"

namespace pss_hil {

// h[t + T q] applied to the bin a thread holds in slot q, then the conjugate (input of the second forward transform)
__device__ __forceinline__ double2 mask_conj(double2 X, int q, int t)
{
    const double f = q == 0 ? (t == 0 ? 1.0 : 2.0) : (q < 8 ? 2.0 : (q == 8 ? (t == 0 ? 1.0 : 0.0) : 0.0));
    return make_double2(X.x * f, -(X.y * f));
}

// |re| maximum of a frame as the bit pattern of a non-negative double (orders like the value), into mx[f]
__device__ __forceinline__ void track_max(double m, unsigned long long *mx)
{
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off);
        m = (o != o || o > m) ? o : m;           // NaN propagates, as np.max does
    }
    if ((threadIdx.x & 63) == 0) atomicMax(mx, (unsigned long long)__double_as_longlong(m));
}

// np.int16(v * 32767) (io_manager.py:26), as pss_device.h pcm16 (that header's __constant__ tables keep it out of this unit)
__device__ __forceinline__ unsigned pcm_pair(double a)
{
    const double v = __dmul_rn(a, 32767.0);
    const unsigned s = (unsigned)(unsigned short)((v != v) ? (short)0 : (short)(int)v);
    return s | (s << 16);                                   // mono_to_stereo: left = right
}

// NaN-propagating maximum (np.max) of non-negative values
__device__ __forceinline__ double nanmax(double a, double b) { return (b != b || b > a) ? b : a; }

// OUT = 2 tail, shared by both kernels: samples / np.max(np.abs(samples)) * 0.95 (signal_processing.py:213-216) with the frame
// peak mx, float64 audio (optional) and int16 stereo PCM
__device__ __forceinline__ double normalise95(double re, double mx) { return __dmul_rn(__ddiv_rn(re, mx), 0.95); }

// N = 256 * R3 (R3 = 1..16).  OUT: 0 = complex128 analytic signal to `out`, 1 = real part only (in place allowed) + frame peak,
// 2 = demodulate_ssb's tail in the same kernel: real part / frame peak * 0.95 -> `out` (float64 audio, may be NULL) and `pcm`
template <int LOG_R3, int OUT>
__global__ __launch_bounds__(256) void k_hilbert_r16(const double *x, double *out, const double2 *__restrict__ tw,
                                                     long n_rows, unsigned long long *__restrict__ mxbits, unsigned *__restrict__ pcm)
{
    __shared__ double red[2][4];             // OUT = 2: wave maxima, double-buffered by loop parity
    using C = pss_r16::Cfg<LOG_R3>;
    constexpr int R3 = C::R3, T = C::T, N = C::N, FPW = C::FPW;
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex_all = reinterpret_cast<double2 *>(smem);
    double2 *tw2 = ex_all + (size_t)FPW * C::EX;
    const int tid = threadIdx.x, fl = tid / T, t = tid % T;
    double2 *ex = ex_all + (size_t)fl * C::EX;
    double2 tw1[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)t * k2];
    if (tid < R3 * 16) tw2[C::tw2_slot(tid)] = tw[(size_t)((tid / 16) * (tid % 16)) * 16];
    __syncthreads();
    constexpr bool WL = T <= 64;
    constexpr double INV_N = 1.0 / (double)N;
    const long groups = (n_rows + FPW - 1) / FPW;
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long f = g * FPW + fl;
        const bool valid = f < n_rows;
        const double *row = x + (size_t)(valid ? f : 0) * N;
        double2 v[16], y[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = make_double2(row[t + T * q], 0.0);
        // bin k = 256 j1 + t + T c sits in slot (k - t) / T = c + (16 / R3) j1 of the next transform's input
        pss_r16::r16_core<LOG_R3, WL>(v, ex, tw1, tw2, t, [&](int i, int, double2 X) { y[(i / R3) + (16 / R3) * (i % R3)] = X; });
        pss_r16::frame_sync<WL>();
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = mask_conj(y[q], q, t);
        double m = 0.0;
        pss_r16::r16_core<LOG_R3, WL>(v, ex, tw1, tw2, t, [&](int i, int, double2 W) {
            const int q = (i / R3) + (16 / R3) * (i % R3);
            const double re = W.x * INV_N, im = -(W.y * INV_N);      // conj(fft(conj(Z))) / N
            if (OUT == 2) y[q].x = re;
            else if (valid) {
                if (OUT == 0) reinterpret_cast<double2 *>(out)[(size_t)f * N + t + T * q] = make_double2(re, im);
                else out[(size_t)f * N + t + T * q] = re;
            }
            m = nanmax(m, fabs(re));
        });
        if (OUT == 1 && mxbits) {
            // the T threads of a frame: T >= 64 whole wavefronts, T < 64 a slice of one (reduce over the slice only)
            if (T >= 64) { if (valid) track_max(m, &mxbits[f]); }
            else {
                for (int off = T / 2; off > 0; off >>= 1) m = nanmax(m, __shfl_xor(m, off));
                if (valid && t == 0) atomicMax(&mxbits[f], (unsigned long long)__double_as_longlong(m));
            }
        }
        if (OUT == 2) {
            // frame peak: over the frame's T lanes — a slice of one wavefront, or T / 64 whole wavefronts through LDS
            for (int off = (T < 64 ? T : 64) / 2; off > 0; off >>= 1) m = nanmax(m, __shfl_xor(m, off));
            if (T > 64) {
                const int par = (int)(((g - blockIdx.x) / gridDim.x) & 1);
                if ((tid & 63) == 0) red[par][tid >> 6] = m;
                __syncthreads();
                m = red[par][fl * (T / 64)];
#pragma unroll
                for (int w = 1; w < T / 64; w++) m = nanmax(m, red[par][fl * (T / 64) + w]);
            }
            if (valid) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const double a = normalise95(y[q].x, m);
                    if (out) out[(size_t)f * N + t + T * q] = a;
                    if (pcm) pcm[(size_t)f * N + t + T * q] = pcm_pair(a);
                }
            }
        }
        pss_r16::frame_sync<WL>();
    }
}

// N = 4096 * R4 (R4 = 2, 4): one workgroup of T = N / 16 threads per frame
// (OUT is a launch argument here, not a template parameter: the epilogue is a few stores per value; three instantiations per length
// were 75 KB of code object)
template <int LOG_R4>
__global__ __launch_bounds__(256 << LOG_R4, 4) void k_hilbert_xl(const double *x, double *out,
                                                                 const double2 *__restrict__ tw, long n_rows,
                                                                 unsigned long long *__restrict__ mxbits, unsigned *__restrict__ pcm, int OUT)
{
    __shared__ double red[2][16];            // OUT = 2: wave maxima, double-buffered by loop parity
    using C = pss_xl::CfgX<LOG_R4>;
    constexpr int T = C::T, N = C::N, T2 = C::T2, R4 = C::R4;
    extern __shared__ __align__(16) unsigned char smem[];
    double *ex = reinterpret_cast<double *>(smem);
    const int t = threadIdx.x;
    const double2 w1 = tw[t], w2 = tw[(size_t)(t % T2) * 16], w3 = tw[(size_t)(t % R4) * 256];
    constexpr double INV_N = 1.0 / (double)N;
    for (long f = blockIdx.x; f < n_rows; f += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rx = pss_xl::make_rsrc(x + (size_t)f * N, N * 8);
        const __amdgpu_buffer_rsrc_t ro = pss_xl::make_rsrc(out + (out ? (size_t)f * N * (OUT == 0 ? 2 : 1) : 0), out ? N * (OUT == 0 ? 16 : 8) : 0);
        const __amdgpu_buffer_rsrc_t rp = pss_xl::make_rsrc(pcm + (pcm ? (size_t)f * N : 0), (OUT == 2 && pcm) ? N * 4 : 0);
        double2 u1 = w1, u2 = w2, u3 = w3;
        asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        double2 v[16], y[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = make_double2(pss_xl::buf_load_f64(rx, t * 8, T * q * 8), 0.0);
        // bin 4096 k + T j + t sits in slot j + (16 / R4) k
        pss_xl::xl_core<LOG_R4>(v, ex, u1, u2, u3, t, [&](int, int j, int k, double2 X) { y[j + (16 / R4) * k] = X; });
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = mask_conj(y[q], q, t);
        asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        double m = 0.0;
        pss_xl::xl_core<LOG_R4>(v, ex, u1, u2, u3, t, [&](int, int j, int k, double2 W) {
            const int q = j + (16 / R4) * k;
            y[q] = make_double2(W.x * INV_N, -(W.y * INV_N));
            m = nanmax(m, fabs(y[q].x));
        });
        if (OUT == 0) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
                v4u_t pk = {(unsigned)__double2loint(y[q].x), (unsigned)__double2hiint(y[q].x), (unsigned)__double2loint(y[q].y), (unsigned)__double2hiint(y[q].y)};
                // The row offset rides in the per-lane offset, NOT in the scalar-offset operand: behind a 128-bit buffer store whose
                // soffset is an SGPR the compiler's hazard recogniser leaves no wait state before a VALU write of the data registers
                // (LLVM: "this hazard only exists if soffset is not a register"), and gfx950 does sample the data late — the
                // `m = nanmax(...)` code that followed the last store overwrote lo(y[15].y) in lanes 12-15 of each row of 16 on a
                // cold launch (first use of the kernel in a process: 16-48 wrong samples, relative error 1e-7).  With a constant
                // soffset the recogniser inserts the wait states itself.  tools/check_store_hazard.py scans the ISA for the pattern.
                __builtin_amdgcn_raw_buffer_store_b128(pk, ro, t * 16 + T * q * 16, 0, 0);
            }
        } else if (OUT == 1) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                pss_xl::v2u_t pk = {(unsigned)__double2loint(y[q].x), (unsigned)__double2hiint(y[q].x)};
                __builtin_amdgcn_raw_buffer_store_b64(pk, ro, t * 8, T * q * 8, 0);
            }
        }
        if (OUT == 2) {
            // frame peak over the workgroup (= the frame), then demodulate_ssb's normalisation and the int16 conversion from registers
            for (int off = 32; off > 0; off >>= 1) m = nanmax(m, __shfl_xor(m, off));
            const int par = (int)(((f - blockIdx.x) / gridDim.x) & 1);
            if ((t & 63) == 0) red[par][t >> 6] = m;
            __syncthreads();
            m = red[par][0];
#pragma unroll
            for (int w = 1; w < T / 64; w++) m = nanmax(m, red[par][w]);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const double a = normalise95(y[q].x, m);
                pss_xl::v2u_t pk = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a)};
                __builtin_amdgcn_raw_buffer_store_b64(pk, ro, t * 8, T * q * 8, 0);       // out == NULL: zero-sized resource, dropped
                __builtin_amdgcn_raw_buffer_store_b32(pcm_pair(a), rp, t * 4, T * q * 4, 0);
            }
        }
        if (OUT == 1 && mxbits) track_max(m, &mxbits[f]);
    }
}

// ---- demodulate_ssb in ONE kernel for frames of 8192 / 16 384 samples (round 3) ---------------------------------------------------
// signal_processing.py:203-216: z = lfilter(taps, 1, x) (complex 65-tap FIR: scipy -> np.convolve -> cblas_zdotu, of which only the
// real part survives :205), hilbert(real(z)), its real part, / max|.| * 0.95, int16.  k_ssb_fir used to write real(z) as float64
// (1.07 GB at cfg 3) for k_hilbert_xl to read straight back; here the FIR runs inside the transform kernel:
//   1. the frame's I samples (taps and window are real: Q never reaches the real part) are staged as float64 in the exchange buffer,
//      element i at i + i / 16 (rows of 16 + one pad: a thread's window of consecutive elements is conflict-free at a lane stride of 17);
//   2. thread t computes its 16 CONSECUTIVE outputs 16 t .. 16 t + 15, two at a time, in the accumulation order of OpenBLAS's
//      zdot_microk_haswell (8 accumulators (a, p) per output over 8 steps of 8 elements: element j = 8 it + 2 a + p; the pairs
//      (acc[0] + acc[1]) + (acc[2] + acc[3]) per p; c0 + c1; fma(x[i], tap[0], .)) — k_ssb_fir's tree bit for bit — with the taps as
//      scalar operands (kernarg) and a sliding window of 9 inputs per step; the 64 outputs whose windows are shorter than 65 taps
//      (their own zdot shapes) by the lanes of wavefront 0 with the predicated per-lane tree (zdot_re_skx_lane, as k_ssb_edge);
//   3. the outputs go back into the (now free) staging area and are re-read in the transform's input layout x[t + T q].
// From there on k_hilbert_xl<LOG_R4, 2>: forward transform, one-sided mask, inverse transform, frame peak, normalisation, PCM.
struct SsbTaps {
    double rev[72];   // rev[j] = taps[64 - j] (65 taps, zeros behind them)
    double fwd[72];   // taps[j]
};

__device__ __forceinline__ int pad17(int i) { return i + (i >> 4); }

template <int LOG_R4>
__global__ __launch_bounds__(256 << LOG_R4, 4) void k_ssb_hilbert_xl(const float2 *__restrict__ iq, double *out, const double2 *__restrict__ tw,
                                                                     long n_rows, unsigned *__restrict__ pcm, SsbTaps taps)
{
    __shared__ double red[2][16];
    __shared__ double ltaps[72];
    using C = pss_xl::CfgX<LOG_R4>;
    constexpr int T = C::T, N = C::N, T2 = C::T2, R4 = C::R4;
    static_assert(N + N / 16 <= C::EXD, "the padded staging area must fit the exchange buffer");
    extern __shared__ __align__(16) unsigned char smem[];
    double *ex = reinterpret_cast<double *>(smem);
    const int t = threadIdx.x;
    const double2 w1 = tw[t], w2 = tw[(size_t)(t % T2) * 16], w3 = tw[(size_t)(t % R4) * 256];
    if (t < 72) ltaps[t] = taps.fwd[t];
    constexpr double INV_N = 1.0 / (double)N;
    for (long f = blockIdx.x; f < n_rows; f += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rx = pss_xl::make_rsrc(iq + (size_t)f * N, N * 8);
        const __amdgpu_buffer_rsrc_t ro = pss_xl::make_rsrc(out + (out ? (size_t)f * N : 0), out ? N * 8 : 0);
        const __amdgpu_buffer_rsrc_t rp = pss_xl::make_rsrc(pcm + (pcm ? (size_t)f * N : 0), pcm ? N * 4 : 0);
        // (tt: the thread index behind an opaque statement per frame — with the plain index the compiler computes the ~200 per-lane LDS
        // addresses of the staging accesses, the windows and the edge tree once, ahead of the frame loop, and spills them: 196 VGPRs)
        int tt = t;
        asm volatile("" : "+v"(tt));
        // element t + T q of the frame sits at pad17(t) + q (T + T / 16): one lane address, the rest immediate offsets
        double *sb = ex + pad17(tt);
        constexpr int QS = T + T / 16;
        // 1. stage the in-phase samples
        __syncthreads();                          // the previous frame's last exchange is done with the buffer
#pragma unroll
        for (int q = 0; q < 16; q++) sb[q * QS] = (double)pss_xl::buf_load_f2(rx, tt * 8, T * q * 8).x;
        __syncthreads();
        // 2. the FIR: outputs 16 t + c, c = 0..15 (threads 0..3 own the left edge: wavefront 0's lanes compute it below)
        double o16[16];
        if (t >= 4) {
            const double *xw = ex + 17 * (tt - 4);   // x[16 t - 64 + e] at xw[e + (e >> 4)]
            // four outputs per pass (c = 4 g .. 4 g + 3), a ROLLED loop (unrolled, the scheduler interleaved the passes and spilled 194
            // VGPRs).  The FIR is LDS-bandwidth-bound: B outputs per pass read (16 / B) 8 (B + 7) window elements per thread — 576 at
            // B = 2 (measured: the fused kernel no faster than k_ssb_fir + k_hilbert_xl), 352 at B = 4 (32 accumulators: the register
            // budget's limit beside the 16 finished outputs).  o16[] stays in registers without run-time indexing: every finished output
            // shifts it down by one and takes the last slot.
#pragma unroll 1
            for (int g = 0; g < 4; g++) {
                int z = 0;
                asm volatile("" : "+s"(z));         // opaque zero: the taps are re-read per pass, not hoisted into 130 loop-invariant SGPRs
                double acc[4][4][2];
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    double xs[11];
#pragma unroll
                    for (int e = 0; e < 11; e++) { const int k = 4 * g + 8 * it + e; xs[e] = xw[k + (k >> 4)]; }
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int pp = 0; pp < 2; pp++) {
                            const double y = taps.rev[z + 8 * it + 2 * a + pp];
#pragma unroll
                            for (int o = 0; o < 4; o++)
                                acc[o][a][pp] = __fma_rn(xs[o + 2 * a + pp], y, it == 0 ? 0.0 : acc[o][a][pp]);
                        }
                }
                const int k64 = 4 * g + 64;
                const double y64 = taps.rev[z + 64];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const double c0 = __dadd_rn(__dadd_rn(acc[o][0][0], acc[o][1][0]), __dadd_rn(acc[o][2][0], acc[o][3][0]));
                    const double c1 = __dadd_rn(__dadd_rn(acc[o][0][1], acc[o][1][1]), __dadd_rn(acc[o][2][1], acc[o][3][1]));
                    const int k = k64 + o;
                    const double r = __fma_rn(xw[k + (k >> 4)], y64, __dadd_rn(c0, c1));
#pragma unroll
                    for (int c = 0; c < 15; c++) o16[c] = o16[c + 1];      // after sixteen shifts o16[c] = output 16 t + c
                    o16[15] = r;
                }
            }
        }
        double edge = 0.0;
        if (t < 64) edge = pss::zdot_re_skx_lane([&](int j) { return ex[pad17(j)]; }, [&](int j) { return ltaps[tt - j]; }, tt + 1);   // (tt: the tree's predicates stay inside the loop)
        __syncthreads();                          // every window has been read: the staging area takes the outputs
        if (t >= 4) {
#pragma unroll
            for (int c = 0; c < 16; c++) ex[17 * tt + c] = o16[c];      // pad17(16 t + c) = 17 t + c
        }
        if (t < 64) sb[0] = edge;
        __syncthreads();
        double2 u1 = w1, u2 = w2, u3 = w3;
        asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        double2 v[16], y[16];
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = make_double2(sb[q * QS], 0.0);
        __syncthreads();                          // ... before the first exchange writes into the buffer
        pss_xl::xl_core<LOG_R4>(v, ex, u1, u2, u3, t, [&](int, int j, int k, double2 X) { y[j + (16 / R4) * k] = X; });
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = mask_conj(y[q], q, t);
        asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        double m = 0.0;
        pss_xl::xl_core<LOG_R4>(v, ex, u1, u2, u3, t, [&](int, int j, int k, double2 W) {
            const int q = j + (16 / R4) * k;
            const double re = W.x * INV_N;
            y[q].x = re;
            m = nanmax(m, fabs(re));
        });
        for (int off = 32; off > 0; off >>= 1) m = nanmax(m, __shfl_xor(m, off));
        const int par = (int)(((f - blockIdx.x) / gridDim.x) & 1);
        if ((t & 63) == 0) red[par][t >> 6] = m;
        __syncthreads();
        m = red[par][0];
#pragma unroll
        for (int w = 1; w < T / 64; w++) m = nanmax(m, red[par][w]);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const double a = normalise95(y[q].x, m);
            pss_xl::v2u_t pk = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a)};
            __builtin_amdgcn_raw_buffer_store_b64(pk, ro, t * 8, T * q * 8, 0);       // out == NULL: zero-sized resource, dropped
            __builtin_amdgcn_raw_buffer_store_b32(pcm_pair(a), rp, t * 4, T * q * 4, 0);
        }
    }
}

}  // namespace pss_hil
