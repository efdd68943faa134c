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

// ---- demodulate_ssb in ONE kernel for frames of 8192 / 16 384 samples (round 3; LDS plan of round 4) -------------------------------
// signal_processing.py:203-216: z = lfilter(taps, 1, x) (complex 65-tap FIR: scipy -> np.convolve -> cblas_zdotu, of which only the
// real part survives :205), hilbert(real(z)), its real part, / max|.| * 0.95, int16.  k_ssb_fir used to write real(z) as float64
// (1.07 GB at cfg 3) for k_hilbert_xl to read straight back; here the FIR runs inside the transform kernel:
//   1. the frame's I samples (taps and window are real: Q never reaches the real part) are staged in LDS AS THE FLOAT32 THEY ARE, element i
//      at i + i / 16 (rows of 16 + one pad: a thread's window of consecutive elements is conflict-free at a lane stride of 17) — half the
//      bytes of the float64 staging of round 3, which leaves room beside it for half of the outputs (below);
//   2. thread t computes its 16 CONSECUTIVE outputs 16 t .. 16 t + 15, four at a time, in the accumulation order of OpenBLAS's
//      zdot_microk_haswell (8 accumulators (a, p) per output over 8 steps of 8 elements: element j = 8 it + 2 a + p; the pairs
//      (acc[0] + acc[1]) + (acc[2] + acc[3]) per p; c0 + c1; fma(x[i], tap[0], .)) — k_ssb_fir's tree bit for bit — with the taps as
//      scalar operands (kernarg); a window element is read and converted to float64 ONCE per pass and feeds every accumulator that
//      takes it in that step (an accumulator's own sequence of fused multiply-adds is what the tree fixes, not the order between
//      accumulators); the 64 outputs whose windows are shorter than 65 taps (their own zdot shapes) by the lanes of wavefront 0 with
//      the predicated per-lane tree (zdot_re_skx_lane, as k_ssb_edge);
//   3. outputs 0..7 of a thread go straight into region A of the exchange buffer (beside the staged samples: nothing is held in
//      registers for them), outputs 8..15 wait in 16 VGPRs for the barrier behind the last window read and then take region B, the
//      dead staging area; both regions hold a thread's 8 doubles at 9 t + c (conflict-free 8-byte stores), and the transform's input
//      layout x[t + T q] reads them back with one lane address per region.
// Round 3's form (float64 staging, all 16 outputs in registers through the barrier, eleven converted window elements live per step)
// needed 166 / 184 VGPRs on a budget of 128: 38 / 56 spilled, 148 / 220 bytes of scratch per lane, 2.5 x the algorithmic HBM bytes.
// From there on k_hilbert_xl<LOG_R4, 2>: forward transform, one-sided mask, inverse transform, frame peak, normalisation, PCM.
struct SsbTaps {
    double rev[72];   // rev[j] = taps[64 - j] (65 taps, zeros behind them)
    double fwd[72];   // taps[j]
};

__device__ __forceinline__ int pad17(int i) { return i + (i >> 4); }

template <int LOG_R4>
struct SsbLay {
    using C = pss_xl::CfgX<LOG_R4>;
    static constexpr int T = C::T, N = C::N;
    static constexpr int FLT = N + N / 16;                 // float32 slots of the staged samples
    static constexpr int HALF = 9 * T;                     // doubles of one output region: thread t's 8 outputs at 9 t + c
    static constexpr int A0 = HALF + 16;                   // region A (doubles from the buffer's start); region B = [0, HALF)
    static_assert((size_t)FLT * 4 <= (size_t)HALF * 8, "region B covers the staged samples it replaces");
    static constexpr int EXD = (A0 + HALF > C::EXD) ? A0 + HALF : C::EXD;
    static constexpr size_t LDS = (size_t)EXD * sizeof(double);
};

template <int LOG_R4>
__global__ __launch_bounds__(256 << LOG_R4, 4) void k_ssb_hilbert_xl(const float2 *__restrict__ iq, double *out, const double2 *__restrict__ tw,
                                                                     long n_rows, unsigned *__restrict__ pcm, SsbTaps taps)
{
    __shared__ double red[2][16];
    __shared__ double ltaps[72];
    using C = pss_xl::CfgX<LOG_R4>;
    using LY = SsbLay<LOG_R4>;
    constexpr int T = C::T, N = C::N, T2 = C::T2, R4 = C::R4;
    extern __shared__ __align__(16) unsigned char smem[];
    double *ex = reinterpret_cast<double *>(smem);
    float *exf = reinterpret_cast<float *>(smem);
    const int t = threadIdx.x;
    const double2 w1 = tw[t], w2 = tw[(size_t)(t % T2) * 16], w3 = tw[(size_t)(t % R4) * 256];
    if (t < 72) ltaps[t] = taps.fwd[t];
    constexpr double INV_N = 1.0 / (double)N;
    for (long f = blockIdx.x; f < n_rows; f += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rx = pss_xl::make_rsrc(iq + (size_t)f * N, N * 8);
        const __amdgpu_buffer_rsrc_t ro = pss_xl::make_rsrc(out + (out ? (size_t)f * N : 0), out ? N * 8 : 0);
        const __amdgpu_buffer_rsrc_t rp = pss_xl::make_rsrc(pcm + (pcm ? (size_t)f * N : 0), pcm ? N * 4 : 0);
        // (tt: the thread index behind an opaque statement per frame — with the plain index the compiler computes the per-lane LDS addresses
        // of the staging accesses, the windows and the edge tree once, ahead of the frame loop, and spills them)
        int tt = t;
        asm volatile("" : "+v"(tt));
        // element t + T q of the frame sits at pad17(t) + q (T + T / 16): one lane address, the rest immediate offsets
        float *sb = exf + pad17(tt);
        constexpr int QS = T + T / 16;
        // 1. stage the in-phase samples
        __syncthreads();                          // the previous frame's last exchange is done with the buffer
#pragma unroll
        for (int q = 0; q < 16; q++) sb[q * QS] = pss_xl::buf_load_f2(rx, tt * 8, T * q * 8).x;
        __syncthreads();
        // 2. the FIR: outputs 16 t + c, c = 0..15 (threads 0..3 own the left edge: wavefront 0's lanes compute it below)
        double o8[8];                              // outputs 8..15 (region B is still the staging area)
        double *pa = ex + LY::A0 + 9 * tt;         // this thread's 8 slots of region A
        if (t >= 4) {
            const float *xw = exf + 17 * (tt - 4);   // x[16 t - 64 + e] at xw[e + (e >> 4)]
            // four outputs per pass (c = 4 g .. 4 g + 3): 32 accumulators.  The passes are unrolled (every window address is the lane base
            // + an immediate) and fenced from each other: interleaved by the scheduler they would need four sets of accumulators.
#pragma unroll
            for (int g = 0; g < 4; g++) {
                int z = 0;
                asm volatile("" : "+s"(z));         // opaque zero: the taps are re-read per pass, not hoisted into 130 loop-invariant SGPRs
                double acc[4][4][2];
#pragma unroll
                for (int it = 0; it < 8; it++) {
#pragma unroll
                    for (int e = 0; e < 11; e++) {
                        const int k = 4 * g + 8 * it + e;
                        const double xv = (double)xw[k + (k >> 4)];
#pragma unroll
                        for (int a = 0; a < 4; a++)
#pragma unroll
                            for (int pp = 0; pp < 2; pp++) {
                                const int o = e - 2 * a - pp;
                                if (o >= 0 && o < 4)
                                    acc[o][a][pp] = __fma_rn(xv, taps.rev[z + 8 * it + 2 * a + pp], it == 0 ? 0.0 : acc[o][a][pp]);
                            }
                    }
                }
                const double y64 = taps.rev[z + 64];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const double c0 = __dadd_rn(__dadd_rn(acc[o][0][0], acc[o][1][0]), __dadd_rn(acc[o][2][0], acc[o][3][0]));
                    const double c1 = __dadd_rn(__dadd_rn(acc[o][0][1], acc[o][1][1]), __dadd_rn(acc[o][2][1], acc[o][3][1]));
                    const int k = 4 * g + 64 + o;
                    const double r = __fma_rn((double)xw[k + (k >> 4)], y64, __dadd_rn(c0, c1));
                    if (g < 2) pa[4 * g + o] = r;
                    else o8[4 * (g - 2) + o] = r;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double edge = 0.0;
        if (t < 64) {
            edge = pss::zdot_re_skx_lane([&](int j) { return (double)exf[pad17(j)]; }, [&](int j) { return ltaps[tt - j]; }, tt + 1);   // (tt: the tree's predicates stay inside the loop)
            if ((tt & 15) < 8) ex[LY::A0 + 9 * (tt >> 4) + (tt & 15)] = edge;      // element tt = output (tt & 15) of thread tt >> 4
        }
        __syncthreads();                          // every window has been read: the staging area becomes region B
        if (t >= 4) {
#pragma unroll
            for (int c = 0; c < 8; c++) ex[9 * tt + c] = o8[c];
        }
        if (t < 64 && (tt & 15) >= 8) ex[9 * (tt >> 4) + (tt & 15) - 8] = edge;
        __syncthreads();
        double2 u1 = w1, u2 = w2, u3 = w3;
        asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        double2 v[16], y[16];
        {
            // element t + T q: output (t & 15) of thread (t >> 4) + (T / 16) q — one region and one lane address per thread
            const double *src = ex + ((tt & 8) ? 0 : LY::A0) + 9 * (tt >> 4) + (tt & 7);
            const double *src_hi = src + 9 * (T / 16) * 8;      // q >= 8 (beyond the offset field's reach from src at N = 16384)
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = make_double2((q < 8 ? src : src_hi)[9 * (T / 16) * (q & 7)], 0.0);
        }
        __syncthreads();                          // ... before the first exchange writes into the buffer
        pss_xl::xl_core<LOG_R4>(v, ex, u1, u2, u3, tt, [&](int, int j, int k, double2 X) { y[j + (16 / R4) * k] = X; });
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = mask_conj(y[q], q, tt);
        asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        double m = 0.0;
        pss_xl::xl_core<LOG_R4>(v, ex, u1, u2, u3, tt, [&](int, int j, int k, double2 W) {
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


// ---- demodulate_ssb in one kernel, hilbert() evaluated as a REAL transform pair (round 4) ------------------------------------------------
// hilbert(r) of a real row r of N samples is fft -> one-sided mask -> ifft, and demodulate_ssb keeps the real part (signal_processing.py:205-213).
// Both transforms are evaluated here the way real-input FFTs are (pocketfft's r2c does the same for scipy): with z[m] = r[2m] + i r[2m+1] and
// Z = FFT_M(z), M = N / 2,
//     E[k] = (Z[k] + conj Z[M-k]) / 2,   O[k] = (Z[k] - conj Z[M-k]) / 2i,   X[k] = E[k] + W_N^k O[k],   conj X[M-k] = E[k] - W_N^k O[k]
// is the spectrum of r; the analytic signal's spectrum is Y = h X (h = 1, 2 .. 2, 1, 0 .. 0), and the REAL part of ifft(Y) is the inverse real
// transform of its Hermitian part H[k] = (Y[k] + conj Y[N-k]) / 2 (= X[k]: the mask doubles, the Hermitian part halves — both exact):
//     Z'[k] = (H[k] + conj H[M-k]) / 2 + i (H[k] - conj H[M-k]) / 2 conj W_N^k,   z' = IFFT_M(Z'),   real(hilbert(r))[2m], [2m+1] = Re, Im z'[m].
// Two M-point transforms instead of two N-point ones: half the butterflies and half the exchange traffic, and — what matters most — a frame of
// 16 384 samples is a workgroup of 512 threads with 70 KB of LDS, so TWO frames are resident per CU and one's LDS / HBM phases run under the
// other's arithmetic (the N-point form is one 1024-thread workgroup per CU whose phases can only follow each other: DESIGN.md §4.2).
// The pair (k, M - k) lives in threads t and T - t: one component-wise exchange fetches the partner, the rest is local to a thread.
//   1. the frame's I samples are staged in LDS as float32, unpadded (element i at i);
//   2. thread t computes the FIR outputs 2 (t + T q), 2 (t + T q) + 1, q < 16 — exactly its transform inputs z[t + T q], so the FIR output
//      never passes through LDS — a pair at a time in zdot_microk_haswell's order (k_ssb_fir's tree bit for bit; 16 accumulators), the pair's
//      66-element window by 33 conflict-free 8-byte reads; the 64 outputs with shorter windows (z[0..31]: q = 0, t < 32) by the predicated
//      per-lane tree;
//   3. FFT_M, partner exchange + the algebra above, FFT_M on the conjugate, frame peak, normalisation, float64 audio + int16 PCM.
__device__ __forceinline__ double2 rf_cq(int q)     // exp(-2 pi i q / 32) = W_N^(T q), N = 32 T
{
    constexpr double tab[16][2] = {
        {1.0, -0.0},
        {0.9807852804032304, -0.19509032201612825},
        {0.9238795325112867, -0.3826834323650898},
        {0.8314696123025452, -0.5555702330196022},
        {0.7071067811865476, -0.7071067811865475},
        {0.5555702330196023, -0.8314696123025452},
        {0.38268343236508984, -0.9238795325112867},
        {0.19509032201612833, -0.9807852804032304},
        {6.123233995736766e-17, -1.0},
        {-0.1950903220161282, -0.9807852804032304},
        {-0.3826834323650897, -0.9238795325112867},
        {-0.555570233019602, -0.8314696123025455},
        {-0.7071067811865475, -0.7071067811865476},
        {-0.8314696123025453, -0.5555702330196022},
        {-0.9238795325112867, -0.3826834323650899},
        {-0.9807852804032304, -0.1950903220161286}};
    return make_double2(tab[q][0], tab[q][1]);
}

#ifndef PSS_EXP_SSB_WAVES    // timing experiment: -DPSS_EXP_SSB_WAVES=2 gives the kernel 256 VGPRs (no spill) at one resident frame per CU
#define PSS_EXP_SSB_WAVES 4
#endif
template <int LOG_R4>   // of the M-point transform: M = 4096 << LOG_R4, frames of N = 2 M samples
__global__ __launch_bounds__(256 << LOG_R4, PSS_EXP_SSB_WAVES) void k_ssb_rfft(const float2 *__restrict__ iq, double *out, const double2 *__restrict__ tw,
                                                               long n_rows, unsigned *__restrict__ pcm, SsbTaps taps)
{
    __shared__ double red[2][8];
    __shared__ double ltaps[72];
    using C = pss_xl::CfgX<LOG_R4>;
    constexpr int T = C::T, M = C::N, N = 2 * M, T2 = C::T2, R4 = C::R4;
    static_assert((size_t)N * 4 <= C::LDS && 16 * T + 1 <= C::EXD, "staged samples and the partner exchange fit the exchange buffer");
    extern __shared__ __align__(16) unsigned char smem[];
    double *ex = reinterpret_cast<double *>(smem);
    float *exf = reinterpret_cast<float *>(smem);
    const int t = threadIdx.x;
    if (t < 72) ltaps[t] = taps.fwd[t];
    constexpr double INV_M = 1.0 / (double)M;
#ifdef PSS_EXP_STAGGER   // timing experiment: the second workgroup of a CU starts late, so that the two are in different phases
    if ((blockIdx.x / 256) & 1)
        for (int i = 0; i < PSS_EXP_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
#endif
    for (long f = blockIdx.x; f < n_rows; f += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rx = pss_xl::make_rsrc(iq + (size_t)f * N, N * 8);
        const __amdgpu_buffer_rsrc_t ro = pss_xl::make_rsrc(out + (out ? (size_t)f * N : 0), out ? N * 8 : 0);
        const __amdgpu_buffer_rsrc_t rp = pss_xl::make_rsrc(pcm + (pcm ? (size_t)f * N : 0), pcm ? N * 4 : 0);
        int tt = t;
        asm volatile("" : "+v"(tt));              // per-lane addresses are recomputed per frame, not hoisted out of the loop and spilled
        // 1. stage the in-phase samples (the first four bytes of every complex64)
        __syncthreads();                          // the previous frame's last exchange is done with the buffer
#pragma unroll
        for (int q = 0; q < 32; q++) exf[tt + T * q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, tt * 8, T * q * 8, 0));
        __syncthreads();
        // 2. the FIR, straight into the transform's input registers: v[q] = (r[2 (t + T q)], r[2 (t + T q) + 1])
        double2 v[16];
        // the pipeline's state: the five sample pairs and eight taps of the step about to run (primed with the first window's first step)
        const double y64h = taps.rev[64];
        float2 c0, c1, c2, c3, c4;
        double tc[8];
        {
            const float *x0 = (tt < 32) ? exf : exf + 2 * tt - 64;
            c0 = *reinterpret_cast<const float2 *>(x0); c1 = *reinterpret_cast<const float2 *>(x0 + 2); c2 = *reinterpret_cast<const float2 *>(x0 + 4);
            c3 = *reinterpret_cast<const float2 *>(x0 + 6); c4 = *reinterpret_cast<const float2 *>(x0 + 8);
            int z0 = 0;
            asm volatile("" : "+s"(z0));
#pragma unroll
            for (int k = 0; k < 8; k++) tc[k] = taps.rev[z0 + k];
        }
#pragma unroll 1
        for (int g = 0; g < 4; g++) {
            int z = 0;
            asm volatile("" : "+s"(z));           // opaque zero: the taps are re-read per group, not hoisted into 130 loop-invariant SGPRs
            const float *xg = exf + 2 * tt - 64 + 8 * T * g;          // window of the group's first pair: r[2 (t + 4 T g) - 64 ..]
            double2 r4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                // (q = 0, t < 32: the window starts before the frame — those 64 outputs are the edge tree's below; read in bounds)
                const float *xp = (i == 0) ? ((g == 0 && tt < 32) ? exf : xg) : xg + 2 * T * i;
                double acc[2][4][2];
                // software pipeline over the steps of the windows: the NEXT step's samples (LDS) and taps (kernarg segment: scalar loads) — at a
                // window's last step: the next window's first step — are requested before this step's sixteen FMAs and waited for at the top of the
                // next step.  Requested at the step that uses them, every step began with an s_waitcnt lgkmcnt(0) on loads issued two
                // instructions earlier: an LDS + scalar-cache round trip per step, 128 of them per thread and frame.
                const float *xnext = (i < 3) ? xg + 2 * T * (i + 1) : (g < 3 ? xg + 8 * T : xg);   // (after the last window: any in-bounds address, unused)
                float2 nx7 = c4;
#pragma unroll
                for (int it = 0; it < 8; it++) {
                    const double xs[9] = {(double)c0.x, (double)c0.y, (double)c1.x, (double)c1.y, (double)c2.x,
                                          (double)c2.y, (double)c3.x, (double)c3.y, (double)c4.x};
                    if (it == 7) nx7 = c4;
                    __builtin_amdgcn_sched_barrier(0);
                    float2 n0 = c4, n1, n2, n3, n4;
                    double tn[8];
                    if (it < 7) {
                        asm volatile("" : "+s"(z));   // (a step's 8 taps = 16 SGPRs, two steps' worth live: all 65 at once spill)
                        n1 = *reinterpret_cast<const float2 *>(xp + 8 * it + 10);
                        n2 = *reinterpret_cast<const float2 *>(xp + 8 * it + 12);
                        n3 = *reinterpret_cast<const float2 *>(xp + 8 * it + 14);
                        n4 = *reinterpret_cast<const float2 *>(xp + 8 * it + 16);
#pragma unroll
                        for (int k = 0; k < 8; k++) tn[k] = taps.rev[z + 8 * it + 8 + k];
                    } else {
                        asm volatile("" : "+s"(z));
                        n0 = *reinterpret_cast<const float2 *>(xnext);
                        n1 = *reinterpret_cast<const float2 *>(xnext + 2);
                        n2 = *reinterpret_cast<const float2 *>(xnext + 4);
                        n3 = *reinterpret_cast<const float2 *>(xnext + 6);
                        n4 = *reinterpret_cast<const float2 *>(xnext + 8);
#pragma unroll
                        for (int k = 0; k < 8; k++) tn[k] = taps.rev[z + k];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int pp = 0; pp < 2; pp++) {
                            const double y = tc[2 * a + pp];
                            acc[0][a][pp] = __fma_rn(xs[2 * a + pp], y, it == 0 ? 0.0 : acc[0][a][pp]);
                            acc[1][a][pp] = __fma_rn(xs[2 * a + pp + 1], y, it == 0 ? 0.0 : acc[1][a][pp]);
                        }
                    c0 = n0; c1 = n1; c2 = n2; c3 = n3; c4 = n4;
#pragma unroll
                    for (int k = 0; k < 8; k++) tc[k] = tn[k];
                }
                const float2 nx = nx7;
                const double y64 = y64h;
                double r[2];
#pragma unroll
                for (int o = 0; o < 2; o++) {
                    const double c0 = __dadd_rn(__dadd_rn(acc[o][0][0], acc[o][1][0]), __dadd_rn(acc[o][2][0], acc[o][3][0]));
                    const double c1 = __dadd_rn(__dadd_rn(acc[o][0][1], acc[o][1][1]), __dadd_rn(acc[o][2][1], acc[o][3][1]));
                    r[o] = __fma_rn((double)(o ? nx.y : nx.x), y64, __dadd_rn(c0, c1));
                }
                r4[i] = make_double2(r[0], r[1]);
            }
            // v[] without run-time indexing: every group shifts it down by four and takes the last four slots
#pragma unroll
            for (int c = 0; c < 12; c++) v[c] = v[c + 4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[12 + i] = r4[i];
        }
        if (t < 32) {   // the left edge: outputs 2 t and 2 t + 1 of 2 t + 1 and 2 t + 2 terms (their own zdot shapes)
            const int i0 = 2 * tt;
            v[0].x = pss::zdot_re_skx_lane([&](int j) { return (double)exf[j]; }, [&](int j) { return ltaps[i0 - j]; }, i0 + 1);
            v[0].y = pss::zdot_re_skx_lane([&](int j) { return (double)exf[j]; }, [&](int j) { return ltaps[i0 + 1 - j]; }, i0 + 2);
        }
        __syncthreads();                          // every window has been read: the buffer is the transforms' now
        // tw = exp(-2 pi i k / N), k < N: the M-point transform's bases are its even entries.  Fetched per frame, here (the table is 256 KB,
        // L2-resident): held across the frame loop and the FIR these 16 registers are spilled (37 -> 21 spilled VGPRs at N = 16384; the
        // kernel's time is the same either way, 1.39 ms min in a paired A/B — the scratch traffic was never what paced it)
        int t3 = tt;
        asm volatile("" : "+v"(t3));
        double2 u1 = tw[2 * t3], u2 = tw[(size_t)(t3 % T2) * 32], u3 = tw[(size_t)(t3 % R4) * 512];
        asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        // 3. Z = FFT_M(z): bin t + T q in y[q]
        double2 y[16];
        pss_xl::xl_core<LOG_R4>(v, ex, u1, u2, u3, tt, [&](int, int j, int k, double2 X) { y[j + (16 / R4) * k] = X; });
        // 4. the partner Z[M - k]: thread T - t, slot 15 - q (thread 0: itself, slot (16 - q) mod 16 — its slot 0 is duplicated behind the rows)
        double oi[16];
        {
            double *wr = ex + tt;
            const double *rd = ex + T - tt;
#pragma unroll
            for (int q = 0; q < 16; q++) wr[q * T] = y[q].x;
            if (t == 0) ex[16 * T] = y[0].x;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const double px = rd[(15 - q) * T], zx = y[q].x;
                y[q].x = 0.5 * (zx + px);         // Re E
                oi[q] = -0.5 * (zx - px);         // Im O
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 16; q++) wr[q * T] = y[q].y;
            if (t == 0) ex[16 * T] = y[0].y;
            __syncthreads();
            double2 wn = tw[t3];                  // W_N^t
            asm volatile("" : "+v"(wn.x), "+v"(wn.y));
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const double py = rd[(15 - q) * T], zy = y[q].y;
                const double2 E = make_double2(y[q].x, 0.5 * (zy - py)), O = make_double2(0.5 * (zy + py), oi[q]);
                const double2 W = pss_r16::cmul(wn, rf_cq(q));                       // W_N^(t + T q)
                const double2 WO = pss_r16::cmul(W, O);
                const double2 X = make_double2(E.x + WO.x, E.y + WO.y);              // spectrum of the real row at bin k
                const double2 Xc = make_double2(E.x - WO.x, E.y - WO.y);             // conj of it at bin M - k
                // one-sided mask (2; bins 0 and M: 1) and the Hermitian part of the masked spectrum (halves it again; bins 0 / M keep it)
                const bool ends = (q == 0) && (t == 0);
                const double hk = ends ? 1.0 : 2.0, hh = ends ? 1.0 : 0.5;
                const double2 H = make_double2(X.x * hk * hh, X.y * hk * hh), Hc = make_double2(Xc.x * hk * hh, Xc.y * hk * hh);
                const double2 E2 = make_double2(0.5 * (H.x + Hc.x), 0.5 * (H.y + Hc.y));
                const double2 D = make_double2(0.5 * (H.x - Hc.x), 0.5 * (H.y - Hc.y));
                const double2 O2 = pss_r16::cmul(D, make_double2(W.x, -W.y));
                v[q] = make_double2(E2.x - O2.y, -(E2.y + O2.x));                    // conj(Z'[k]): ifft = conj(fft(conj .)) / M
            }
            __syncthreads();                      // the partner reads are done before the next exchange writes
        }
        // the three twiddle bases again, from the (L2-resident) table: held across the partner exchange above they were ten spilled VGPRs
        {
            int t4 = tt;
            asm volatile("" : "+v"(t4));
            u1 = tw[2 * t4]; u2 = tw[(size_t)(t4 % T2) * 32]; u3 = tw[(size_t)(t4 % R4) * 512];
            asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        }
        // 5. z' = IFFT_M(Z'): sample pair m = t + T q in y[q]
        double m = 0.0;
        pss_xl::xl_core<LOG_R4>(v, ex, u1, u2, u3, tt, [&](int, int j, int k, double2 W) {
            const int q = j + (16 / R4) * k;
            y[q] = make_double2(W.x * INV_M, -(W.y * INV_M));
            m = nanmax(nanmax(m, fabs(y[q].x)), fabs(y[q].y));
        });
        // 6. frame peak, normalisation, float64 audio and int16 stereo PCM (samples 2 m, 2 m + 1 side by side)
        for (int off = 32; off > 0; off >>= 1) m = nanmax(m, __shfl_xor(m, off));
        const int par = (int)(((f - blockIdx.x) / gridDim.x) & 1);
        if ((t & 63) == 0) red[par][t >> 6] = m;
        __syncthreads();
        m = red[par][0];
#pragma unroll
        for (int w = 1; w < T / 64; w++) m = nanmax(m, red[par][w]);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const double a0 = normalise95(y[q].x, m), a1 = normalise95(y[q].y, m);
            typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
            const v4u_t pk = {(unsigned)__double2loint(a0), (unsigned)__double2hiint(a0), (unsigned)__double2loint(a1), (unsigned)__double2hiint(a1)};
            // (row offset in the per-lane operand, constant soffset: the store-data hazard note in k_hilbert_xl)
            __builtin_amdgcn_raw_buffer_store_b128(pk, ro, tt * 16 + T * q * 16, 0, 0);      // out == NULL: zero-sized resource, dropped
            const pss_xl::v2u_t pp = {pcm_pair(a0), pcm_pair(a1)};
            __builtin_amdgcn_raw_buffer_store_b64(pp, rp, tt * 8 + T * q * 8, 0, 0);
        }
    }
}

}  // namespace pss_hil
