// pss_fft_xl.h — register-resident FOUR-stage FFT for N = 8192 and 16384 (N = 16 x 16 x 16 x R4, R4 = 2 / 4): one workgroup
// of T = N / 16 threads per frame, 16 points per thread, nothing but the frame itself and its result ever touches HBM
// (12 algorithmic bytes per sample for a spectrum).  The previous big-N kernel parked the radix-R pre-pass in a float64
// scratch (16 B / sample written and read back) and completed every dB row by R interleaved 4-byte store passes:
// 44 B / sample moved and 4.7 x the algorithmic traffic at N = 16384.
//
// Decimation in frequency, three radix-16 stages and one radix-R4 stage (a, b, c = the sample index inside the
// sub-sequence of stage 1, 2, 3; T2 = T / 16 = 16 R4):
//   stage 1  thread t = a             x[a + T q], q < 16           16-point DFT over q,  times W_N^(a r)     -> y_r[a]
//   stage 2  thread (r, b)            y_r[b + T2 q2]               16-point DFT over q2, times W_T^(b r2)    -> z_{r r2}[b]
//   stage 3  thread (r, r2, c)        z_{r r2}[c + R4 q3]          16-point DFT over q3, times W_T2^(c r3)   -> u_{r r2 r3}[c]
//   stage 4  thread rho, j < 16/R4    u_g[0 .. R4-1], g = rho + T j = r + 16 r2 + 256 r3     radix-R4        -> X[4096 k + g]
// For fixed (j, k) the T threads of the frame hold T CONSECUTIVE bins, so rows leave in full lines straight from registers.
// The three exchanges go through LDS one COMPONENT at a time (real parts, then imaginary parts): N doubles (+ padding)
// = 140 KB at N = 16384, which is what lets the whole frame stay on the CU.  Layouts are padded so that every
// ds_read_b64 / ds_write_b64 of a 32-lane group touches 32 different bank pairs (derivations at each exchange).
// Twiddles: one table load per stage and thread (W^t), the 15 powers by repeated multiplication (relative error ~1e-15,
// far below the 1e-4 contract) — 45 per-thread table loads per frame would be 5 x the frame's own bytes.
#pragma once
#include <hip/hip_runtime.h>

#include "pss_fft_r16.h"

namespace pss_xl {

using pss_r16::brev;
using pss_r16::cmul;
using pss_r16::fft_reg;

template <int LOG_R4>
struct CfgX {
    static constexpr int R4 = 1 << LOG_R4;
    static constexpr int T = 256 * R4;              // threads per frame
    static constexpr int N = 16 * T;
    static constexpr int T2 = 16 * R4;              // T / 16
    static constexpr int E2 = T2 + R4;              // row stride of exchange 2 (doubles): groups of R4 lanes land R4 apart mod 32
#ifdef PSS_EXP_OLDPAD
    static constexpr int P3 = 16 * 272 + 32 / R4;
#else
    // plane stride of exchange 3 (doubles): 17-double rows; a ds_write_b64 is served 16 consecutive lanes at a time against 32 banks
    // (16 eight-byte slots): 16 / R4 values of r2 x R4 planes -> slots 17 r2 + P3 c, all different iff P3 = 16 / R4 (mod 16)
    static constexpr int P3 = 16 * 272 + 16 / R4;
#endif
    static constexpr int EXD = (R4 * P3 > 256 * E2) ? R4 * P3 : 256 * E2;   // doubles (>= 16 T for exchange 1)
    static constexpr size_t LDS = (size_t)EXD * sizeof(double);
};

// v[brev(r)] (the in-register DFT leaves X[r] there) *= w^r, r = 1..15, in place; the powers come from two interleaved
// product chains (odd / even exponents) so that only a few complex temporaries are live — the kernel runs 16 wavefronts
// per CU and must fit 128 VGPRs with its 16 complex float64 points.
__device__ __forceinline__ void twiddle_powers(double2 (&v)[16], double2 w)
{
    const double2 w2 = cmul(w, w);
    double2 po = w, pe = w2;               // w^(2j+1), w^(2j+2)
    v[brev(1, 4)] = cmul(v[brev(1, 4)], po);
    v[brev(2, 4)] = cmul(v[brev(2, 4)], pe);
#pragma unroll
    for (int j = 1; j < 7; j++) {
        po = cmul(po, w2);
        pe = cmul(pe, w2);
        v[brev(2 * j + 1, 4)] = cmul(v[brev(2 * j + 1, 4)], po);
        v[brev(2 * j + 2, 4)] = cmul(v[brev(2 * j + 2, 4)], pe);
    }
    po = cmul(po, w2);
    v[brev(15, 4)] = cmul(v[brev(15, 4)], po);
}

using pss_r16::v2u_t;
using pss_r16::make_rsrc;
using pss_r16::buf_load_f2;
using pss_r16::buf_load_f64;
using pss_r16::buf_store_f32;

// keeps the compiler from hoisting the next block's loads (and their registers) above this point
__device__ __forceinline__ void sched_fence() { asm volatile("" ::: "memory"); }

// One component-wise exchange: every thread writes its 16 values (real parts, then imaginary parts) and gathers 16 others.
// wr(i) / rd(i) return the LDS location of the i-th value written / read as (base pointer, COMPILE-TIME element offset), so
// that every access is one VGPR base + an immediate offset (the DS offset field holds 65535 bytes; bases are shared by as
// many accesses as that reach allows).  Four workgroup barriers.
// (Measured: the last barrier moved in FRONT of the next exchange's writes — a transform stage later, when every wavefront
// has long finished reading — changes nothing at N = 4096 / 8192 and costs 5 % at 16 384.)
// WAVE_LOCAL: every value a thread reads was written by a lane of its OWN wavefront (exchange 2: thread (r, b) -> thread (16 r + r2, c) keeps r,
// and r is constant over a wavefront or a power-of-two part of one), into a part of the buffer no other wavefront touches during this
// exchange: a wavefront's LDS instructions execute in order, so the four workgroup barriers become compiler fences and the wavefronts of a
// frame stop marching in step for a third of the transform (the caller separates the exchange from its neighbours' buffer-wide accesses).
template <bool WAVE_LOCAL = false>
__device__ __forceinline__ void xsync()
{
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
template <bool WAVE_LOCAL = false, class Wr, class Rd>
__device__ __forceinline__ void exchange(double2 (&v)[16], Wr wr, Rd rd)
{
    double im[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { *wr(i) = v[brev(i, 4)].x; im[i] = v[brev(i, 4)].y; }
    xsync<WAVE_LOCAL>();
#pragma unroll
    for (int i = 0; i < 16; i++) v[i].x = *rd(i);
    xsync<WAVE_LOCAL>();
#pragma unroll
    for (int i = 0; i < 16; i++) *wr(i) = im[i];
    xsync<WAVE_LOCAL>();
#pragma unroll
    for (int i = 0; i < 16; i++) v[i].y = *rd(i);
    xsync<WAVE_LOCAL>();
}

// One frame whose 16 stage-1 inputs x[t + T q] (windowed, as complex float64) are in v[].  w1 = W_N^t, w2 = W_T^(t % T2),
// w3 = W_T2^(t % R4) (per-thread constants).  emit(i, j, k, X) receives the thread's 16 results: X = bin 4096 k + T j + t.
// Twelve workgroup barriers; the last exchange ends with one after its reads, so the caller adds none between frames.
template <int LOG_R4, class Emit>
__device__ __forceinline__ void xl_core(double2 (&v)[16], double *ex, double2 w1, double2 w2, double2 w3, int t, Emit emit)
{
    using C = CfgX<LOG_R4>;
    constexpr int R4 = C::R4, T = C::T, T2 = C::T2, E2 = C::E2, P3 = C::P3;
    const int r_s = t / T2, b_s = t % T2;   // stage-2 role (r, b)
    const int g3 = t / R4, c_s = t % R4;    // stage-3 role (g = 16 r + r2, c)
    // ---- stage 1
    fft_reg<16>(v);
    twiddle_powers(v, w1);
    // exchange 1: L1[r][a] (row stride T).  Writes: a wavefront stores 64 consecutive doubles of one row.  Reads: thread
    // (r, b) takes L1[r][b + T2 q2]: 32 lanes = 32 consecutive doubles (R4 = 2: one row; R4 = 4: half a row).
    {
        double *w_lo = ex + t, *w_hi = ex + 8 * T + t;      // rows 0..7 / 8..15 (8 T doubles = 64 KiB at N = 16384)
        double *rb = ex + r_s * T + b_s;
        exchange(v, [&](int r) { return (r < 8 ? w_lo : w_hi) + (r & 7) * T; }, [&](int q) { return rb + T2 * q; });
    }
    // ---- stage 2
    fft_reg<16>(v);
    twiddle_powers(v, w2);
    // exchange 2: L2[r][r2][b] (row stride E2 = T2 + R4).  Writes: lanes of one r store consecutive b.  Reads: thread (g, c)
    // takes L2[g][c + R4 q3]: a 32-lane group is 32 / R4 rows x R4 doubles, rows E2 = R4 (mod 32) apart -> 32 distinct
    // bank pairs.
    {
        // rows 16 r .. 16 r + 15 are written and read by the T2 threads of ONE r only (g3 / 16 = t / T2 = r): wave-local — no workgroup
        // barrier inside; the one behind it keeps exchange 3's buffer-wide writes away from wavefronts still reading here
        // (exchange 1 ended with a workgroup barrier after its reads)
        double *wb = ex + (r_s * 16) * E2 + b_s, *rb = ex + g3 * E2 + c_s;
#ifdef PSS_EXP_XL_BARRIERS
        exchange(v, [&](int r2) { return wb + r2 * E2; }, [&](int q) { return rb + R4 * q; });
#else
        exchange<true>(v, [&](int r2) { return wb + r2 * E2; }, [&](int q) { return rb + R4 * q; });
        __syncthreads();
#endif
    }
    // ---- stage 3
    fft_reg<16>(v);
    if constexpr (R4 > 1) twiddle_powers(v, w3);   // R4 = 1 (N = 4096): stage 3 is the last one, its twiddles are all 1
    // exchange 3: L3[c][r3][r2][r] with 17-double (r2) rows, 272-double r3 blocks and planes P3 apart.  Writes: a 32-lane
    // group is 32 / R4 values of r2 x R4 planes at fixed r: offsets 17 r2 + (32 / R4) c (mod 32) are all different.
    // Reads: thread rho takes g = rho + T j in bin order (r fastest): 32 lanes = two rows of 16 consecutive doubles, 17 apart
    // (one shared bank pair in 32: a single extra LDS cycle).  Value i = j R4 + k is u_g[k]; g = t + T j only moves r3
    // (T = 256 R4), i.e. a compile-time offset of 272 R4 j.
    {
        double *wb = ex + c_s * P3 + 17 * (g3 & 15) + (g3 >> 4);
        double *rb0 = ex + 272 * (t >> 8) + 17 * ((t >> 4) & 15) + (t & 15);
        double *rb1 = rb0 + (R4 > 2 ? 2 * P3 : 0);          // planes 2, 3 (beyond the offset field's reach from rb0)
        exchange(v, [&](int r3) { return wb + 272 * r3; },
                 [&](int i) { return ((i % R4) < 2 ? rb0 : rb1) + ((i % R4) & 1) * P3 + 272 * R4 * (i / R4); });
    }
    // ---- stage 4: radix-R4 butterflies over c; bin = 4096 k + g
#pragma unroll
    for (int j = 0; j < 16 / R4; j++) {
        double2 b[R4];
#pragma unroll
        for (int k = 0; k < R4; k++) b[k] = v[j * R4 + k];
        fft_reg<R4>(b);
#pragma unroll
        for (int k = 0; k < R4; k++) emit(j * R4 + k, j, k, b[brev(k, LOG_R4)]);
    }
}

// compute_fft (signal_processing.py:243-264) for frames of N = 4096 * R4 points (R4 = 1, 2, 4): Hamming window (np.hamming,
// float64 table), transform, fftshift, 10 log10(|X|^2 + 1e-10) as float32.  WINDOW = false: an unwindowed transform.
// SCAN: the inline scanner's slice (pyspecsdr.py:2542-2552) — unwindowed, plus the slice's peak and the number of bins within
// 20 dB of it; db may then be NULL.  The dB values wait in the (then idle) exchange buffer for the peak to be known.
template <int LOG_R4, bool WINDOW, bool SCAN = false, bool EXACT = false>
__global__ __launch_bounds__(256 << LOG_R4, 4) void k_spectrum_xl(const float2 *__restrict__ iq, float *__restrict__ db,
                                                                  const double2 *__restrict__ tw, const double *__restrict__ win,
                                                                  long n_frames, float *__restrict__ peak, double *__restrict__ bw,
                                                                  int *__restrict__ count, double bin_hz, int flags)
{
    using C = CfgX<LOG_R4>;
    constexpr int T = C::T, N = C::N, T2 = C::T2, R4 = C::R4;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float red_f[16];
    __shared__ int red_i[16];
    __shared__ uint4 l10[SCAN ? 16 : 1];       // the float32 log10 model's coefficient sets (pss_npf32.h): LDS copy for the scanner's exact rows
    double *ex = reinterpret_cast<double *>(smem);
    const int t = threadIdx.x;
    if (SCAN && t < 16) l10[t] = pss::L10_PACK[t];
    if (SCAN) __syncthreads();
    const double2 w1 = tw[t];                                  // W_N^t
    const double2 w2 = tw[(size_t)(t % T2) * 16];              // W_T^b = W_N^(16 b)
    const double2 w3 = tw[(size_t)(t % R4) * 256];             // W_T2^c = W_N^(256 c)
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(win, WINDOW ? N * 8 : 0);
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
#ifdef PSS_EXP_SPEC_L2IQ   // timing experiment: every load hits the cache
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(iq + (size_t)(f & 15) * N, N * 8);
#else
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(iq + (size_t)f * N, N * 8);
#endif
#ifdef PSS_EXP_SPEC_NOSTORE   // timing experiment: the row is computed and dropped by the bounds check
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(db + (size_t)f * N, (flags & 0x100) ? N * 4 : 0);
#else
        const __amdgpu_buffer_rsrc_t ro = make_rsrc(db ? db + (size_t)f * N : nullptr, db ? N * 4 : 0);   // no rows wanted: every store dropped
#endif
        // the 45 twiddle powers are loop-invariant and the compiler would compute them once, park them in scratch memory
        // (180 VGPRs) and reload them every frame; recomputing them from the three bases is cheaper than that traffic
        double2 u1 = w1, u2 = w2, u3 = w3;
        asm volatile("" : "+v"(u1.x), "+v"(u1.y), "+v"(u2.x), "+v"(u2.y), "+v"(u3.x), "+v"(u3.y));
        double2 v[16];
#pragma unroll
        for (int h = 0; h < 2; h++) {              // two batches of eight: 16 sample + 16 window loads in flight would not fit
            float2 s[8];
            double wv[8];
#pragma unroll
            for (int q = 0; q < 8; q++) s[q] = buf_load_f2(rx, t * 8, T * (8 * h + q) * 8);
#pragma unroll
            for (int q = 0; q < 8; q++) wv[q] = WINDOW ? buf_load_f64(rw, t * 8, T * (8 * h + q) * 8) : 1.0;
#pragma unroll
            for (int q = 0; q < 8; q++) v[8 * h + q] = make_double2((double)s[q].x * wv[q], (double)s[q].y * wv[q]);
            sched_fence();
        }
        float lmax = -INFINITY;
        float *held = reinterpret_cast<float *>(ex);           // SCAN: this thread's 16 dB values at held[i * T + t]
        xl_core<LOG_R4>(v, ex, u1, u2, u3, t, [&](int i, int j, int k, double2 X) {
            // fftshift; bin 4096 k + T j + t: T consecutive bins per store instruction
            float d;
            if constexpr (SCAN) d = (flags & pss_r16::FLAG_SCAN_EXACT) ? pss::scan_db_np(X.x, X.y, l10) : pss_r16::db_of_fast(pss_r16::power_of(X));
            else d = EXACT ? pss_r16::db_of_exact(pss_r16::power_of(X)) : pss_r16::db_of_fast(pss_r16::power_of(X));
#ifdef PSS_EXP_SPEC_NODB
            d = (float)X.x + (float)X.y;
#endif
            buf_store_f32(ro, t * 4, ((4096 * k + T * j + N / 2) & (N - 1)) * 4, d);
            if (SCAN) { held[i * T + t] = d; lmax = fmaxf(lmax, d); }
        });
        if (SCAN) {
            // per-frame peak and 20-dB-down bin count (pyspecsdr.py:2546-2552), as k_spectrum_r16 computes them
            float m = lmax;
            for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if ((t & 63) == 0) red_f[t >> 6] = m;
            __syncthreads();
            m = red_f[0];
#pragma unroll
            for (int wv = 1; wv < T / 64; wv++) m = fmaxf(m, red_f[wv]);
            const float thr = m - 20.0f;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) cnt += held[i * T + t] > thr;
            for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            if ((t & 63) == 0) red_i[t >> 6] = cnt;
            __syncthreads();                       // also: every thread has read its held[] before the next frame's exchange writes
            if (t == 0) {
                cnt = 0;
                for (int wv = 0; wv < T / 64; wv++) cnt += red_i[wv];
                peak[f] = m;
                if (bw) bw[f] = (double)cnt * bin_hz;
                if (count) count[f] = cnt;
            }
        }
    }
}

}  // namespace pss_xl
