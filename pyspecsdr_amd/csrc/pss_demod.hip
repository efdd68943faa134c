// pss_demod.hip — demodulation kernels for gfx950 (NFM / AM / SSB), AGC power, int16 PCM.
//
// Reference lines replaced: signal_processing.py:91-116 (demodulate_nfm), :179-195 (demodulate_am),
// :198-217 (demodulate_ssb), :325-328 (measure_signal_power), :83-88 (mono_to_stereo),
// io_manager.py:25-26 (int16 conversion), pyspecsdr.py:898-919 (adjust_gain).
//
// Bit-exactness contract: this translation unit is compiled with -ffp-contract=off; every fused
// multiply-add is an explicit fma that the reference's native code (NumPy AVX-512 loops, SVML, OpenBLAS
// ddot/zdot) also executes, and the IIR recurrences use SciPy's exact un-fused association.  The IIR is a
// serial recurrence in time, so it runs ONE LANE PER FRAME (64 frames per wavefront, data staged so that
// every global access is a coalesced 512-byte row); everything else is sample-parallel.
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <functional>
#include <vector>

#include "pss_ctx.h"
#include "pss_device.h"

namespace {

using namespace pss;

constexpr int TPB = 256;
constexpr int TILE = 64;   // frames per wavefront in the lane-per-frame kernels
constexpr int EDGE = 27;   // sosfiltfilt: 3 * (2*4 + 1)

struct NfmCoef {
    Biquad s[4];
    double zi[8];
};
struct AmCoef {
    Biquad s[5];
};
struct WfmCoef {
    Biquad lp[3], pil[5], lmr[5];  // butter(5) low-pass 15 kHz, band-pass 19 kHz +- 200 Hz, band-pass 23..53 kHz
    double b0d, a1d;               // de-emphasis lfilter([1 - alpha], [1, -alpha])
};

// The 65 FIR taps travel as a by-value kernel argument (kernarg segment: wave-uniform scalar loads), not a
// __constant__ symbol: two contexts on one device may run different sample rates concurrently.
struct TapsArg {
    double fwd[65];  // taps[j]
    double rev[65];  // taps[64 - j]
    // rev[0..63] in the order the fused NFM kernel's ring FIR consumes them (pss_nfm_fused.h fir_ring): phase k (16 taps), half h = the
    // accumulator lanes l = 2h, 2h + 1 -> ya[l], ya[l+1], ya[l+4], ya[l+5], yb[l], yb[l+1], yb[l+4], yb[l+5] with ya[i] = rev[8k + i],
    // yb[i] = rev[32 + 8k + i]: one 64-byte scalar load per half phase
    double ring[64];
};

// ---------------------------------------------------------------------------------------------------
// NFM front end: discriminator (float32, signal_processing.py:94,97) + 65-tap FIR (float64, :108).
// Sample-parallel.  One work item = 1024 consecutive FIR outputs of one frame, done by a 128-thread
// workgroup (small LDS footprint -> many resident workgroups hide the HBM latency of the IQ read):
//   1. threads read IQ coalesced along time, compute the discriminator and store float64 d[] into LDS
//      (64 samples of history + 1024 new; index padded p + p/8 so that per-thread windows at a stride of
//      8 samples are bank-conflict free for ds_read_b64);
//   2. each thread computes 8 CONSECUTIVE outputs (their windows overlap, so a tap group's x values are
//      loaded once for all 8) in the exact OpenBLAS-ddot accumulation tree; the taps are wave-uniform SGPR
//      operands, fed 16 at a time (accumulator group k of the ddot kernel);
//   3. u[] goes out in natural [frame][time] layout at offset 27, and the workgroup that owns a frame's
//      first / last samples also writes scipy's odd extension (27 samples each side) around it.
// ---------------------------------------------------------------------------------------------------
constexpr int FIR_T = 128;                    // threads per workgroup
constexpr int FB = 8;                         // consecutive outputs per thread
constexpr int FIR_CH = FIR_T * FB;            // outputs per work item
constexpr int FIR_XS = (FIR_CH + 64) * 9 / 8 + 8;

__device__ __forceinline__ int xpad(int p) { return p + (p >> 3); }

// FB consecutive full-window outputs: output o uses x(o .. o+64) (time ascending), yrev[j] = taps[64-j].
// Tree (pss_device.h, ddot tree for n = 65): a5[k][l] = fma(x[32+8k+l],y[32+8k+l], fma(x[8k+l],y[8k+l],0));
// a[k][l] = a5[k][l] + a5[k][l+4]; s[l] = ((a[0][l]+a[1][l])+a[2][l])+a[3][l]; dot = (s0+s2)+(s1+s3); + tap 64.
// xb points at the thread's window start inside the padded LDS array: x(c) = xb[c + c/8].
__device__ __forceinline__ void fir65_batch(const double *__restrict__ xb, const double *__restrict__ yrev, double (&out)[FB])
{
    auto x = [&](int c) { return xb[c + (c >> 3)]; };
    double s[FB][4];
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
        double ya[8], yb[8];  // wave-uniform: scalar loads, SGPR operands
#pragma unroll
        for (int l = 0; l < 8; l++) { ya[l] = yrev[8 * k + l]; yb[l] = yrev[32 + 8 * k + l]; }
        const double *xk = xb + 9 * k;  // x(8k + c) = xk[c + c/8]
        auto xx = [&](int c) { return xk[c + (c >> 3)]; };
#pragma unroll
        for (int o = 0; o < FB; o++) {
#pragma unroll
            for (int l = 0; l < 4; l++) {
                double lo = __fma_rn(xx(o + 32 + l), yb[l], __fma_rn(xx(o + l), ya[l], 0.0));
                double hi = __fma_rn(xx(o + 36 + l), yb[l + 4], __fma_rn(xx(o + l + 4), ya[l + 4], 0.0));
                double a = __dadd_rn(lo, hi);
                s[o][l] = (k == 0) ? a : __dadd_rn(s[o][l], a);
            }
        }
    }
    const double y64 = yrev[64];
#pragma unroll
    for (int o = 0; o < FB; o++) {
        double dot = __dadd_rn(__dadd_rn(s[o][0], s[o][2]), __dadd_rn(s[o][1], s[o][3]));
        out[o] = __fma_rn(y64, x(o + 64), dot);
    }
}

__global__ __launch_bounds__(FIR_T) void k_nfm_front(const float2 *__restrict__ iq, double *__restrict__ U, int n,
                                                     long n_frames, int cpf, long Lp, float kscale, int swapped, TapsArg taps)
{
    __shared__ double xs[FIR_XS];
    __shared__ double edge[EDGE + 1];  // u[M-28..M-1] of the frame, for the right odd extension
    const int tid = threadIdx.x;
    const int M = n - 1;
    const long items = n_frames * cpf;
    for (long item = blockIdx.x; item < items; item += gridDim.x) {
        const long f = item / cpf;
        const int i0 = (int)(item % cpf) * FIR_CH;
        const float2 *x = iq + (size_t)f * n;
        double *Uf = U + (size_t)f * Lp;
        __syncthreads();
        for (int k = tid; k < FIR_CH + 64; k += FIR_T) {
            const int t = i0 - 64 + k;
            float d = 0.0f;
            if (t >= 0 && t < M) d = disc_sample(x[t + 1], x[t], kscale, swapped != 0);
            xs[xpad(k)] = (double)d;
        }
        __syncthreads();
        const int o0 = FB * tid, ibase = i0 + o0;  // first local / global output index of this thread
        auto put = [&](int i, double v) {
            Uf[EDGE + i] = v;
            if (i >= M - 1 - EDGE) edge[i - (M - 1 - EDGE)] = v;
        };
        if (ibase >= 64 && ibase < M) {  // outputs 0..63 (shorter windows) come from k_nfm_edge
            double out[FB];
            fir65_batch(xs + 9 * tid, taps.rev, out);  // xpad(8 tid + c) = 9 tid + c + c/8
#pragma unroll
            for (int o = 0; o < FB; o++)
                if (ibase + o < M) put(ibase + o, out[o]);
        }
        __syncthreads();
        // right odd extension (scipy _arraytools.odd_ext): ext[27+M+k] = 2u[M-1] - u[M-2-k]; only when the last
        // 28 outputs of the frame all have full windows (M - 28 >= 64) and live in this work item
        if (M - 1 - EDGE >= 64 && i0 + FIR_CH >= M && i0 <= M - 1 - EDGE && tid < EDGE)
            Uf[EDGE + M + tid] = __dsub_rn(__dmul_rn(2.0, edge[EDGE]), edge[EDGE - 1 - tid]);
    }
}

// Left edge of the FIR and the odd extension around it: outputs i < 64 have windows shorter than 65 samples, so
// every output has its own ddot shape — one output per lane, fully predicated (pss_device.h ddot_skx_lane).
// One 128-thread workgroup per frame.  Also covers whole frames of <= 92 samples, where np.convolve's operand
// order (M <= 65) or the right extension (M - 28 < 64) need the edge outputs too.
__global__ __launch_bounds__(128) void k_nfm_edge(const float2 *__restrict__ iq, double *__restrict__ U, int n,
                                                  long n_frames, long Lp, float kscale, int swapped, TapsArg taps)
{
    __shared__ double d[128];
    __shared__ double ltaps[72];
    __shared__ double u[128];
    const int tid = threadIdx.x;
    const int M = n - 1;
    const long f = blockIdx.x;
    const float2 *x = iq + (size_t)f * n;
    double *Uf = U + (size_t)f * Lp;
    const int ne = M < 92 ? M : 64;  // outputs computed here
    if (tid < 65) ltaps[tid] = taps.fwd[tid];
    d[tid] = (tid < M && tid < 92) ? (double)disc_sample(x[tid + 1], x[tid], kscale, swapped != 0) : 0.0;
    __syncthreads();
    if (tid < ne) {
        const int i = tid;
        double v;
        if (M <= 65)  // np.convolve does not swap its operands: the dot runs over ascending TAP index
            v = ddot_skx_lane([&](int j) { return ltaps[j]; }, [&](int j) { return d[i - j]; }, i + 1);
        else if (i < 64)  // x[0..i] against taps[i..0]
            v = ddot_skx_lane([&](int j) { return d[j]; }, [&](int j) { return ltaps[i - j]; }, i + 1);
        else              // full window (only reached for 66 <= M < 92)
            v = ddot_skx_lane([&](int j) { return d[i - 64 + j]; }, [&](int j) { return ltaps[64 - j]; }, 65);
        u[i] = v;
        Uf[EDGE + i] = v;
    }
    __syncthreads();
    // odd extension (scipy _arraytools.odd_ext): ext[p] = 2u[0] - u[27-p], ext[27+M+k] = 2u[M-1] - u[M-2-k]
    if (tid < EDGE) Uf[tid] = __dsub_rn(__dmul_rn(2.0, u[0]), u[EDGE - tid]);
    if (M < 92 && tid >= 32 && tid < 32 + EDGE) {
        const int k = tid - 32;
        Uf[EDGE + M + k] = __dsub_rn(__dmul_rn(2.0, u[M - 1]), u[M - 2 - k]);
    }
}

// ---------------------------------------------------------------------------------------------------
// NFM back end: scipy.signal.decimate(u, q) = sosfiltfilt(cheby1 sos) then [::q]  (signal_processing.py:112),
// peak normalisation (:115), stereo duplication (:116) and int16 conversion (io_manager.py:26).
// One wavefront per tile of 64 frames, lane = frame (the recurrence is serial in time; 64 frames advance in
// lock step).
//  * The four biquad sections run as a SKEWED pipeline: at step t section s works on sample t-s, so the four
//    section updates of a step are independent (4-way ILP for the ~9-cycle float64 latency) while every
//    section still sees exactly the reference's operation sequence.
//  * Inputs stream through a 2 x 32-deep register prefetch.  The forward pass reads u in natural
//    [frame][time] layout — each lane walks its own row with 16-byte loads — and writes y_fwd TRANSPOSED
//    ([tile][time][lane], 512-byte rows); the backward pass reads that back in reverse.
// ---------------------------------------------------------------------------------------------------
constexpr int IIR_CH = 32;

template <bool B121, int CH, class Load, class LoadChunk, class Out>
__device__ __forceinline__ void iir4_pass(const NfmCoef &c, double (&z)[8], long T, Load load, LoadChunk loadc, Out out)
{
    auto sec = [&](int s, double x) {
        return (B121 && s > 0) ? biquad_step_121(c.s[s], x, z[2 * s], z[2 * s + 1])
                               : biquad_step(c.s[s], x, z[2 * s], z[2 * s + 1]);
    };
    double p0, p1, p2;
    // fill the pipeline (T > 27 always)
    p0 = sec(0, load(0));
    { double t1 = sec(1, p0); p0 = sec(0, load(1)); p1 = t1; }
    { double t2 = sec(2, p1); double t1 = sec(1, p0); p0 = sec(0, load(2)); p2 = t2; p1 = t1; }
    // One steady-state step: sections 0..3 work on samples t, t-1, t-2, t-3.  The four updates are independent;
    // they are written stage by stage (with scheduling barriers) so that dependent float64 ops are never
    // back to back — each section still performs exactly biquad_step()'s operations in its own order.
    auto step = [&](double x) {
        const double xs[4] = {x, p0, p1, p2};
        double m[4], xn[4], t[4], u[4], v[4], w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // numerator products (exact no-ops for a [1,2,1] section)
            const bool one = B121 && k > 0;
            m[k] = one ? xs[k] : __dmul_rn(c.s[k].b0, xs[k]);
            v[k] = one ? __dadd_rn(xs[k], xs[k]) : __dmul_rn(c.s[k].b1, xs[k]);
            w[k] = one ? xs[k] : __dmul_rn(c.s[k].b2, xs[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; k++) xn[k] = __dadd_rn(m[k], z[2 * k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; k++) { t[k] = __dmul_rn(c.s[k].a1, xn[k]); u[k] = __dmul_rn(c.s[k].a2, xn[k]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = __dsub_rn(v[k], t[k]); w[k] = __dsub_rn(w[k], u[k]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; k++) { z[2 * k] = __dadd_rn(v[k], z[2 * k + 1]); z[2 * k + 1] = w[k]; }
        __builtin_amdgcn_sched_barrier(0);
        p0 = xn[0]; p1 = xn[1]; p2 = xn[2];
        return xn[3];
    };
    out(0, step(load(3)));  // one single step so that the chunked part starts at an even input index
    double b0[CH], b1[CH];
    auto runc = [&](double (&b)[CH], long r) {
#pragma unroll
        for (int t = 0; t < CH; t++) out(r + t - 3, step(b[t]));
    };
    const long nfull = (T - 4) / CH;
    long r = 4;
    if (nfull > 0) loadc(b0, r);
    for (long ch = 0; ch < nfull; ch += 2) {
        if (ch + 1 < nfull) loadc(b1, r + CH);
        runc(b0, r);
        if (ch + 1 < nfull) {
            if (ch + 2 < nfull) loadc(b0, r + 2 * CH);
            runc(b1, r + CH);
        }
        r += 2 * CH;
    }
    for (r = 4 + nfull * CH; r < T; r++) out(r - 3, step(load(r)));
    // drain
    out(T - 3, sec(3, p2)); p2 = sec(2, p1); p1 = sec(1, p0);
    out(T - 2, sec(3, p2)); p2 = sec(2, p1);
    out(T - 1, sec(3, p2));
}

template <bool B121, bool WFM = false>
__global__ __launch_bounds__(TILE) void k_nfm_iir(const double *__restrict__ U, double *__restrict__ Y,
                                                  double *__restrict__ A, int n, int q, int n_out, long n_frames,
                                                  long Lp, NfmCoef c, int16_t *__restrict__ pcm,
                                                  double *__restrict__ audio)
{
    const int lane = threadIdx.x;
    const long tile = blockIdx.x;
    const long f = tile * TILE + lane;
    const long fr = f < n_frames ? f : n_frames - 1;  // masked lanes replay the last frame (results dropped)
    const int M = n - 1;
    const long L = (long)M + 2 * EDGE;
    const double *Uf = U + (size_t)fr * Lp;  // 16-byte aligned: Lp is even
    double *Yt = Y + (size_t)tile * L * TILE + lane;
    double *At = A + (size_t)tile * n_out * TILE + lane;
#define YAT(p) Yt[(size_t)(p) * TILE]
    double z[8];
    // ---- forward pass over the odd-extended u: z = zi * ext[0]
    {
        const double x0 = Uf[0];
#pragma unroll
        for (int i = 0; i < 8; i++) z[i] = __dmul_rn(c.zi[i], x0);
    }
    double ylast = 0.0;
    iir4_pass<B121, IIR_CH>(c, z, L, [&](long r) { return Uf[r]; },
              [&](double (&b)[IIR_CH], long r) {  // r is even: 16-byte loads
                  const double2 *p = reinterpret_cast<const double2 *>(Uf + r);
#pragma unroll
                  for (int t = 0; t < IIR_CH / 2; t++) { double2 v = p[t]; b[2 * t] = v.x; b[2 * t + 1] = v.y; }
              },
              [&](long r, double v) { YAT(r) = v; ylast = v; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    // ---- backward pass over the reversed sequence (input r = position L-1-r); z = zi * y_fwd[L-1].
    // Positions p < 27 are trimmed by sosfiltfilt, so only T = L - 27 inputs are run.
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = __dmul_rn(c.zi[i], ylast);
    double mx = 0.0;
    bool nan = false;
    long next = EDGE + (long)(n_out - 1) * q;  // largest kept position (p = 27 + j q)
    int j = n_out - 1;
    iir4_pass<B121, IIR_CH>(c, z, L - EDGE, [&](long r) { return YAT(L - 1 - r); },
              [&](double (&b)[IIR_CH], long r) {
#pragma unroll
                  for (int t = 0; t < IIR_CH; t++) b[t] = YAT(L - 1 - (r + t));
              },
              [&](long r, double v) {
                  if (L - 1 - r == next) {
                      At[(size_t)j * TILE] = v;
                      double av = fabs(v);
                      nan = nan || (av != av);
                      mx = av > mx ? av : mx;
                      next -= q;
                      j--;
                  }
              });
    if (nan) mx = __builtin_nan("");
    if (WFM) {  // rows are (frame, channel) pairs: k_wfm_finalize normalises both channels by their joint peak
        if (f < n_frames) audio[f] = mx;
        return;
    }
    if (f < n_frames) {
        for (int k = 0; k < n_out; k++) {
            double a = __dmul_rn(__ddiv_rn(At[(size_t)k * TILE], mx), 0.95);  // audio / max|audio| * 0.95
            if (audio) audio[(size_t)f * n_out + k] = a;
            if (pcm) {
                uint16_t s = (uint16_t)pcm16(a);
                reinterpret_cast<uint32_t *>(pcm)[(size_t)f * n_out + k] = (uint32_t)s | ((uint32_t)s << 16);
            }
        }
    }
#undef YAT
}

}  // namespace
// floats per discriminator row: the frame's n - 1 values, zero-filled up to the end: the workers fetch one chunk (24 values) past the
// last one they use, i.e. up to index n + 45

#include "pss_nfm_fused.h"
namespace {

// ---------------------------------------------------------------------------------------------------
// numpy float32 pairwise mean of |x| (KIND 1, AM: signal_processing.py:185) or |x|^2 (KIND 0, power: :327).
// The reduction tree (8192-element chunks added sequentially; inside a chunk: block 128, 8 accumulators, halves
// rounded down to multiples of 8) is generated on the host for the frame length and replayed level by level; one
// workgroup per frame.
// ---------------------------------------------------------------------------------------------------
// (k_pairwise itself follows the shared reduction helpers further down)

// ---------------------------------------------------------------------------------------------------
// AM: envelope - mean (float32) -> 5-section Butterworth band-pass, forward only, zero state (float64): k_am_grp below.
// (Rounds 1-3 also had k_am_iir, one lane per frame with the five sections as a skewed pipeline, for batches of 32 768 frames and more;
// the section-per-lane array is faster at every batch size since round 4 — 0.99 against 1.66 ms at 262 144 x 1024 — and it is gone.)
// ---------------------------------------------------------------------------------------------------
constexpr int AM_NS = 5;

__device__ __forceinline__ double dpp_row_shr1(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);  // row_shr:1, out-of-row lanes read 0
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------------
// AM band-pass (every batch size).  With one lane per frame a batch of F frames keeps only F / 64 wavefronts busy, each running 45 dependent
// float64 instructions per sample; here the five sections of one frame sit in five lanes and the recurrence runs as a systolic array: a
// wavefront carries 12 frames (5.3x more wavefronts), each lane one section's state update per sample.
// GROUP-systolic (round 4): a section hands its output to the next one in REGISTERS, eight samples at a time.  The lanes of a 16-lane
// DPP row are laid out r = 3 s + g' (section s = 0..4, frame g' = 0..2 of the row; r = 15 idle), so `row_shr:3` moves the eight outputs
// of (g', s) to (g', s + 1): 16 v_mov_b32_dpp per group-step.  The lanes r < 3 (section 0) have no source lane inside their row: with
// bound_ctrl off they keep the DPP's `old` operand — the eight input samples they just read from LDS — so the hand-off needs no select.
// At group-step j lane s works on group j - s; only section 0's LDS reads (the staged envelope - mean) and section 4's LDS writes (a
// 128-sample ring per frame) carry data.  Every section still executes exactly biquad_step()'s operations on exactly its own sample
// sequence from a (+0, +0) state (the first group-steps of a lane, before its data arrives, run on a copy of the state that is not
// committed), so the output bits are those of a plain per-frame cascade (the oracle's).
// Four wavefronts per workgroup: the recurrence, two that load and stage the even / odd input blocks, one that stores — see the kernel.
// History: round 3's k_am_sys handed whole 64-sample blocks from section to section through LDS (128 full-wavefront LDS instructions per
// block) and had ONE memory wavefront: 3.1 us per block, 0.82 ms at cfg 3.  This kernel: 2.5 us per block, 0.64 ms (one workgroup alone:
// 2.0 us, of which the 576 float64 instructions of a block are 1.05 us, the 32 ds_write_b128 of the outputs 0.35 us — ~26 clocks of
// issue each, masked or not — and the 128 DPP moves 0.1 us; NOTEBOOK.md has the ablations).
// ---------------------------------------------------------------------------------------------------
#ifndef PSS_GRP_Q
#define PSS_GRP_Q 8
#endif
constexpr int GRP_G = 12, GRP_T = 64, GRP_Q = PSS_GRP_Q;   // samples per group-step
constexpr int GRP_OFF = (5 - 1) * GRP_Q;                     // the last section runs this many samples behind the first
static_assert(GRP_T % (2 * GRP_Q) == 0 && GRP_OFF <= GRP_T, "whole pairs of group-steps per block; the output ring holds two blocks");
constexpr int GRP_ES = GRP_T + 2;          // input row stride (doubles): even -> 16-byte aligned rows for ds_read_b128
constexpr int GRP_YS = 2 * GRP_T + 2;      // output ring row stride

__device__ __forceinline__ double dpp_row_shr3_keep(double old, double src)
{
    // row_shr:3, bound_ctrl off: lanes whose source lane lies outside their row of 16 keep `old`
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x113, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x113, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void k_am_grp(const float *__restrict__ env, const float *__restrict__ mu,
                                                double *__restrict__ Yf, double *__restrict__ mxout, int n,
                                                long n_frames, AmCoef c)
{
    __shared__ __align__(16) double ebuf[2][GRP_G][GRP_ES];   // envelope - mean: input block m in ebuf[m & 1]
    __shared__ __align__(16) double ybuf[GRP_G + 1][GRP_YS];  // last section's output, sample i at i & 127; row GRP_G: where the other lanes' stores go
    const int lane = threadIdx.x & 63;
    // role of this wavefront: 0 recurrence, 1 / 2 loads of the even / odd blocks, 3 stores.  (The dispatcher puts wavefront w of the three
    // workgroups that share a CU on three different SIMDs — tools/ubench/simd_placement_probe.hip 683 25152 — so every recurrence wavefront
    // has a SIMD's issue slots to itself, next to two load / store wavefronts of the other workgroups; rotating the roles by workgroup
    // index undid that and cost 60 %.)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long f0 = (long)blockIdx.x * GRP_G;
    const long nblk = ((long)n + GRP_T - 1) / GRP_T;
    // macro-step m = 0 .. nblk: the recurrence wavefront runs the 64 / GRP_Q group-steps of input block m (block nblk: zeros, four
    // group-steps — the drain of sections 1..4); the load wavefront of block m + 1's parity stages it and requests block m + 3; the store
    // wavefront writes back the 64 outputs [64 (m - 1) - GRP_OFF, 64 m - GRP_OFF) that were complete when macro-step m - 1 ended.
    // Loads and stores sit in DIFFERENT wavefronts, and each load wavefront has ONE block in flight.  Round 3's single memory wavefront
    // kept two blocks of loads and a block of stores in flight and needed the oldest twelve loads to stage a block; the waits the compiler
    // put there were vmcnt(11) .. vmcnt(0) — everything younger drained too, i.e. the round trip of the stores just issued, every block
    // (its in-order counting does not survive this loop's bounds-checked paths; a loads-only wavefront with two blocks in flight got the
    // same).  Here vmcnt(0) is the right wait: a load wavefront waits for exactly the block it requested two macro-steps earlier, and the
    // store wavefront never waits.
    if (wave == 1 || wave == 2) {
        // ---------------- load wavefronts: lane = sample inside a block; this one owns the blocks of parity par ----------------
        const int par = wave - 1;
        float pre[GRP_G], mus[GRP_G];
#pragma unroll
        for (int gg = 0; gg < GRP_G; gg++) mus[gg] = (f0 + gg < n_frames) ? mu[f0 + gg] : 0.0f;
        auto load = [&](long blk) __attribute__((always_inline)) {
            const long i = blk * GRP_T + lane;
            if ((blk + 1) * GRP_T <= n && f0 + GRP_G <= n_frames) {
                const float *src = env + (size_t)f0 * n + i;
#pragma unroll
                for (int gg = 0; gg < GRP_G; gg++) pre[gg] = src[(size_t)gg * n];
            } else {
#pragma unroll
                for (int gg = 0; gg < GRP_G; gg++) {
                    const long ff = f0 + gg;
                    pre[gg] = (ff < n_frames && i < n) ? env[(size_t)ff * n + i] : 0.0f;
                }
            }
        };
        auto stage = [&](long blk) __attribute__((always_inline)) {     // pre[] holds block blk (blk >= nblk: zeros)
            if ((blk + 1) * GRP_T <= n && f0 + GRP_G <= n_frames) {     // a whole block of twelve frames (wave-uniform): no selects
#pragma unroll
                for (int gg = 0; gg < GRP_G; gg++) ebuf[blk & 1][gg][lane] = (double)__fsub_rn(pre[gg], mus[gg]);
                return;
            }
#pragma unroll
            for (int gg = 0; gg < GRP_G; gg++) {
                const long ff = f0 + gg;
                double e = 0.0;
                if (ff < n_frames && blk * GRP_T + lane < n) e = (double)__fsub_rn(pre[gg], mus[gg]);  // float32 subtract (:185)
                ebuf[blk & 1][gg][lane] = e;
            }
        };
        load(par);
        if (par == 0) {
            stage(0);
            load(2);
        }
        fused::lds_barrier();
        for (long m = 0; m <= nblk; m++) {
            if (((m + 1) & 1) == par) {         // block m + 1 is this wavefront's: into ebuf[par] while the recurrence reads the other buffer
                stage(m + 1);
                load(m + 3);
            }
            fused::lds_barrier();
        }
        return;
    }
    if (wave == 3) {
        // ---------------- store wavefront: lane = sample inside a 64-sample window of the output ring ----------------
        double mxl[GRP_G];
#pragma unroll
        for (int gg = 0; gg < GRP_G; gg++) mxl[gg] = 0.0;
        unsigned nanmask = 0;
        auto writeback = [&](long k) __attribute__((always_inline)) {  // outputs [64 k - OFF, 64 k - OFF + 64): complete when macro-step k has ended
            const long i = k * GRP_T - GRP_OFF + lane;
            double yv[GRP_G];
#pragma unroll
            for (int gg = 0; gg < GRP_G; gg++) yv[gg] = ybuf[gg][(int)(i & (2 * GRP_T - 1))];
#pragma unroll
            for (int gg = 0; gg < GRP_G; gg++) {
                const long ff = f0 + gg;
                if (ff < n_frames && i >= 0 && i < n) {
                    const double v = yv[gg];
                    Yf[(size_t)ff * n + i] = v;
                    const double av = fabs(v);
                    if (av != av) nanmask |= 1u << gg;
                    mxl[gg] = av > mxl[gg] ? av : mxl[gg];
                }
            }
        };
        fused::lds_barrier();
        for (long m = 0; m <= nblk; m++) {
            if (m >= 1) writeback(m - 1);
            fused::lds_barrier();
        }
        writeback(nblk);   // (the range [64 nblk - GRP_OFF, 64 nblk): complete after the drain)
#pragma unroll
        for (int gg = 0; gg < GRP_G; gg++) {
            double mm = mxl[gg];
            int nn = (nanmask >> gg) & 1;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_xor(mm, off);
                mm = o > mm ? o : mm;
                nn |= __shfl_xor(nn, off);
            }
            if (lane == 0 && f0 + gg < n_frames) mxout[f0 + gg] = nn ? __builtin_nan("") : mm;
        }
        return;
    }
    // ---------------- recurrence wavefront ----------------
    __builtin_amdgcn_s_setprio(3);                             // the SIMD's other wavefronts (loads / stores of other workgroups) fill its gaps
    const int row = lane >> 4, r = lane & 15;
    const int s = r / 3, g3 = r - 3 * s;                       // r = 15: s = 5 -> idle lane
    const int g = row * 3 + g3;
    Biquad cs = c.s[0];
#pragma unroll
    for (int k = 1; k < AM_NS; k++)
        if (s == k) cs = c.s[k];
    double z0 = 0.0, z1 = 0.0;
    double pa[GRP_Q], pb[GRP_Q];                               // this lane's outputs of the last two group-steps (alternating: no copies)
#pragma unroll
    for (int k = 0; k < GRP_Q; k++) pa[k] = pb[k] = 0.0;
    const double *const ein = &ebuf[0][g][0];                  // (every lane reads its frame's row: only section 0 uses what it read)
    // every lane stores every group-step — the last sections into their frame's ring, the others into a shared dump row: a store under
    // `if (s == 4)` is a branch around it, and a basic-block boundary per group-step keeps the scheduler from running the next group's
    // hand-off and first steps under the tail of this group's dependent chain (measured: 0.4 us of a 2.0 us block)
    double *const yout = &ybuf[s == AM_NS - 1 ? g : GRP_G][0];
    // one group-step: e[] = the frame's next eight input samples (used by section 0; the DPP overwrites the other lanes' in place),
    // pin = the previous group-step's outputs, pout = this one's; jg = global group-step index
    auto group = [&](double (&e)[GRP_Q], const double (&pin)[GRP_Q], double (&pout)[GRP_Q], long jg, bool gate) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < GRP_Q; k++) e[k] = dpp_row_shr3_keep(e[k], pin[k]);
        double a0 = z0, a1 = z1;
#pragma unroll
        for (int k = 0; k < GRP_Q; k++) {
            const double xn = __dadd_rn(__dmul_rn(cs.b0, e[k]), a0);
            a0 = __dadd_rn(__dsub_rn(__dmul_rn(cs.b1, e[k]), __dmul_rn(cs.a1, xn)), a1);
            a1 = __dsub_rn(__dmul_rn(cs.b2, e[k]), __dmul_rn(cs.a2, xn));
            pout[k] = xn;
        }
        if (gate) {                                            // the lane's data has not arrived yet (jg < s): the state stays (+0, +0)
            const bool on = jg >= s;
            z0 = on ? a0 : z0;
            z1 = on ? a1 : z1;
        } else {
            z0 = a0;
            z1 = a1;
        }
        if (!gate || jg >= AM_NS - 1) {   // (wave-uniform; a constant in the unrolled first block) group jg - 4 of the last section into the ring
            double *dst = yout + (int)(((jg - (AM_NS - 1)) * GRP_Q) & (2 * GRP_T - 1));
#pragma unroll
            for (int k = 0; k < GRP_Q; k += 2) *reinterpret_cast<double2 *>(dst + k) = make_double2(pout[k], pout[k + 1]);
        }
    };
    auto fetch = [&](double (&e)[GRP_Q], int par, int jj) __attribute__((always_inline)) {  // group jj of the block in ebuf[par]
        const double *src = ein + par * (GRP_G * GRP_ES) + jj * GRP_Q;
#pragma unroll
        for (int k = 0; k < GRP_Q; k += 2) {
            const double2 v = *reinterpret_cast<const double2 *>(src + k);
            e[k] = v.x;
            e[k + 1] = v.y;
        }
    };
    fused::lds_barrier();                                      // block 0 is staged
    for (long m = 0; m <= nblk; m++) {
        const int par = (int)(m & 1);
        const long j0 = m * (GRP_T / GRP_Q);
        double ea[GRP_Q], eb[GRP_Q];
        fetch(ea, par, 0);
        if (m == 0) {                                          // the first block: sections 1..4 start one group-step apart
#pragma unroll
            for (int jj = 0; jj < GRP_T / GRP_Q; jj += 2) {
                fetch(eb, par, jj + 1);
                group(ea, pb, pa, j0 + jj, jj < AM_NS - 1);
                if (jj + 2 < GRP_T / GRP_Q) fetch(ea, par, jj + 2);
                group(eb, pa, pb, j0 + jj + 1, jj + 1 < AM_NS - 1);
            }
        } else if (m < nblk) {
#pragma unroll
            for (int jj = 0; jj < GRP_T / GRP_Q; jj += 2) {
                fetch(eb, par, jj + 1);
                group(ea, pb, pa, j0 + jj, false);
                if (jj + 2 < GRP_T / GRP_Q) fetch(ea, par, jj + 2);
                group(eb, pa, pb, j0 + jj + 1, false);
            }
        } else {                                               // the drain: four group-steps on the zero block
            static_assert((AM_NS - 1) % 2 == 0, "the drain is whole pairs of group-steps");
#pragma unroll
            for (int jj = 0; jj < AM_NS - 1; jj += 2) {
                fetch(eb, par, jj + 1);
                group(ea, pb, pa, j0 + jj, nblk == 0);
                if (jj + 2 < AM_NS - 1) fetch(ea, par, jj + 2);
                group(eb, pa, pb, j0 + jj + 1, nblk == 0);
            }
        }
        fused::lds_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------
// SSB: real part of the complex 65-tap FIR (signal_processing.py:204/209; OpenBLAS zdotu accumulation order).
// hilbert(real(z)).real == real(z) up to 1e-16 round-off, so the analytic-signal round trip is not run.
// Sample-parallel: a workgroup takes 1024 outputs of one frame, each thread four consecutive ones (four
// independent fma chains per accumulator); taps are SGPR operands, 32 at a time (one real/imag slot p of the zdot
// kernel per pass); x is staged in LDS as float64.  Outputs with windows shorter than 65 samples (i < 64, or whole
// frames of <= 65 samples) take the predicated per-lane tree.  Per-frame max|y| via atomicMax on the IEEE bit
// pattern (NaN sorts above +inf).
// ---------------------------------------------------------------------------------------------------
constexpr int SSB_CH = 1024, SSB_FB = 4;
constexpr int SSB_XS = (SSB_CH + 64) * 5 / 4 + 8;

__device__ __forceinline__ int xpad4(int p) { return p + (p >> 2); }  // lane stride 5 doubles: conflict-free ds_read_b64

__device__ __forceinline__ void ssb_track_max(unsigned long long m, unsigned long long *wmax, unsigned long long *mxbits_f)
{
    const int tid = threadIdx.x;
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(m, off);
        m = o > m ? o : m;
    }
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); k++) m = wmax[k] > m ? wmax[k] : m;
        atomicMax(mxbits_f, m);
    }
    __syncthreads();
}

template <class IN>   // float2: the reference's complex64 read buffer (widened as lfilter widens it); double2: a complex128 buffer as it is
__global__ __launch_bounds__(TPB) void k_ssb_fir(const IN *__restrict__ iq, double *__restrict__ Yf,
                                                 unsigned long long *__restrict__ mxbits, int n, long n_frames,
                                                 int chunks_per_frame, TapsArg taps)
{
    __shared__ double xr[SSB_XS];  // xr[xpad4(k)] = real(x[i0 - 64 + k])
    __shared__ unsigned long long wmax[TPB / 64];
    const int tid = threadIdx.x;
    const long total = n_frames * chunks_per_frame;
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const long f = w / chunks_per_frame;
        const int i0 = (int)(w % chunks_per_frame) * SSB_CH;
        const IN *x = iq + (size_t)f * n;
        for (int k = tid; k < SSB_CH + 64; k += TPB) {
            int i = i0 - 64 + k;
            xr[xpad4(k)] = (i >= 0 && i < n) ? (double)x[i].x : 0.0;
        }
        __syncthreads();
        unsigned long long m = 0;
        const int ibase = i0 + SSB_FB * tid;
        if (n > 65 && ibase >= 64 && ibase < n) {
            // four consecutive full windows: output o uses x(4 tid + o + j), j = 0..64; xpad4(4 tid + c) = 5 tid + c + c/4.
            // The accumulators (a, p) of the zdot kernel (element j = 8 it + 2a + p) are taken two a's at a time (run-time
            // h: a = 2h, 2h+1 -> 32 taps live in SGPRs); 4h elements further along the row is 5h doubles in the padded array.
            const double *xb = xr + 5 * tid;
            double cp[2][SSB_FB];  // c_p = (acc[0][p] + acc[1][p]) + (acc[2][p] + acc[3][p]), built up over h
#pragma unroll 1
            for (int h = 0; h < 2; h++) {
                const double *xh = xb + 5 * h;
                auto xw = [&](int cc) { return xh[cc + (cc >> 2)]; };
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    double acc[2][SSB_FB];
#pragma unroll
                    for (int a2 = 0; a2 < 2; a2++) {
                        double y[8];
#pragma unroll
                        for (int it = 0; it < 8; it++) y[it] = taps.rev[8 * it + 4 * h + 2 * a2 + p];
#pragma unroll
                        for (int o = 0; o < SSB_FB; o++) {
                            double v = 0.0;
#pragma unroll
                            for (int it = 0; it < 8; it++) v = __fma_rn(xw(o + 8 * it + 2 * a2 + p), y[it], v);
                            acc[a2][o] = v;
                        }
                    }
#pragma unroll
                    for (int o = 0; o < SSB_FB; o++) {
                        const double pair = __dadd_rn(acc[0][o], acc[1][o]);
                        cp[p][o] = (h == 0) ? pair : __dadd_rn(cp[p][o], pair);
                    }
                }
            }
            const double y64 = taps.rev[64];
#pragma unroll
            for (int o = 0; o < SSB_FB; o++)
                if (ibase + o < n) {
                    double y = __fma_rn(xb[(o + 64) + ((o + 64) >> 2)], y64, __dadd_rn(cp[0][o], cp[1][o]));
                    Yf[(size_t)f * n + ibase + o] = y;
                    unsigned long long bb = (unsigned long long)__double_as_longlong(fabs(y));
                    m = bb > m ? bb : m;
                }
        }
        ssb_track_max(m, wmax, &mxbits[f]);
    }
}

// Outputs whose window is shorter than 65 samples — i < 64, or every output of a frame of <= 65 samples (where
// np.convolve keeps the taps as its first operand) — one output per lane with the predicated zdot tree.
template <class IN>
__global__ __launch_bounds__(128) void k_ssb_edge(const IN *__restrict__ iq, double *__restrict__ Yf,
                                                  unsigned long long *__restrict__ mxbits, int n, long n_frames,
                                                  TapsArg taps)
{
    __shared__ double xr[128];
    __shared__ double ltaps[72];
    __shared__ unsigned long long wmax[2];
    const int tid = threadIdx.x;
    const long f = blockIdx.x;
    const IN *x = iq + (size_t)f * n;
    if (tid < 65) ltaps[tid] = taps.fwd[tid];
    xr[tid] = tid < n ? (double)x[tid].x : 0.0;
    __syncthreads();
    unsigned long long m = 0;
    const int ne = n <= 65 ? n : 64;
    if (tid < ne) {
        const int i = tid;
        double y = n <= 65 ? zdot_re_skx_lane([&](int j) { return ltaps[j]; }, [&](int j) { return xr[i - j]; }, i + 1)
                           : zdot_re_skx_lane([&](int j) { return xr[j]; }, [&](int j) { return ltaps[i - j]; }, i + 1);
        Yf[(size_t)f * n + i] = y;
        m = (unsigned long long)__double_as_longlong(fabs(y));
    }
    ssb_track_max(m, wmax, &mxbits[f]);
}

// ---------------------------------------------------------------------------------------------------
// Zero-phase decimator for SMALL batches (a handful of long frames: the reference's interactive loop hands over one
// 32768-sample buffer at a time).  With lane = frame a batch of F frames keeps F/64 wavefronts busy and a single frame
// runs 4 sections x ~9 float64 instructions per sample on one lane of one wavefront.  Here the four cheby1 sections of
// one frame sit in four adjacent lanes (systolic array, one DPP row shift per step, as k_am_sys): 16 frames per
// wavefront, each step a single biquad deep.  One launch filters in one direction:
//   forward  : in = u with SciPy's odd extension around it (k_nfm_front / k_nfm_edge), state zi * in[0], all L outputs kept;
//   backward : in = y_fwd read back to front, state zi * y_fwd[L-1], only the positions 27 + j q survive ([::q] after the
//              27-sample trim), written to A[row][j] together with the row's peak.
// The initial state is non-zero, so section s must not see anything before its first sample: the first four steps update
// the state under a predicate (t >= s), every later step is unconditional; what the array computes after the last sample
// has left a section is never read.
// ---------------------------------------------------------------------------------------------------
constexpr int IS_G = 16, IS_T = 64;
__global__ __launch_bounds__(128) void k_iir4_sys(const double *__restrict__ in, long in_stride, int backward, long L, long T,
                                                  NfmCoef c, double *__restrict__ out, long out_stride, int q, int n_out,
                                                  double *__restrict__ mxout, long n_rows)
{
    // Block-systolic: at macro-step m lane (frame g, section s) of wavefront 0 filters the whole 64-sample block m - s of its frame and
    // leaves it in LDS for lane (g, s + 1), which filters it one macro-step later.  Inside a block a lane's only
    // dependent chain is its own state (xn -> a1*xn -> ... -> z0 -> next xn); nothing crosses lanes sample by sample.
    // Wavefront 1 does the memory traffic (round 3, as in k_am_sys): loads three blocks ahead, stages the next input block, writes the
    // finished one back — the recurrence wavefront never waits for a global access.
    __shared__ double ebuf[2][IS_G][IS_T + 1];          // staged input block of section 0 (two blocks in flight)
    // block handed from section s to s + 1, IN PLACE: within a group of 8 steps every lane first reads its 8 inputs, then
    // writes its 8 outputs, and a wavefront's LDS operations execute in program order — so lane s overwrites positions
    // t0..t0+7 of its row only after lane s + 1 has read them (the previous block), and never touches t0 + 8... early
    __shared__ double xbuf[IS_G][3][IS_T + 1];
    __shared__ double ybuf[2][IS_G][IS_T + 1];          // block leaving section 3
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long f0 = (long)blockIdx.x * IS_G;
    const long nblk = (T + IS_T - 1) / IS_T;
    const long nstep = nblk + 3;
    if (wave == 1) {
        // ---------------- memory wavefront: lane = sample inside a block ----------------
        double mxl[IS_G];
        unsigned nanmask = 0;
#pragma unroll
        for (int gg = 0; gg < IS_G; gg++) mxl[gg] = 0.0;
        double preA[IS_G], preB[IS_G];        // block k is loaded into preA (k even) / preB (k odd)
        // (always_inline: a lambda left as a call keeps what it captures by reference — mxl[], the row pointers — in scratch memory;
        // measured on this kernel: 297 scratch accesses, 11.4 instead of 4.1 ms for 64 frames of 32 768 samples)
        auto load = [&](long blk, double (&pre)[IS_G]) __attribute__((always_inline)) {
#pragma unroll
            for (int gg = 0; gg < IS_G; gg++) {
                const long ff = f0 + gg, r = blk * IS_T + lane;
                pre[gg] = (ff < n_rows && r < T) ? in[(size_t)ff * in_stride + (backward ? L - 1 - r : r)] : 0.0;
            }
        };
        auto stage = [&](long blk, double (&pre)[IS_G]) __attribute__((always_inline)) {
#pragma unroll
            for (int gg = 0; gg < IS_G; gg++) ebuf[blk & 1][gg][lane] = pre[gg];
        };
        auto writeback = [&](long blk, int par) __attribute__((always_inline)) {  // ybuf[par] holds block blk of every frame, lane = sample inside the block
            double yv[IS_G];
#pragma unroll
            for (int gg = 0; gg < IS_G; gg++) yv[gg] = ybuf[par][gg][lane];
#pragma unroll
            for (int gg = 0; gg < IS_G; gg++) {
                const long ff = f0 + gg, r = blk * IS_T + lane;
                if (ff < n_rows && r < T) {
                    const double v = yv[gg];
                    if (!backward) out[(size_t)ff * out_stride + r] = v;
                    else {
                        const int p = (int)(L - 1 - r - EDGE);  // position after the 27-sample trim (0 <= p < M survive it)
                        const unsigned j = (unsigned)p / (unsigned)q;
                        if (p >= 0 && p < (int)(L - 2 * EDGE) && j * (unsigned)q == (unsigned)p) {
                            out[(size_t)ff * out_stride + j] = v;
                            const double av = fabs(v);
                            if (av != av) nanmask |= 1u << gg;
                            mxl[gg] = av > mxl[gg] ? av : mxl[gg];
                        }
                    }
                }
            }
        };
        load(0, preA);
        stage(0, preA);
        if (nblk > 1) load(1, preB);
        if (nblk > 2) load(2, preA);
        fused::lds_barrier();   // LDS-only: the memory wavefront's loads and stores stay in flight across it
        auto beside = [&](long m, double (&pre)[IS_G]) __attribute__((always_inline)) {     // pre: the set holding block m + 1
            if (m + 1 < nblk) stage(m + 1, pre);
            if (m + 3 < nblk) load(m + 3, pre);
            if (m >= 4) writeback(m - 4, (int)((m - 1) & 1));    // section 3 finished block m - 4 in the previous macro-step
            fused::lds_barrier();
        };
        for (long m = 0; m < nstep; m += 2) {
            beside(m, preB);
            if (m + 1 < nstep) beside(m + 1, preA);
        }
        writeback(nblk - 1, (int)((nstep - 1) & 1));
        if (backward) {
#pragma unroll
            for (int gg = 0; gg < IS_G; gg++) {
                double mm = mxl[gg];
                int nn = (nanmask >> gg) & 1;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double o = __shfl_xor(mm, off);
                    mm = o > mm ? o : mm;
                    nn |= __shfl_xor(nn, off);
                }
                if (lane == 0 && f0 + gg < n_rows) mxout[f0 + gg] = nn ? __builtin_nan("") : mm;
            }
        }
        return;
    }
    // ---------------- recurrence wavefront: 16 frames x 4 sections ----------------
    const int g = lane >> 2, s = lane & 3;
    const long fg = (f0 + g < n_rows) ? f0 + g : n_rows - 1;
    const Biquad cs = c.s[s];
    const double x0 = in[(size_t)fg * in_stride + (backward ? L - 1 : 0)];
    double z0 = __dmul_rn(c.zi[2 * s], x0), z1 = __dmul_rn(c.zi[2 * s + 1], x0);
    auto step = [&](double x) {
        const double xn = __dadd_rn(__dmul_rn(cs.b0, x), z0);
        z0 = __dadd_rn(__dsub_rn(__dmul_rn(cs.b1, x), __dmul_rn(cs.a1, xn)), z1);
        z1 = __dsub_rn(__dmul_rn(cs.b2, x), __dmul_rn(cs.a2, xn));
        return xn;
    };
    fused::lds_barrier();
    for (long m = 0; m < nstep; m++) {
        const long blk = m - s;
        const bool active = blk >= 0 && blk < nblk;
        const double *src = s == 0 ? ebuf[m & 1][g] : xbuf[g][s > 0 ? s - 1 : 0];
        double *dst = s == 3 ? ybuf[m & 1][g] : xbuf[g][s];
        const int cnt = !active ? 0 : ((T - blk * IS_T) < IS_T ? (int)(T - blk * IS_T) : IS_T);
        // all lanes must walk the block in lockstep for the in-place hand-off: one code path per macro-step, chosen wave-wide
        if (__all(!active || cnt == IS_T)) {
            if (active) {
#ifdef PSS_EXP_SYS_NOPF
                for (int t0 = 0; t0 < IS_T; t0 += 8) {
                    double e8[8], y8[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) e8[k] = src[t0 + k];
#pragma unroll
                    for (int k = 0; k < 8; k++) y8[k] = step(e8[k]);
#pragma unroll
                    for (int k = 0; k < 8; k++) dst[t0 + k] = y8[k];
                }
#else
                // two groups of eight per turn, each group's inputs requested from LDS before the OTHER group's recurrence steps (a lone
                // wavefront: the LDS round trip of a group otherwise stands in front of its 8 x 48 clocks); reading ahead is safe for the
                // in-place hand-off — a position is read before this macro-step's write of it either way.  No register copies: the two
                // groups alternate between two register sets (a rotating single set cost 16 moves per group and lost 14 %).
                static_assert(IS_T % 16 == 0, "block length");
                double ea[8], eb[8], y8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) ea[k] = src[k];
#pragma unroll 1
                for (int t0 = 0; t0 < IS_T; t0 += 16) {
#pragma unroll
                    for (int k = 0; k < 8; k++) eb[k] = src[t0 + 8 + k];
#pragma unroll
                    for (int k = 0; k < 8; k++) y8[k] = step(ea[k]);
#pragma unroll
                    for (int k = 0; k < 8; k++) dst[t0 + k] = y8[k];
                    if (t0 + 16 < IS_T) {
#pragma unroll
                        for (int k = 0; k < 8; k++) ea[k] = src[t0 + 16 + k];
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) y8[k] = step(eb[k]);
#pragma unroll
                    for (int k = 0; k < 8; k++) dst[t0 + 8 + k] = y8[k];
                }
#endif
            }
        } else {
            for (int t = 0; t < IS_T; t++) {  // a partial block somewhere: masked steps, read-then-write per position
                if (t < cnt) {
                    const double e = src[t];
                    dst[t] = step(e);
                }
            }
        }
        fused::lds_barrier();
    }
}

// audio = y / max|y| * 0.95  -> float64 mono and/or int16 stereo (AM :194, SSB :216, io_manager.py:26)
__global__ __launch_bounds__(TPB) void k_finalize(const double *__restrict__ Yf, const double *__restrict__ mx, int n,
                                                  long n_frames, int16_t *__restrict__ pcm, double *__restrict__ audio)
{
    const size_t total = (size_t)n_frames * n;
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        long f = (long)(idx / n);
        double a = __dmul_rn(__ddiv_rn(Yf[idx], mx[f]), 0.95);
        if (audio) audio[idx] = a;
        if (pcm) {
            uint16_t s = (uint16_t)pcm16(a);
            reinterpret_cast<uint32_t *>(pcm)[idx] = (uint32_t)s | ((uint32_t)s << 16);
        }
    }
}

// ---- iq_correction (signal_processing.py:46-80) ------------------------------------------------------------------
// One workgroup per frame.  The routine is a chain of eight float32 reductions whose results feed the next
// elementwise step; each reduction walks numpy's own summation tree (PssPairwisePlan) so every intermediate scalar
// has numpy's bits.  The frame (8 KB at n=1024) is re-read from L1/L2 per pass; intermediates are recomputed.
struct PlanDev {
    const int *leaf_off, *leaf_len, *node_l, *node_r, *level_start, *roots;
    int n_leaves, n_levels, n_roots, n_nodes;
    int wave_tree;  // see PssPairwisePlan::wave_tree
};
// The plan tables are walked with DEPENDENT loads several times per reduction (offsets -> elements, one round per tree
// level): read from global memory that was ~8 us per pass and dominated k_iqcorr (8 passes per frame).  Every workgroup
// copies the tables it needs into LDS once and works from there.
__device__ __forceinline__ int plan_ints(const PlanDev &p)
{
    return p.n_leaves ? 2 * p.n_leaves + 2 * p.n_nodes + (p.n_levels + 1) + p.n_roots : 0;
}
__device__ __forceinline__ void plan_to_lds(PlanDev &p, int *&cur)
{
    if (!p.n_leaves) return;
    const int tid = threadIdx.x, T = blockDim.x;
    int *lo = cur, *ll = lo + p.n_leaves, *nl = ll + p.n_leaves, *nr = nl + p.n_nodes, *ls = nr + p.n_nodes, *rt = ls + p.n_levels + 1;
    for (int i = tid; i < p.n_leaves; i += T) { lo[i] = p.leaf_off[i]; ll[i] = p.leaf_len[i]; }
    for (int i = tid; i < p.n_nodes; i += T) { nl[i] = p.node_l[i]; nr[i] = p.node_r[i]; }
    for (int i = tid; i <= p.n_levels; i += T) ls[i] = p.level_start[i];
    for (int i = tid; i < p.n_roots; i += T) rt[i] = p.roots[i];
    p.leaf_off = lo; p.leaf_len = ll; p.node_l = nl; p.node_r = nr; p.level_start = ls; p.roots = rt;
    cur = rt + p.n_roots;
}
// A frame is reduced in GROUPS of up to RED_K ufunc chunks (8192 elements each): numpy adds the chunk sums sequentially,
// sum = ((S0 + S1) + S2) + ..., so a group's plan is a forest (one pairwise tree per chunk) whose roots are added in
// order onto the running sum.  Frames up to RED_K chunks are one (tail) group; longer ones loop over full groups first —
// the LDS footprint no longer grows with the frame (1 Mi-sample read buffers, pyspecsdr.py:2236 with SAMPLES = 12).
struct RedPlan {
    PlanDev full, tail;
    int n_full, glen;  // full groups of glen elements each, then the tail group (tail.n_leaves may be 0)
};
// A leaf of numpy's tree is <= 128 elements summed into 8 running accumulators; those 8 partial sums are independent,
// so a lane owns one (leaf, accumulator) pair — at n = 1024 that is exactly one wavefront per frame — and one lane per
// leaf then folds the 8 partials and the < 8 tail elements in numpy's order.  part: 8 floats per leaf.
template <bool WT = false, class F>
__device__ __forceinline__ float wg_rsum(const PlanDev &p, float *part, float *val, F elem, float carry, bool have)
{
    const int tid = threadIdx.x, T = blockDim.x;
    if constexpr (WT) {  // the host guarantees: p.wave_tree, a 64-thread workgroup, a single group (no carry)
        // n = 1024: 8 leaves x 8 accumulators = the 64 lanes of the one wavefront of this workgroup, and numpy's tree is
        // perfectly balanced — ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) inside a leaf, adjacent halves above it.  IEEE addition is
        // commutative, so an xor-butterfly computes exactly those sums (in every lane): no LDS, no barriers.
        const int lane = tid & 63;  // (in a wider workgroup every wavefront computes the same sums redundantly)
        const int l = lane >> 3, k = lane & 7, off = l * 128;
        float r = elem(off + k);
#pragma unroll
        for (int i = 8; i < 128; i += 8) r = __fadd_rn(r, elem(off + i + k));
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) r = __fadd_rn(r, __shfl_xor(r, m));
        return r;
    }
    for (int slot = tid; slot < p.n_leaves * 8; slot += T) {
        const int l = slot >> 3, k = slot & 7, off = p.leaf_off[l], len = p.leaf_len[l];
#ifndef PSS_EXP_RSUM4
        if (len == 128) {
            // a full leaf: all 16 operands of this accumulator requested before the dependent chain of additions starts (four ahead,
            // a CU's 2048 threads kept ~64 KB in flight and the pass ran at 3.7 TB/s)
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = elem(off + 8 * j + k);
            float r = v[0];
#pragma unroll
            for (int j = 1; j < 16; j++) r = __fadd_rn(r, v[j]);
            part[slot] = r;
        } else
#endif
        if (len >= 8) {
            // the additions are a dependent chain in numpy's order; the operands are not: fetch four ahead of the chain
            float r = elem(off + k);
            const int end = len - (len % 8);
            int i = 8;
            for (; i + 24 < end; i += 32) {
                const float a = elem(off + i + k), b = elem(off + i + 8 + k), c = elem(off + i + 16 + k), d = elem(off + i + 24 + k);
                r = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(r, a), b), c), d);
            }
            for (; i < end; i += 8) r = __fadd_rn(r, elem(off + i + k));
            part[slot] = r;
        }
    }
    __syncthreads();
    for (int l = tid; l < p.n_leaves; l += T) {
        const int off = p.leaf_off[l], len = p.leaf_len[l];
        float res;
        if (len < 8) {
            res = 0.0f;
            for (int i = 0; i < len; i++) res = __fadd_rn(res, elem(off + i));
        } else {
            const float *r = part + 8 * l;
            res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                            __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
            for (int i = len - (len % 8); i < len; i++) res = __fadd_rn(res, elem(off + i));
        }
        val[l] = res;
    }
    __syncthreads();
    for (int lv = 0; lv < p.n_levels; lv++) {
        for (int k = p.level_start[lv] + tid; k < p.level_start[lv + 1]; k += T)
            val[p.n_leaves + k] = __fadd_rn(val[p.node_l[k]], val[p.node_r[k]]);
        __syncthreads();
    }
    const int res = p.n_leaves + p.level_start[p.n_levels];  // free slot behind the nodes
    if (tid == 0) {
        float acc = have ? __fadd_rn(carry, val[p.roots[0]]) : val[p.roots[0]];
        for (int k = 1; k < p.n_roots; k++) acc = __fadd_rn(acc, val[p.roots[k]]);
        val[res] = acc;
    }
    __syncthreads();
    const float sum = val[res];
    __syncthreads();
    return sum;
}
// complex64 reduce: elem(ci) -> float2 of complex element ci; leaves are float ranges of the interleaved array, the 8
// float accumulators are 4 complex ones: a lane owns one (leaf, complex accumulator) pair.  part: 4 float2 per leaf.
template <bool WT = false, class F>
__device__ __forceinline__ float2 wg_csum(const PlanDev &p, float2 *part, float2 *val, F elem, float2 carry, bool have)
{
    const int tid = threadIdx.x, T = blockDim.x;
    if constexpr (WT) {
        // 1024 complex = 2048 floats: 16 leaves x 4 complex accumulators = 64 lanes; same butterfly as wg_rsum
        const int lane = tid & 63;
        const int l = lane >> 2, k = lane & 3, off = l * 64;  // complex offset of the leaf
        float2 r = elem(off + k);
#pragma unroll
        for (int i = 4; i < 64; i += 4) {
            const float2 v = elem(off + i + k);
            r.x = __fadd_rn(r.x, v.x);
            r.y = __fadd_rn(r.y, v.y);
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            r.x = __fadd_rn(r.x, __shfl_xor(r.x, m));
            r.y = __fadd_rn(r.y, __shfl_xor(r.y, m));
        }
        return r;
    }
    for (int slot = tid; slot < p.n_leaves * 4; slot += T) {
        const int l = slot >> 2, k = slot & 3, off = p.leaf_off[l] >> 1, len = p.leaf_len[l];  // off: complex, len: floats
        if (len >= 8) {
            float2 r = elem(off + k);
            const int end = len - (len % 8);
            int i = 8;
            for (; i + 24 < end; i += 32) {  // operands fetched four ahead of the dependent additions
                const float2 a = elem(off + (i >> 1) + k), b = elem(off + ((i + 8) >> 1) + k), c = elem(off + ((i + 16) >> 1) + k),
                             d = elem(off + ((i + 24) >> 1) + k);
                r.x = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(r.x, a.x), b.x), c.x), d.x);
                r.y = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(r.y, a.y), b.y), c.y), d.y);
            }
            for (; i < end; i += 8) {
                const float2 v = elem(off + (i >> 1) + k);
                r.x = __fadd_rn(r.x, v.x);
                r.y = __fadd_rn(r.y, v.y);
            }
            part[slot] = r;
        }
    }
    __syncthreads();
    for (int l = tid; l < p.n_leaves; l += T) {
        const int off = p.leaf_off[l] >> 1, len = p.leaf_len[l];
        float rr, ri;
        if (len < 8) {
            rr = 0.0f; ri = 0.0f;
            for (int i = 0; i < len; i += 2) { const float2 v = elem(off + (i >> 1)); rr = __fadd_rn(rr, v.x); ri = __fadd_rn(ri, v.y); }
        } else {
            const float2 *r = part + 4 * l;
            rr = __fadd_rn(__fadd_rn(r[0].x, r[1].x), __fadd_rn(r[2].x, r[3].x));
            ri = __fadd_rn(__fadd_rn(r[0].y, r[1].y), __fadd_rn(r[2].y, r[3].y));
            for (int i = len - (len % 8); i < len; i += 2) { const float2 v = elem(off + (i >> 1)); rr = __fadd_rn(rr, v.x); ri = __fadd_rn(ri, v.y); }
        }
        val[l] = make_float2(rr, ri);
    }
    __syncthreads();
    for (int lv = 0; lv < p.n_levels; lv++) {
        for (int k = p.level_start[lv] + tid; k < p.level_start[lv + 1]; k += T) {
            const float2 a = val[p.node_l[k]], b = val[p.node_r[k]];
            val[p.n_leaves + k] = make_float2(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y));
        }
        __syncthreads();
    }
    const int res = p.n_leaves + p.level_start[p.n_levels];
    if (tid == 0) {
        float2 acc = val[p.roots[0]];
        if (have) acc = make_float2(__fadd_rn(carry.x, acc.x), __fadd_rn(carry.y, acc.y));
        for (int k = 1; k < p.n_roots; k++) {
            const float2 v = val[p.roots[k]];
            acc = make_float2(__fadd_rn(acc.x, v.x), __fadd_rn(acc.y, v.y));
        }
        val[res] = acc;
    }
    __syncthreads();
    const float2 sum = val[res];
    __syncthreads();
    return sum;
}
// whole-frame reductions: loop over the groups, elem(i) indexed from the start of the frame
template <bool WT = false, class F>
__device__ __forceinline__ float frame_rsum(const RedPlan &rp, float *part, float *val, F elem)
{
    if constexpr (WT) return wg_rsum<true>(rp.tail, part, val, elem, 0.0f, false);
    float acc = 0.0f;
    bool have = false;
    for (int g = 0; g < rp.n_full; g++) {
        const int base = g * rp.glen;
        acc = wg_rsum(rp.full, part, val, [&](int i) { return elem(base + i); }, acc, have);
        have = true;
    }
    if (rp.tail.n_leaves) {
        const int base = rp.n_full * rp.glen;
        acc = wg_rsum(rp.tail, part, val, [&](int i) { return elem(base + i); }, acc, have);
    }
    return acc;
}
template <bool WT = false, class F>
__device__ __forceinline__ float2 frame_csum(const RedPlan &rp, float2 *part, float2 *val, F elem)
{
    if constexpr (WT) return wg_csum<true>(rp.tail, part, val, elem, make_float2(0.0f, 0.0f), false);
    float2 acc = make_float2(0.0f, 0.0f);
    bool have = false;
    for (int g = 0; g < rp.n_full; g++) {
        const int base = g * rp.glen;
        acc = wg_csum(rp.full, part, val, [&](int i) { return elem(base + i); }, acc, have);
        have = true;
    }
    if (rp.tail.n_leaves) {
        const int base = rp.n_full * rp.glen;
        acc = wg_csum(rp.tail, part, val, [&](int i) { return elem(base + i); }, acc, have);
    }
    return acc;
}

// np.mean float32 of |x| (KIND 1, AM: signal_processing.py:185) or |x|^2 (KIND 0, power: :327), one workgroup per frame;
// the (leaf, accumulator) lanes read global memory directly (every element is used exactly once; staging the frame in
// LDS first measured 2-3x slower: fewer resident workgroups, one more barrier).  LDS: [part: 8 floats per leaf][val]
template <int KIND, bool WT = false>
__global__ __launch_bounds__(256) void k_pairwise(const float2 *__restrict__ iq, int n, long n_frames, RedPlan rp,
                                                  int part_slots, int val_slots, float *__restrict__ out, float *__restrict__ env)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *part = reinterpret_cast<float *>(smem), *val = part + part_slots;
    {
        int *cur = reinterpret_cast<int *>(val + val_slots);
        plan_to_lds(rp.full, cur);
        plan_to_lds(rp.tail, cur);
        __syncthreads();
    }
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *x = iq + (size_t)f * n;
        const float sum = frame_rsum<WT>(rp, part, val, [&](int i) {
            const float2 v = x[i];
            const float m = cabsf_np(v.x, v.y);
            if (KIND == 1 && env) env[(size_t)f * n + i] = m;  // the AM envelope (:182), reused by the band-pass kernel
            return KIND == 0 ? __fmul_rn(m, m) : m;
        });
        if (threadIdx.x == 0) {
            const float mean = __fdiv_rn(sum, (float)n);
            out[f] = KIND == 0 ? __fmul_rn(10.0f, log10f_np(__fadd_rn(mean, 1e-10f))) : mean;  // 10*log10(power + 1e-10), float32 (SVML model)
        }
    }
}

// Two real float32 reductions over the SAME elements with the SAME (real) plan in one walk: elem(i) -> (a_i, b_i); both sums have numpy's
// bits (the trees are identical, the additions component-wise).  part / val: float2 arrays of the slot counts of the float versions.
template <bool WT = false, class F>
__device__ __forceinline__ float2 wg_rsum2(const PlanDev &p, float2 *part, float2 *val, F elem, float2 carry, bool have)
{
    const int tid = threadIdx.x, T = blockDim.x;
    auto add = [](float2 a, float2 b) { return make_float2(__fadd_rn(a.x, b.x), __fadd_rn(a.y, b.y)); };
    if constexpr (WT) {
        const int lane = tid & 63;
        const int l = lane >> 3, k = lane & 7, off = l * 128;
        float2 r = elem(off + k);
#pragma unroll
        for (int i = 8; i < 128; i += 8) r = add(r, elem(off + i + k));
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) r = add(r, make_float2(__shfl_xor(r.x, m), __shfl_xor(r.y, m)));
        return r;
    }
    for (int slot = tid; slot < p.n_leaves * 8; slot += T) {
        const int l = slot >> 3, k = slot & 7, off = p.leaf_off[l], len = p.leaf_len[l];
        if (len == 128) {
            float2 v[16];
#pragma unroll
            for (int j = 0; j < 16; j++) v[j] = elem(off + 8 * j + k);
            float2 r = v[0];
#pragma unroll
            for (int j = 1; j < 16; j++) r = add(r, v[j]);
            part[slot] = r;
        } else if (len >= 8) {
            float2 r = elem(off + k);
            const int end = len - (len % 8);
            for (int i = 8; i < end; i += 8) r = add(r, elem(off + i + k));
            part[slot] = r;
        }
    }
    __syncthreads();
    for (int l = tid; l < p.n_leaves; l += T) {
        const int off = p.leaf_off[l], len = p.leaf_len[l];
        float2 res;
        if (len < 8) {
            res = make_float2(0.0f, 0.0f);
            for (int i = 0; i < len; i++) res = add(res, elem(off + i));
        } else {
            const float2 *r = part + 8 * l;
            res = add(add(add(r[0], r[1]), add(r[2], r[3])), add(add(r[4], r[5]), add(r[6], r[7])));
            for (int i = len - (len % 8); i < len; i++) res = add(res, elem(off + i));
        }
        val[l] = res;
    }
    __syncthreads();
    for (int lv = 0; lv < p.n_levels; lv++) {
        for (int k = p.level_start[lv] + tid; k < p.level_start[lv + 1]; k += T) val[p.n_leaves + k] = add(val[p.node_l[k]], val[p.node_r[k]]);
        __syncthreads();
    }
    const int res = p.n_leaves + p.level_start[p.n_levels];
    if (tid == 0) {
        float2 acc = have ? add(carry, val[p.roots[0]]) : val[p.roots[0]];
        for (int k = 1; k < p.n_roots; k++) acc = add(acc, val[p.roots[k]]);
        val[res] = acc;
    }
    __syncthreads();
    const float2 sum = val[res];
    __syncthreads();
    return sum;
}
template <bool WT = false, class F>
__device__ __forceinline__ float2 frame_rsum2(const RedPlan &rp, float2 *part, float2 *val, F elem)
{
    if constexpr (WT) return wg_rsum2<true>(rp.tail, part, val, elem, make_float2(0.0f, 0.0f), false);
    float2 acc = make_float2(0.0f, 0.0f);
    bool have = false;
    for (int g = 0; g < rp.n_full; g++) {
        const int base = g * rp.glen;
        acc = wg_rsum2(rp.full, part, val, [&](int i) { return elem(base + i); }, acc, have);
        have = true;
    }
    if (rp.tail.n_leaves) {
        const int base = rp.n_full * rp.glen;
        acc = wg_rsum2(rp.tail, part, val, [&](int i) { return elem(base + i); }, acc, have);
    }
    return acc;
}

// measure_signal_power (signal_processing.py:325-328) AND the AM demodulator's mean of |x| + envelope (:182-185) from ONE pass over the
// frame: both are float32 np.mean reductions of functions of the same np.abs value (|x|^2 and |x|), so both trees are walked together —
// the main loop measures the power of every read buffer before it demodulates it (pyspecsdr.py:2251, :2262): one IQ read instead of two.
template <bool WT = false>
__global__ __launch_bounds__(256) void k_pairwise2(const float2 *__restrict__ iq, int n, long n_frames, RedPlan rp, int part_slots, int val_slots,
                                                   float *__restrict__ out_power, float *__restrict__ out_mean, float *__restrict__ env)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *part = reinterpret_cast<float2 *>(smem), *val = part + part_slots;
    {
        int *cur = reinterpret_cast<int *>(val + val_slots);
        plan_to_lds(rp.full, cur);
        plan_to_lds(rp.tail, cur);
        __syncthreads();
    }
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *x = iq + (size_t)f * n;
        const float2 sum = frame_rsum2<WT>(rp, part, val, [&](int i) {
            const float2 v = x[i];
            const float m = cabsf_np(v.x, v.y);
            if (env) env[(size_t)f * n + i] = m;
            return make_float2(__fmul_rn(m, m), m);
        });
        if (threadIdx.x == 0) {
            out_power[f] = __fmul_rn(10.0f, log10f_np(__fadd_rn(__fdiv_rn(sum.x, (float)n), 1e-10f)));
            out_mean[f] = __fdiv_rn(sum.y, (float)n);
        }
    }
}

// iq_correction's elementwise tail (signal_processing.py:55-80) for one sample, given the frame's five scalars: scl = 1 / q_amplitude (:55),
// ia = 1 / alpha and qa2 = -sin(phi) / alpha (:67-68), sc = 1 / cos(phi) (:71), g = sqrt(input_power / var(corrected)) (:80).  Shared by
// k_iqcorr (which also derives the scalars) and the fused WFM forward kernel, which applies the correction as it reads the raw IQ.
struct IqcScal { float scl, ia, qa2, sc, g; };
__device__ __forceinline__ float2 iqc_corrected(float2 v, float scl, float ia, float qa2, float sc)
{
    // :67-71 (the 1j*q_new complex multiply only touches the sign of zeros)
    const float is = __fmul_rn(v.x, scl), qs = __fmul_rn(v.y, scl);
    const float i_new = __fmul_rn(ia, is), q_new = __fadd_rn(__fmul_rn(qa2, is), qs);
    const float jr = __fmaf_rn(0.0f, q_new, -0.0f), ji = __fmaf_rn(0.0f, 0.0f, q_new);
    return make_float2(__fmul_rn(__fadd_rn(i_new, jr), sc), __fmul_rn(__fadd_rn(0.0f, ji), sc));
}
__device__ __forceinline__ float2 iqc_apply(float2 v, const IqcScal &k)
{
    float2 c = iqc_corrected(v, k.scl, k.ia, k.qa2, k.sc);
    c.x = __fmul_rn(c.x, k.g);
    c.y = __fmul_rn(c.y, k.g);
    return c;
}

// STAGED: the frame is copied to LDS once and every pass reads it from there (frames up to 8192 samples).
// scal (nullable): the frame's five scalars [n_frames][8] (IqcScal + padding) — with out == raw == nullptr the corrected frame is not
// written at all (pss_frame_pipeline(WFM): the forward kernel corrects the samples as it loads them).
// LDS: [frame: n float2 if STAGED][part: part_slots float2][val: val_slots float2]
template <bool STAGED, bool WT = false>
__global__ __launch_bounds__(256) void k_iqcorr(const float2 *__restrict__ iq, int n, long n_frames, RedPlan rp, RedPlan cp,
                                                int part_slots, int val_slots, float2 *__restrict__ out, float *__restrict__ raw,
                                                float *__restrict__ scal)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    float2 *cpart = xs + (STAGED ? n : 0), *cval = cpart + part_slots;
    float *rpart = reinterpret_cast<float *>(cpart), *rval = reinterpret_cast<float *>(cval);
    {
        int *cur = reinterpret_cast<int *>(cval + val_slots);
        plan_to_lds(rp.full, cur);
        plan_to_lds(rp.tail, cur);
        plan_to_lds(cp.full, cur);
        plan_to_lds(cp.tail, cur);
        __syncthreads();
    }
    const float fn = (float)n;
    const int T = blockDim.x;
    // WT (one wavefront per frame, 1024 samples): the NEXT frame's 16 samples per lane are requested before this frame's reductions and
    // staged behind them — the wavefront no longer sits through a global round trip per frame
    float2 nxt[WT ? 16 : 1];
    if constexpr (WT) {
        if ((long)blockIdx.x < n_frames) {
#pragma unroll
            for (int j = 0; j < 16; j++) nxt[j] = iq[(size_t)blockIdx.x * n + threadIdx.x + 64 * j];
        }
    }
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *xg = iq + (size_t)f * n;
        if constexpr (WT) {
#pragma unroll
            for (int j = 0; j < 16; j++) xs[threadIdx.x + 64 * j] = nxt[j];
            if (f + gridDim.x < n_frames) {
#pragma unroll
                for (int j = 0; j < 16; j++) nxt[j] = iq[(size_t)(f + gridDim.x) * n + threadIdx.x + 64 * j];
            }
            __syncthreads();
        } else if (STAGED) {
            for (int i = threadIdx.x; i < n; i += T) xs[i] = xg[i];
            __syncthreads();
        }
        auto X = [&](int i) { return STAGED ? xs[i] : xg[i]; };
        // The eight reductions in DEPENDENCY order rather than source order: (:48 mean, :52 q power) need only the samples, (:49 second mean,
        // :60 alpha, :61 sin phi) need those two, (:49 variance, :80 mean of the corrected frame) the next three.  Each reduction is the same
        // tree as before (same bits); with one wavefront per frame (WT) they are straight-line code and the independent ones of a group
        // overlap their 16-deep dependent addition chains instead of running them back to back.
        // :48 centered = samples - mean(samples)
        float2 s = frame_csum<WT>(cp, cpart, cval, X);
        // :52 q_amplitude
        const float qsum = frame_rsum<WT>(rp, rpart, rval, [&](int i) { float q = X(i).y; return __fmul_rn(q, q); });
        const float mr = __fdiv_rn(s.x, fn), mi = __fdiv_rn(s.y, fn);
        const float qa = sqrtf(__fmul_rn(2.0f, __fdiv_rn(qsum, fn)));
        const float scl = __fdiv_rn(1.0f, qa);  // :55 complex64 / float32 scalar multiplies by the reciprocal
        // :49 input_power = var(centered): mean again, re^2 + im^2 as three separately rounded float32 operations, mean
        s = frame_csum<WT>(cp, cpart, cval, [&](int i) { float2 v = X(i); return make_float2(__fsub_rn(v.x, mr), __fsub_rn(v.y, mi)); });
        // :60-61 alpha, sin(phi)
        float asum, psum;
        if constexpr (WT) {     // both trees in one walk over the same elements (wg_rsum2: component-wise, numpy's bits for both)
            const float2 ap = frame_rsum2<true>(rp, cpart, cval, [&](int i) {
                float2 v = X(i);
                const float is = __fmul_rn(v.x, scl);
                return make_float2(__fmul_rn(is, is), __fmul_rn(is, __fmul_rn(v.y, scl)));
            });
            asum = ap.x;
            psum = ap.y;
        } else {
            asum = frame_rsum<WT>(rp, rpart, rval, [&](int i) {
                const float is = __fmul_rn(X(i).x, scl);
                return __fmul_rn(is, is);
            });
            psum = frame_rsum<WT>(rp, rpart, rval, [&](int i) {
                float2 v = X(i);
                return __fmul_rn(__fmul_rn(v.x, scl), __fmul_rn(v.y, scl));
            });
        }
        const float m2r = __fdiv_rn(s.x, fn), m2i = __fdiv_rn(s.y, fn);
        const float alpha = sqrtf(__fmul_rn(2.0f, __fdiv_rn(asum, fn)));
        const float sinphi = __fmul_rn(__fdiv_rn(2.0f, alpha), __fdiv_rn(psum, fn));
        const float cosphi = sqrtf(__fsub_rn(1.0f, __fmul_rn(sinphi, sinphi)));  // :64
        const float ia = __fdiv_rn(1.0f, alpha), qa2 = __fdiv_rn(-sinphi, alpha), sc = __fdiv_rn(1.0f, cosphi);
        auto corrected = [&](int i) { return iqc_corrected(X(i), scl, ia, qa2, sc); };
        const float ipsum = frame_rsum<WT>(rp, rpart, rval, [&](int i) {
            float2 v = X(i);
            const float dr = __fsub_rn(__fsub_rn(v.x, mr), m2r), di = __fsub_rn(__fsub_rn(v.y, mi), m2i);
            return __fadd_rn(__fmul_rn(dr, dr), __fmul_rn(di, di));  // np.var fast path: squares, then add (no fma)
        });
        // :80 var(corrected), rescale to the input power
        s = frame_csum<WT>(cp, cpart, cval, corrected);
        const float input_power = __fdiv_rn(ipsum, fn);
        const float m3r = __fdiv_rn(s.x, fn), m3i = __fdiv_rn(s.y, fn);
        const float v2 = __fdiv_rn(frame_rsum<WT>(rp, rpart, rval, [&](int i) {
            float2 c = corrected(i);
            const float dr = __fsub_rn(c.x, m3r), di = __fsub_rn(c.y, m3i);
            return __fadd_rn(__fmul_rn(dr, dr), __fmul_rn(di, di));  // np.var fast path: squares, then add (no fma)
        }), fn);
        const float g = sqrtf(__fdiv_rn(input_power, v2));
        if (scal && threadIdx.x == 0) {
            float *k = scal + (size_t)f * 8;
            k[0] = scl; k[1] = ia; k[2] = qa2; k[3] = sc; k[4] = g;
        }
        if (out || raw) {
            for (int i = threadIdx.x; i < n; i += T) {
                float2 c = corrected(i);
                c.x = __fmul_rn(c.x, g);
                c.y = __fmul_rn(c.y, g);
                if (out) out[(size_t)f * n + i] = c;
                if (raw) raw[(size_t)f * n + i] = c.x;  // demodulate_signal(..., 'RAW'): np.real(samples) (:238)
            }
        }
        __syncthreads();  // xs is overwritten by the next frame
    }
}

// ---- demodulate_wfm (signal_processing.py:119-176) ----------------------------------------------------------------
// Everything up to the decimator is causal and serial in time, so one lane owns one frame and walks it once:
// discriminator -> {LP15k, BP pilot -> 1-pole -> sign, BP 23..53k} -> x(2*pilot) -> LP15k -> L/R matrix -> de-emphasis.
// The reference's pilot = sin(unwrap(angle(real signal))) is 0.0 or sin(pi) (see oracle/pss_oracle.c), restated as such.
// The two de-emphasised channels land in U as rows 2f (left) and 2f+1 (right), odd-extended by 27 samples each side,
// which is the layout k_nfm_iir (zero-phase cheby1 decimator) consumes.
__device__ __forceinline__ double lfilter1(double b0, double a1, double x, double &z)
{
    const double y = __dadd_rn(z, __dmul_rn(b0, x));
    z = __dsub_rn(__dmul_rn(x, 0.0), __dmul_rn(y, a1));
    return y;
}
// Numerator shapes of SciPy's Butterworth SOS rows (zpk2sos 'nearest' pairing puts the zeros at +-1 / 0 the same way
// at every sample rate): multiplications by 1, 2, -1, -2 are exact and are dropped; a zero coefficient keeps its
// multiply (0*x carries the sign of x into a -0.0).  NUM_GEN is sosfilt's step verbatim.
enum { NUM_GEN = 0, NUM_121 = 1, NUM_10M1 = 2, NUM_1M21 = 3, NUM_110 = 4 };
template <int K>
__device__ __forceinline__ double biquad_num(const Biquad &c, double x, double &z0, double &z1)
{
    if (K == NUM_GEN) return biquad_step(c, x, z0, z1);
    const double xn = __dadd_rn(x, z0);                                     // b0 = 1
    const double t = __dmul_rn(c.a1, xn), u = __dmul_rn(c.a2, xn);
    double v, w;
    if (K == NUM_121) { v = __dadd_rn(x, x); w = x; }
    else if (K == NUM_1M21) { v = -__dadd_rn(x, x); w = x; }
    else if (K == NUM_10M1) { v = __dmul_rn(c.b1, x); w = -x; }
    else { v = x; w = __dmul_rn(c.b2, x); }                                 // NUM_110
    z0 = __dadd_rn(__dsub_rn(v, t), z1);
    z1 = __dsub_rn(w, u);
    return xn;
}
template <bool SPEC>
__device__ __forceinline__ double wfm_lp(const Biquad *c, double x, double *z)
{
    x = biquad_num<NUM_GEN>(c[0], x, z[0], z[1]);
    x = biquad_num<SPEC ? NUM_121 : NUM_GEN>(c[1], x, z[2], z[3]);
    return biquad_num<SPEC ? NUM_110 : NUM_GEN>(c[2], x, z[4], z[5]);
}
template <bool SPEC>
__device__ __forceinline__ double wfm_bp(const Biquad *c, double x, double *z)
{
    x = biquad_num<NUM_GEN>(c[0], x, z[0], z[1]);
    x = biquad_num<SPEC ? NUM_121 : NUM_GEN>(c[1], x, z[2], z[3]);
    x = biquad_num<SPEC ? NUM_10M1 : NUM_GEN>(c[2], x, z[4], z[5]);
    x = biquad_num<SPEC ? NUM_1M21 : NUM_GEN>(c[3], x, z[6], z[7]);
    return biquad_num<SPEC ? NUM_1M21 : NUM_GEN>(c[4], x, z[8], z[9]);
}

constexpr int WFM_CH = 8;   // input samples prefetched per chunk (each lane walks its own row)
template <bool SPEC>
__global__ __launch_bounds__(TILE) void k_wfm_front(const float2 *__restrict__ iq, double *U, int n, long n_frames,
                                                    long Lp, int swapped, WfmCoef c)
{
    const int lane = threadIdx.x;
    const long f = (long)blockIdx.x * TILE + lane;
    const bool live = f < n_frames;
    const long fr = live ? f : n_frames - 1;
    const int M = n - 1;
    const float2 *x = iq + (size_t)fr * n;
    double *UL = U + (size_t)(2 * fr) * Lp, *UR = UL + Lp;
    double zlp[6] = {0, 0, 0, 0, 0, 0}, zl2[6] = {0, 0, 0, 0, 0, 0};
    double zpi[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, zlm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double zp1 = 0.0, zdl = 0.0, zdr = 0.0;
    const double SIN_PI = 0x1.1a62633145c07p-53;  // np.sin(np.pi)
    float2 prev = x[0];
    auto step = [&](int i, float2 cur) {
        const double d = (double)disc_sample(cur, prev, 1.0f, swapped != 0);  // :122 (x1.0f is exact)
        prev = cur;
        const double a = wfm_lp<SPEC>(c.lp, d, zlp);                                                   // :126
        const double p = wfm_bp<SPEC>(c.pil, d, zpi);                                                  // :129
        double m = wfm_bp<SPEC>(c.lmr, d, zlm);                                                        // :133
        const double y = lfilter1(1.0, -0.99, p, zp1);                                                 // :130
        const double pil = (y != y) ? y : ((y < 0.0 || (y == 0.0 && __builtin_signbit(y))) ? SIN_PI : 0.0);
        m = __dmul_rn(m, __dmul_rn(2.0, pil));                                                         // :134
        m = wfm_lp<SPEC>(c.lp, m, zl2);                                                                // :137
        const double l = __dmul_rn(__dadd_rn(a, m), 0.5), r = __dmul_rn(__dsub_rn(a, m), 0.5);         // :140-141 (/2 exact)
        const double yl = lfilter1(c.b0d, c.a1d, l, zdl), yr = lfilter1(c.b0d, c.a1d, r, zdr);         // :148-149
        if (live) { UL[EDGE + i] = yl; UR[EDGE + i] = yr; }
    };
    // x[1 + i], i = 0..M-1, in chunks of WFM_CH with the next chunk's loads in flight (x + 1 is 8-byte aligned only)
    float2 b0[WFM_CH], b1[WFM_CH];
    auto loadc = [&](float2 (&b)[WFM_CH], int i0) {
#pragma unroll
        for (int t = 0; t < WFM_CH; t++) b[t] = x[1 + i0 + t];
    };
    const int nfull = M / WFM_CH;
    int i = 0;
    if (nfull > 0) loadc(b0, 0);
    for (int ch = 0; ch < nfull; ch += 2) {
        if (ch + 1 < nfull) loadc(b1, i + WFM_CH);
#pragma unroll
        for (int t = 0; t < WFM_CH; t++) step(i + t, b0[t]);
        i += WFM_CH;
        if (ch + 1 < nfull) {
            if (ch + 2 < nfull) loadc(b0, i + WFM_CH);
#pragma unroll
            for (int t = 0; t < WFM_CH; t++) step(i + t, b1[t]);
            i += WFM_CH;
        }
    }
    for (; i < M; i++) step(i, x[i + 1]);
    if (!live) return;
    // odd extension (scipy _arraytools.odd_ext) of both rows, from this lane's own stores
    __threadfence_block();
    for (int ch = 0; ch < 2; ch++) {
        double *u = (ch ? UR : UL) + EDGE;
        const double u0 = u[0], ul = u[M - 1];
        for (int k = 0; k < EDGE; k++) {
            u[k - EDGE] = __dsub_rn(__dmul_rn(2.0, u0), u[EDGE - k]);
            u[M + k] = __dsub_rn(__dmul_rn(2.0, ul), u[M - 2 - k]);
        }
    }
}

// :157-163 — joint peak normalisation of the two decimated channels, column_stack, and the int16 conversion.
// A: k_nfm_iir's transposed decimated rows ([tile][k][lane], row g = 2f + channel); mxrow[g] = max|row g| (NaN kept).
// planar 1: rows are [2*tile + channel][lane] (fused forward kernel) instead of 2f + channel (k_wfm_front);
// planar 2: A is row-major [2f + channel][k] (small-batch decimator).
__global__ __launch_bounds__(TPB) void k_wfm_finalize(const double *__restrict__ A, const double *__restrict__ mxrow,
                                                      int n_out, long n_frames, int planar, int16_t *__restrict__ pcm,
                                                      double *__restrict__ audio)
{
    const size_t total = (size_t)n_frames * n_out;
    for (size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (size_t)gridDim.x * TPB) {
        const long f = (long)(idx / n_out);
        const int k = (int)(idx - (size_t)f * n_out);
        const long gl = planar == 1 ? (f / TILE) * 2 * TILE + (f % TILE) : 2 * f, gr = planar == 1 ? gl + TILE : gl + 1;
        const double ml = mxrow[gl], mr = mxrow[gr];
        const double mx = (mr > ml) ? mr : ml;  // python max(a, b): b only if b > a
        const double vl = planar == 2 ? A[(size_t)gl * n_out + k] : A[(size_t)(gl / TILE) * n_out * TILE + (size_t)k * TILE + (gl % TILE)];
        const double vr = planar == 2 ? A[(size_t)gr * n_out + k] : A[(size_t)(gr / TILE) * n_out * TILE + (size_t)k * TILE + (gr % TILE)];
        const double l = __ddiv_rn(vl, mx), r = __ddiv_rn(vr, mx);
        if (audio) { audio[2 * idx] = l; audio[2 * idx + 1] = r; }
        if (pcm) {
            const uint16_t a = (uint16_t)pcm16(l), b = (uint16_t)pcm16(r);
            reinterpret_cast<uint32_t *>(pcm)[idx] = (uint32_t)a | ((uint32_t)b << 16);
        }
    }
}

// demodulate_wfm with int(sample_rate / target_rate) == 1: the reference skips its decimate() stage (:152-155) and normalises the de-emphasised
// channels themselves.  One wavefront per row g = 2f + channel of k_wfm_front's extended rows: A[g][k] = U[g][EDGE + k] (row-major: k_wfm_finalize's
// planar 2), mxrow[g] = np.max(np.abs(row)) (a NaN propagates).
__global__ __launch_bounds__(256) void k_wfm_rows_q1(const double *__restrict__ U, long Lp, int M, long n_rows, double *__restrict__ A,
                                                     double *__restrict__ mxrow)
{
    const int lane = threadIdx.x & 63;
    for (long g = (long)blockIdx.x * 4 + (threadIdx.x >> 6); g < n_rows; g += (long)gridDim.x * 4) {
        const double *u = U + (size_t)g * Lp + EDGE;
        double mx = 0.0;
        int has_nan = 0;
        for (int k = lane; k < M; k += 64) {
            const double v = u[k];
            A[(size_t)g * M + k] = v;
            const double av = fabs(v);
            has_nan |= av != av;
            mx = av > mx ? av : mx;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(mx, off);
            mx = o > mx ? o : mx;
            has_nan |= __shfl_xor(has_nan, off);
        }
        if (lane == 0) mxrow[g] = has_nan ? __builtin_nan("") : mx;
    }
}

// ---------------------------------------------------------------------------------------------------
// WFM for SMALL batches (the interactive loop: one frame per call).  k_wfm_fwd steps ~200 float64 instructions per sample
// on ONE lane per frame; here every filter section is its own lane (systolic array, one frame per 16-lane DPP row, four
// frames per wavefront) and the chain runs as two cascaded passes with float64 rows in memory between them:
//   pass 1  d -> LP15k (3 lanes) -> a ; d -> BP pilot (5) -> 1-pole -> y -> pilot value p ; d -> BP 23..53k (5) -> m
//   pass 2  m * (2 p) -> LP15k (3) -> (a +- .)/2 -> de-emphasis, once per channel (two 4-lane groups) -> u_l, u_r,
//           written with SciPy's odd extension around them in the layout the small-batch decimator (k_iir4_sys) reads.
// Every lane executes sosfilt's step on its own section; a 1-pole stage is that step with b1 = +0.0 and its second state
// pinned to -0.0 (v + -0.0 == v bit for bit), which is lfilter's  y = z + b0 x ; z = x*0 - y*a1.  All stages start from a
// zero state, so filling and draining the pipeline with zeros is exact.
// ---------------------------------------------------------------------------------------------------
struct CascLane {
    Biquad c;
    int head;      // 1: takes the staged input instead of its left neighbour's output
    int onepole;   // 1: 1-pole stage (second state pinned to -0.0)
    int mix;       // +1 / -1: input = (a +- left neighbour) * 0.5 (the L/R matrix), 0: plain
    int out_slot;  // >= 0: this lane's output is row out_slot of the write-back buffer
};
struct CascArg { CascLane lane[16]; };
constexpr int CS_T = 64;

// (Kept for A/B builds, -DPSS_EXP_CASC_SAMPLE; the product runs k_wfm_blk below.)  Two wavefronts per workgroup (round 3, as k_am_sys /
// k_iir4_sys): wavefront 0 runs the sample-systolic recurrence, wavefront 1 the
// memory side — raw inputs of block m + 3 requested, block m + 1 computed (the discriminator in pass 1, m * 2p in pass 2) and staged,
// block m - 1 written back — through double-buffered LDS blocks and LDS-only barriers.  As one wavefront the kernel computed four
// discriminator samples per lane and waited for its own loads and stores between any two 64-step blocks.
template <int MODE>  // 1: discriminator -> a, p, m      2: m, p, a -> u_l, u_r (+ odd extension)
__global__ __launch_bounds__(128) void k_wfm_casc(const float2 *__restrict__ iq, double *__restrict__ Aa, double *__restrict__ Pp,
                                                  double *__restrict__ Mm, double *U, int n, long n_frames, long Lp, int swapped,
                                                  CascArg arg)
{
    constexpr int NIN = 2, NOUT = 3;
    __shared__ double ebuf[2][4][NIN][CS_T + 1];
    __shared__ double ybuf[2][4][NOUT + 1][CS_T + 1];  // row NOUT: dump row
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long f0 = (long)blockIdx.x * 4;
    const int M = n - 1;
    const int DA = (MODE == 1) ? 2 : 3, DB = (MODE == 1) ? 5 : 3, DC = 4;  // pipeline delay of each output row
    const long TT = (long)M + 5;  // steps incl. drain of the deepest cascade
    const long nblk = (TT + CS_T - 1) / CS_T;
    const double SIN_PI = 0x1.1a62633145c07p-53;  // np.sin(np.pi)
    if (wave == 1) {
        // ---------------- memory wavefront: lane = sample inside a block ----------------
        struct Raw { float2 x0[4], x1[4]; double m[4], p[4], a[4]; };
        Raw rawA, rawB;
        auto load = [&](long blk, Raw &r) __attribute__((always_inline)) {
            const long i = blk * CS_T + lane;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const long f = f0 + g;
                const bool ok = f < n_frames && i < M;
                if (MODE == 1) {
                    const float2 *x = iq + (size_t)(f < n_frames ? f : 0) * n;
                    r.x0[g] = ok ? x[i] : make_float2(0.0f, 0.0f);
                    r.x1[g] = ok ? x[i + 1] : make_float2(0.0f, 0.0f);
                } else {
                    const size_t o = (size_t)(f < n_frames ? f : 0) * M;
                    r.m[g] = ok ? Mm[o + i] : 0.0;
                    r.p[g] = ok ? Pp[o + i] : 0.0;
                    r.a[g] = (f < n_frames && i - 3 >= 0 && i - 3 < M) ? Aa[o + i - 3] : 0.0;   // a[], aligned with the mixing lanes
                }
            }
        };
        auto stage = [&](long blk, const Raw &r) __attribute__((always_inline)) {
            const long i = blk * CS_T + lane;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const long f = f0 + g;
                double e0 = 0.0, e1 = 0.0;
                if (f < n_frames) {
                    if (MODE == 1) {
                        if (i < M) e0 = (double)disc_sample(r.x1[g], r.x0[g], 1.0f, swapped != 0);  // :122
                    } else {
                        if (i < M) e0 = __dmul_rn(r.m[g], __dmul_rn(2.0, r.p[g]));  // :134
                        if (i - 3 >= 0 && i - 3 < M) e1 = r.a[g];
                    }
                }
                ebuf[blk & 1][g][0][lane] = e0;
                ebuf[blk & 1][g][1][lane] = e1;
            }
        };
        auto writeback = [&](long blk) __attribute__((always_inline)) {   // ybuf[blk & 1] holds what the array produced during block blk
            const long c0 = blk * CS_T;
            const int par = (int)(blk & 1);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const long f = f0 + g;
                if (f >= n_frames) continue;
                const long ia = c0 + lane - DA, ib = c0 + lane - DB, ic = c0 + lane - DC;
                if (MODE == 1) {
                    if (ia >= 0 && ia < M) Aa[(size_t)f * M + ia] = ybuf[par][g][0][lane];
                    if (ib >= 0 && ib < M) {
                        const double y = ybuf[par][g][1][lane];  // :130 np.sin(np.unwrap(np.angle(real))) = 0 or sin(pi)
                        Pp[(size_t)f * M + ib] = (y != y) ? y : ((y < 0.0 || (y == 0.0 && __builtin_signbit(y))) ? SIN_PI : 0.0);
                    }
                    if (ic >= 0 && ic < M) Mm[(size_t)f * M + ic] = ybuf[par][g][2][lane];
                } else {
                    if (ia >= 0 && ia < M) U[(size_t)(2 * f) * Lp + EDGE + ia] = ybuf[par][g][0][lane];
                    if (ib >= 0 && ib < M) U[(size_t)(2 * f + 1) * Lp + EDGE + ib] = ybuf[par][g][1][lane];
                }
            }
        };
        load(0, rawA);
        stage(0, rawA);
        if (nblk > 1) load(1, rawB);
        if (nblk > 2) load(2, rawA);
        fused::lds_barrier();
        auto beside = [&](long m, Raw &r) __attribute__((always_inline)) {     // r: the set holding block m + 1
            if (m + 1 < nblk) stage(m + 1, r);
            if (m + 3 < nblk) load(m + 3, r);
            if (m >= 1) writeback(m - 1);
            fused::lds_barrier();
        };
        for (long m = 0; m < nblk; m += 2) {
            beside(m, rawB);
            if (m + 1 < nblk) beside(m + 1, rawA);
        }
        writeback(nblk - 1);
        if (MODE == 2) {
            // odd extension (scipy _arraytools.odd_ext) of both channel rows from this wavefront's own stores
            __threadfence();
            for (int g = 0; g < 4; g++) {
                const long f = f0 + g;
                if (f >= n_frames) continue;
                for (int ch = 0; ch < 2; ch++) {
                    double *u = U + (size_t)(2 * f + ch) * Lp + EDGE;
                    if (lane < EDGE) {
                        const double u0 = __builtin_nontemporal_load(u), ul = __builtin_nontemporal_load(u + M - 1);
                        const double a = __builtin_nontemporal_load(u + EDGE - lane), b = __builtin_nontemporal_load(u + M - 2 - lane);
                        u[lane - EDGE] = __dsub_rn(__dmul_rn(2.0, u0), a);
                        u[M + lane] = __dsub_rn(__dmul_rn(2.0, ul), b);
                    }
                }
            }
        }
        return;
    }
    // ---------------- recurrence wavefront: one frame per 16-lane DPP row, one filter section per lane ----------------
    const int row = lane >> 4, r = lane & 15;
    const CascLane me = arg.lane[r];
    double z0 = 0.0, z1 = me.onepole ? -0.0 : 0.0, xprev = 0.0;
    const int oslot = me.out_slot >= 0 ? me.out_slot : NOUT;
    fused::lds_barrier();
    for (long blk = 0; blk < nblk; blk++) {
        const long c0 = blk * CS_T;
        const int par = (int)(blk & 1);
        const double *e0p = ebuf[par][row][0], *e1p = ebuf[par][row][1];
        double *const yp = ybuf[par][row][oslot];
        const int cnt = (TT - c0) < CS_T ? (int)(TT - c0) : CS_T;
        auto one = [&](int t, double e, double a) {
            const double from_prev = dpp_row_shr1(xprev);
            double x = me.head ? e : from_prev;
            if (MODE == 2) {
                const double mixed = __dmul_rn(me.mix > 0 ? __dadd_rn(a, from_prev) : __dsub_rn(a, from_prev), 0.5);  // :140-141
                x = me.mix ? mixed : x;
            }
            const double xn = __dadd_rn(__dmul_rn(me.c.b0, x), z0);
            z0 = __dadd_rn(__dsub_rn(__dmul_rn(me.c.b1, x), __dmul_rn(me.c.a1, xn)), z1);
            const double n1 = __dsub_rn(__dmul_rn(me.c.b2, x), __dmul_rn(me.c.a2, xn));
            z1 = me.onepole ? -0.0 : n1;
            xprev = xn;
            yp[t] = xn;
        };
        if (cnt == CS_T) {
            for (int t0 = 0; t0 < CS_T; t0 += 8) {
                double e8[8], a8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { e8[k] = e0p[t0 + k]; a8[k] = (MODE == 2) ? e1p[t0 + k] : 0.0; }
#pragma unroll
                for (int k = 0; k < 8; k++) one(t0 + k, e8[k], a8[k]);
            }
        } else {
            for (int t = 0; t < cnt; t++) one(t, e0p[t], (MODE == 2) ? e1p[t] : 0.0);
        }
        fused::lds_barrier();
    }
}


// ---------------------------------------------------------------------------------------------------
// WFM for small batches, BLOCK-systolic (late round 3): the same two passes as k_wfm_casc, but a lane filters a whole
// 64-sample block of its section and leaves it in LDS for the next section's lane (as k_iir4_sys / k_am_sys do) instead of
// handing every sample to its neighbour by DPP.  The recurrence step is then nothing but sosfilt's nine float64 operations and
// the one-pole pin (no DPP moves, no head / mix selects: ~11 instead of 15-22 instructions on a lone wavefront), and the L / R
// matrix of pass 2 is computed by the memory wavefront, sample-parallel, between the LP15k chain and the de-emphasis lanes.
//   pass 1 (14 lanes per frame): d -> LP15k (3) -> a ; d -> BP pilot (5) -> 1-pole -> y -> pilot value p ; d -> BP 23..53k (5) -> m
//   pass 2 ( 5 lanes per frame): m * (2 p) -> LP15k (3) -> lp ; memory wavefront: (a +- lp) / 2 ; de-emphasis 1-pole per channel -> u_l, u_r
// Same operations on the same sample sequences as k_wfm_casc (zero initial state everywhere), so the same bits.
// ---------------------------------------------------------------------------------------------------
struct BlkLane {
    Biquad c;
    int src, dst;    // LDS row (per frame) read / written; rows >= dbl0 are double-buffered by macro-step parity
    int depth;       // the lane works on block (macro-step - depth); -1: idle lane
    int onepole;
};
struct BlkArg { BlkLane lane[16]; };
constexpr int WB_T = 64, WB_G = 4;

template <int MODE>
__global__ __launch_bounds__(128) void k_wfm_blk(const float2 *__restrict__ iq, double *__restrict__ Aa, double *__restrict__ Pp,
                                                 double *__restrict__ Mm, double *U, int n, long n_frames, long Lp, int swapped,
                                                 BlkArg arg)
{
    // rows per frame.  Pass 1: 0,1 = d (staged, two blocks in flight); 2,3 hand-offs of chain a; 4,5 = a out; 6..10 hand-offs of the pilot
    // chain; 11,12 = y out; 13..16 hand-offs of the m chain; 17,18 = m out.  Pass 2: 0,1 = m*2p; 2,3 hand-offs; 4,5 = lp out; 6,7 = l in;
    // 8,9 = r in; 10,11 = u_l out; 12,13 = u_r out.  (src / dst name the EVEN row of a double-buffered pair; the odd one is + 1.)
    constexpr int NROWS = MODE == 1 ? 19 : 14;
    __shared__ double rows[WB_G][NROWS][WB_T + 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long f0 = (long)blockIdx.x * WB_G;
    const int M = n - 1;
    const long nblk = ((long)M + WB_T - 1) / WB_T;
    constexpr int DEEP = MODE == 1 ? 5 : 4;       // depth of the deepest lane
    const long nstep = nblk + DEEP;
    const double SIN_PI = 0x1.1a62633145c07p-53;  // np.sin(np.pi)
    auto is_dbl = [](int row) { return MODE == 1 ? (row <= 1 || row == 4 || row == 5 || row == 11 || row == 12 || row >= 17)
                                                 : (row <= 1 || row >= 4); };
    if (wave == 1) {
        // ---------------- memory wavefront: lane = sample inside a block ----------------
        struct Raw { float2 x0[WB_G], x1[WB_G]; double m[WB_G], p[WB_G]; };
        Raw rawA, rawB;
        auto load = [&](long blk, Raw &r) __attribute__((always_inline)) {
            const long i = blk * WB_T + lane;
#pragma unroll
            for (int g = 0; g < WB_G; g++) {
                const long f = f0 + g;
                const bool ok = f < n_frames && i < M;
                if (MODE == 1) {
                    const float2 *x = iq + (size_t)(f < n_frames ? f : 0) * n;
                    r.x0[g] = ok ? x[i] : make_float2(0.0f, 0.0f);
                    r.x1[g] = ok ? x[i + 1] : make_float2(0.0f, 0.0f);
                } else {
                    const size_t o = (size_t)(f < n_frames ? f : 0) * M;
                    r.m[g] = ok ? Mm[o + i] : 0.0;
                    r.p[g] = ok ? Pp[o + i] : 0.0;
                }
            }
        };
        auto stage = [&](long blk, const Raw &r) __attribute__((always_inline)) {
            const long i = blk * WB_T + lane;
#pragma unroll
            for (int g = 0; g < WB_G; g++) {
                double e = 0.0;
                if (f0 + g < n_frames && i < M) {
                    if (MODE == 1) e = (double)disc_sample(r.x1[g], r.x0[g], 1.0f, swapped != 0);  // :122
                    else e = __dmul_rn(r.m[g], __dmul_rn(2.0, r.p[g]));                             // :134
                }
                rows[g][(int)(blk & 1)][lane] = e;
            }
        };
        // what the recurrence wavefront finished during macro-step ms (ms >= 0): written back / passed on
        auto drain = [&](long ms) __attribute__((always_inline)) {
            const int par = (int)(ms & 1);
#pragma unroll
            for (int g = 0; g < WB_G; g++) {
                const long f = f0 + g;
                if (f >= n_frames) continue;
                if (MODE == 1) {
                    const long ba = ms - 2, by = ms - 5, bm = ms - 4;
                    const long ia = ba * WB_T + lane, iy = by * WB_T + lane, im = bm * WB_T + lane;
                    if (ba >= 0 && ba < nblk && ia < M) Aa[(size_t)f * M + ia] = rows[g][4 + par][lane];
                    if (by >= 0 && by < nblk && iy < M) {
                        const double y = rows[g][11 + par][lane];  // :130 np.sin(np.unwrap(np.angle(real))) = 0 or sin(pi)
                        Pp[(size_t)f * M + iy] = (y != y) ? y : ((y < 0.0 || (y == 0.0 && __builtin_signbit(y))) ? SIN_PI : 0.0);
                    }
                    if (bm >= 0 && bm < nblk && im < M) Mm[(size_t)f * M + im] = rows[g][17 + par][lane];
                } else {
                    // the LP15k chain finished block ms - 2: L / R matrix (:140-141), staged for the de-emphasis lanes (macro-step ms + 2)
                    const long bl = ms - 2, il = bl * WB_T + lane;
                    if (bl >= 0 && bl < nblk) {
                        double l = 0.0, r = 0.0;
                        if (il < M) {
                            const double a = Aa[(size_t)f * M + il], lp = rows[g][4 + par][lane];
                            l = __dmul_rn(__dadd_rn(a, lp), 0.5);
                            r = __dmul_rn(__dsub_rn(a, lp), 0.5);
                        }
                        rows[g][6 + par][lane] = l;      // read by the de-emphasis lanes at macro-step ms + 2 (same parity)
                        rows[g][8 + par][lane] = r;
                    }
                    const long bu = ms - 4, iu = bu * WB_T + lane;
                    if (bu >= 0 && bu < nblk && iu < M) {
                        U[(size_t)(2 * f) * Lp + EDGE + iu] = rows[g][10 + par][lane];
                        U[(size_t)(2 * f + 1) * Lp + EDGE + iu] = rows[g][12 + par][lane];
                    }
                }
            }
        };
        load(0, rawA);
        stage(0, rawA);
        if (nblk > 1) load(1, rawB);
        if (nblk > 2) load(2, rawA);
        fused::lds_barrier();
        auto beside = [&](long m, Raw &r) __attribute__((always_inline)) {     // r: the set holding block m + 1
            if (m + 1 < nblk) stage(m + 1, r);
            if (m + 3 < nblk) load(m + 3, r);
            if (m >= 1) drain(m - 1);
            fused::lds_barrier();
        };
        for (long m = 0; m < nstep; m += 2) {
            beside(m, rawB);
            if (m + 1 < nstep) beside(m + 1, rawA);
        }
        drain(nstep - 1);
        if (MODE == 2) {
            // odd extension (scipy _arraytools.odd_ext) of both channel rows from this wavefront's own stores
            __threadfence();
            for (int g = 0; g < WB_G; g++) {
                const long f = f0 + g;
                if (f >= n_frames) continue;
                for (int ch = 0; ch < 2; ch++) {
                    double *u = U + (size_t)(2 * f + ch) * Lp + EDGE;
                    if (lane < EDGE) {
                        const double u0 = __builtin_nontemporal_load(u), ul = __builtin_nontemporal_load(u + M - 1);
                        const double a = __builtin_nontemporal_load(u + EDGE - lane), b = __builtin_nontemporal_load(u + M - 2 - lane);
                        u[lane - EDGE] = __dsub_rn(__dmul_rn(2.0, u0), a);
                        u[M + lane] = __dsub_rn(__dmul_rn(2.0, ul), b);
                    }
                }
            }
        }
        return;
    }
    // ---------------- recurrence wavefront: one frame per 16 lanes, one filter section per lane ----------------
    const int g = lane >> 4;
    const BlkLane me = arg.lane[lane & 15];
    const bool lane_on = me.depth >= 0;
    const bool src_dbl = is_dbl(me.src), dst_dbl = is_dbl(me.dst);
    double z0 = 0.0, z1 = me.onepole ? -0.0 : 0.0;
    auto step = [&](double x) {
        const double xn = __dadd_rn(__dmul_rn(me.c.b0, x), z0);
        z0 = __dadd_rn(__dsub_rn(__dmul_rn(me.c.b1, x), __dmul_rn(me.c.a1, xn)), z1);
        const double n1 = __dsub_rn(__dmul_rn(me.c.b2, x), __dmul_rn(me.c.a2, xn));
        z1 = me.onepole ? -0.0 : n1;
        return xn;
    };
    fused::lds_barrier();
    for (long m = 0; m < nstep; m++) {
        const long blk = m - me.depth;
        const bool active = lane_on && blk >= 0 && blk < nblk;
        const int par = (int)(m & 1);
        const double *src = rows[g][me.src + (src_dbl ? par : 0)];
        double *dst = rows[g][me.dst + (dst_dbl ? par : 0)];
        const int cnt = !active ? 0 : (((long)M - blk * WB_T) < WB_T ? (int)((long)M - blk * WB_T) : WB_T);
        // in-place hand-off: all lanes walk the block in lockstep, reads of a group of eight before its writes (k_iir4_sys)
        if (__all(!active || cnt == WB_T)) {
            if (active) {
                double ea[8], eb[8], y8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) ea[k] = src[k];
#pragma unroll 1
                for (int t0 = 0; t0 < WB_T; t0 += 16) {
#pragma unroll
                    for (int k = 0; k < 8; k++) eb[k] = src[t0 + 8 + k];
#pragma unroll
                    for (int k = 0; k < 8; k++) y8[k] = step(ea[k]);
#pragma unroll
                    for (int k = 0; k < 8; k++) dst[t0 + k] = y8[k];
                    if (t0 + 16 < WB_T) {
#pragma unroll
                        for (int k = 0; k < 8; k++) ea[k] = src[t0 + 16 + k];
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) y8[k] = step(eb[k]);
#pragma unroll
                    for (int k = 0; k < 8; k++) dst[t0 + 8 + k] = y8[k];
                }
            }
        } else {
            for (int t = 0; t < WB_T; t++) {
                if (t < cnt) {
                    const double e = src[t];
                    dst[t] = step(e);
                }
            }
        }
        fused::lds_barrier();
    }
}


// ---------------------------------------------------------------------------------------------------
// Both WFM passes in ONE block-systolic array (late round 3): 19 lanes per frame (the 14 of pass 1, the LP15k chain and the two
// de-emphasis lanes of pass 2), three frames per wavefront.  The memory wavefront feeds pass 2 from pass 1's LDS rows — pilot value
// and m * 2p when the pilot chain has finished a block, the L / R matrix when the second LP15k chain has — so a, p and m never go
// through global memory and the second pass's latency (~1.1 ms for a 32768-sample frame) disappears.  Every buffer a lane or the memory
// wavefront writes is a ring indexed by BLOCK number, as long as its value has to live (a: 9 blocks, m: 3, the rest 2).
// Timeline of block b (macro-steps): staged b-1 | a ready b+2 | m ready b+4 | y ready b+5 | m*2p staged b+6 | LP15k b+7..b+9 |
// matrix staged b+10 | de-emphasis b+11 | u_l, u_r written back b+12.
// ---------------------------------------------------------------------------------------------------
struct MrgLane {
    Biquad c;
    int src, src_ring, dst, dst_ring;   // first row of the ring read / written, ring length (1: a plain hand-off row)
    int depth;                          // the lane works on block (macro-step - depth); -1: idle lane
    int onepole;
    double zi0, zi1;                    // FWD decimator lanes: initial state per unit of the sequence's first sample (sosfilt_zi); else 0
    int chan;                           // FWD decimator lanes: 0 / 1 = left / right (whose first sample scales zi); else -1
};
struct MrgArg { MrgLane lane[27]; };
constexpr int WM_T = 64;
// rows per frame: D1 0-1 | a hand-offs 2-3 | A 4-12 | pilot hand-offs 13-17 | Y 18-19 | m hand-offs 20-23 | Mo 24-26 | D2 27-28 |
// lp hand-offs 29-30 | LP 31-32 | DL 33-34 | DR 35-36 | UL 37-39 | UR 40-42 | (FWD) EL 43-44 | ER 45-46 | decimator hand-offs 47-49 (left),
// 50-52 (right) | YL 53-54 | YR 55-56
constexpr int WM_D1 = 0, WM_A = 4, WM_Y = 18, WM_MO = 24, WM_D2 = 27, WM_LP = 31, WM_DL = 33, WM_DR = 35, WM_UL = 37, WM_UR = 40,
              WM_EL = 43, WM_ER = 45, WM_YL = 53, WM_YR = 55;

// FWD: the FORWARD half of the zero-phase decimator runs in the same array (8 more lanes per frame, two frames per wavefront): the
// memory wavefront builds SciPy's odd extension (_arraytools.odd_ext: 27 reflected samples either side) from the u_l / u_r rings block by
// block — the extended sequence's blocks are 27 samples out of step with u's — and writes y_fwd in the layout the backward launch of
// k_iir4_sys reads.  u itself then never leaves the CU either.
template <bool FWD>
__global__ __launch_bounds__(128) void k_wfm_mrg(const float2 *__restrict__ iq, double *UY, int n, long n_frames, long row_stride, int swapped,
                                                 MrgArg arg)
{
    constexpr int NL = FWD ? 27 : 19, G = FWD ? 2 : 3, NROWS = FWD ? 57 : 43;
    __shared__ double rows[G][NROWS][WM_T + 1];
    __shared__ double x0s[G][2];                 // FWD: first sample of each channel's extended sequence
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long f0 = (long)blockIdx.x * G;
    const int M = n - 1;
    const long L = (long)M + 2 * EDGE;
    const long nblk = ((long)M + WM_T - 1) / WM_T;
    const long nblk_e = (L + WM_T - 1) / WM_T;   // blocks of the odd-extended sequence
    constexpr int DEEP = FWD ? 16 : 11;
    const long nstep = (FWD ? nblk_e : nblk) + DEEP;
    const double SIN_PI = 0x1.1a62633145c07p-53;  // np.sin(np.pi)
    if (wave == 1) {
        // ---------------- memory wavefront: lane = sample inside a block ----------------
        struct Raw { float2 x0[G], x1[G]; };
        Raw rawA, rawB;
        auto load = [&](long blk, Raw &r) __attribute__((always_inline)) {
            const long i = blk * WM_T + lane;
#pragma unroll
            for (int g = 0; g < G; g++) {
                const long f = f0 + g;
                const bool ok = f < n_frames && i < M;
                const float2 *x = iq + (size_t)(f < n_frames ? f : 0) * n;
                r.x0[g] = ok ? x[i] : make_float2(0.0f, 0.0f);
                r.x1[g] = ok ? x[i + 1] : make_float2(0.0f, 0.0f);
            }
        };
        auto stage = [&](long blk, const Raw &r) __attribute__((always_inline)) {
            const long i = blk * WM_T + lane;
#pragma unroll
            for (int g = 0; g < G; g++) {
                double e = 0.0;
                if (f0 + g < n_frames && i < M) e = (double)disc_sample(r.x1[g], r.x0[g], 1.0f, swapped != 0);  // :122
                rows[g][WM_D1 + (int)(blk & 1)][lane] = e;
            }
        };
        // after macro-step ms: what the recurrence wavefront has finished is passed on / written back
        auto drain = [&](long ms) __attribute__((always_inline)) {
            const long b6 = ms - 5, b9 = ms - 9, b11 = ms - 11, b16 = ms - 16;
#pragma unroll
            for (int g = 0; g < G; g++) {
                const long f = f0 + g;
                if (f >= n_frames) continue;
                if (b6 >= 0 && b6 < nblk) {   // pilot chain done with block b6 (m since the step before): m * (2 p), :130-134
                    const double y = rows[g][WM_Y + (int)(b6 & 1)][lane];
                    const double pv = (y != y) ? y : ((y < 0.0 || (y == 0.0 && __builtin_signbit(y))) ? SIN_PI : 0.0);
                    const double mv = rows[g][WM_MO + (int)(b6 % 3)][lane];
                    rows[g][WM_D2 + (int)(b6 & 1)][lane] = (b6 * WM_T + lane < M) ? __dmul_rn(mv, __dmul_rn(2.0, pv)) : 0.0;
                }
                if (b9 >= 0 && b9 < nblk) {   // second LP15k chain done with block b9: L / R matrix (:140-141) for the de-emphasis lanes
                    const double a = rows[g][WM_A + (int)(b9 % 9)][lane], lp = rows[g][WM_LP + (int)(b9 & 1)][lane];
                    const bool in = b9 * WM_T + lane < M;
                    rows[g][WM_DL + (int)(b9 & 1)][lane] = in ? __dmul_rn(__dadd_rn(a, lp), 0.5) : 0.0;
                    rows[g][WM_DR + (int)(b9 & 1)][lane] = in ? __dmul_rn(__dsub_rn(a, lp), 0.5) : 0.0;
                }
                if constexpr (!FWD) {
                    if (b11 >= 0 && b11 < nblk) {
                        const long iu = b11 * WM_T + lane;
                        if (iu < M) {
                            UY[(size_t)(2 * f) * row_stride + EDGE + iu] = rows[g][WM_UL + (int)(b11 % 3)][lane];
                            UY[(size_t)(2 * f + 1) * row_stride + EDGE + iu] = rows[g][WM_UR + (int)(b11 % 3)][lane];
                        }
                    }
                } else {
                    if (b11 >= 0 && b11 < nblk_e) {
                        // block b11 of the odd-extended sequences (scipy _arraytools.odd_ext): ext[p] = 2 u[0] - u[27 - p] (p < 27),
                        // u[p - 27], 2 u[M-1] - u[2M + 25 - p] (p >= 27 + M); u[i] sits in ring row (i / 64) % 3, column i % 64
                        const long pidx = b11 * WM_T + lane;
#pragma unroll
                        for (int ch = 0; ch < 2; ch++) {
                            const int ub = ch ? WM_UR : WM_UL;
                            auto uat = [&](long i) { return rows[g][ub + (int)((i / WM_T) % 3)][(int)(i % WM_T)]; };
                            double v = 0.0;
                            if (pidx < EDGE) v = __dsub_rn(__dmul_rn(2.0, uat(0)), uat(EDGE - pidx));
                            else if (pidx < EDGE + M) v = uat(pidx - EDGE);
                            else if (pidx < L) v = __dsub_rn(__dmul_rn(2.0, uat(M - 1)), uat(2L * M + 25 - pidx));
                            rows[g][(ch ? WM_ER : WM_EL) + (int)(b11 & 1)][lane] = v;
                            if (pidx == 0) x0s[g][ch] = v;
                        }
                    }
                    if (b16 >= 0 && b16 < nblk_e) {   // forward decimator pass done with block b16: y_fwd rows [2 f + channel][L]
                        const long iy = b16 * WM_T + lane;
                        if (iy < L) {
                            UY[(size_t)(2 * f) * row_stride + iy] = rows[g][WM_YL + (int)(b16 & 1)][lane];
                            UY[(size_t)(2 * f + 1) * row_stride + iy] = rows[g][WM_YR + (int)(b16 & 1)][lane];
                        }
                    }
                }
            }
        };
        load(0, rawA);
        stage(0, rawA);
        if (nblk > 1) load(1, rawB);
        if (nblk > 2) load(2, rawA);
        fused::lds_barrier();
        auto beside = [&](long m, Raw &r) __attribute__((always_inline)) {     // r: the set holding block m + 1
            if (m + 1 < nblk) stage(m + 1, r);
            if (m + 3 < nblk) load(m + 3, r);
            if (m >= 1) drain(m - 1);
            fused::lds_barrier();
        };
        for (long m = 0; m < nstep; m += 2) {
            beside(m, rawB);
            if (m + 1 < nstep) beside(m + 1, rawA);
        }
        drain(nstep - 1);
        if constexpr (!FWD) {
            // odd extension (scipy _arraytools.odd_ext) of both channel rows from this wavefront's own stores
            __threadfence();
            for (int g = 0; g < G; g++) {
                const long f = f0 + g;
                if (f >= n_frames) continue;
                for (int ch = 0; ch < 2; ch++) {
                    double *u = UY + (size_t)(2 * f + ch) * row_stride + EDGE;
                    if (lane < EDGE) {
                        const double u0 = __builtin_nontemporal_load(u), ul = __builtin_nontemporal_load(u + M - 1);
                        const double a = __builtin_nontemporal_load(u + EDGE - lane), b = __builtin_nontemporal_load(u + M - 2 - lane);
                        u[lane - EDGE] = __dsub_rn(__dmul_rn(2.0, u0), a);
                        u[M + lane] = __dsub_rn(__dmul_rn(2.0, ul), b);
                    }
                }
            }
        }
        return;
    }
    // ---------------- recurrence wavefront: NL lanes per frame, one filter section per lane ----------------
    const int g_raw = lane / NL, role = lane - NL * g_raw;
    const bool lane_in = g_raw < G;
    const int g = lane_in ? g_raw : 0;
    const MrgLane me = arg.lane[role];
    const bool lane_on = lane_in && me.depth >= 0;
    const long my_nblk = (FWD && me.chan >= 0) ? nblk_e : nblk;   // the decimator lanes walk the extended sequence
    const long my_len = (FWD && me.chan >= 0) ? L : (long)M;
    double z0 = 0.0, z1 = me.onepole ? -0.0 : 0.0;
    auto step = [&](double x) {
        const double xn = __dadd_rn(__dmul_rn(me.c.b0, x), z0);
        z0 = __dadd_rn(__dsub_rn(__dmul_rn(me.c.b1, x), __dmul_rn(me.c.a1, xn)), z1);
        const double n1 = __dsub_rn(__dmul_rn(me.c.b2, x), __dmul_rn(me.c.a2, xn));
        z1 = me.onepole ? -0.0 : n1;
        return xn;
    };
    int si = 0, di = 0;
    fused::lds_barrier();
    for (long m = 0; m < nstep; m++) {
        const long blk = m - me.depth;
        const bool active = lane_on && blk >= 0 && blk < my_nblk;
        const double *src = rows[g][me.src + si];      // si, di = blk % ring, counted up (a 64-bit modulo by a lane's ring length here cost
        double *dst = rows[g][me.dst + di];            // ~250 instructions per macro-step: a fifth of the step on a lone wavefront)
        const int cnt = !active ? 0 : ((my_len - blk * WM_T) < WM_T ? (int)(my_len - blk * WM_T) : WM_T);
        if (FWD && active && blk == 0 && me.chan >= 0) {   // sosfiltfilt: every section starts from zi * (first sample of the extended sequence)
            const double x0 = x0s[g][me.chan];
            z0 = __dmul_rn(me.zi0, x0);
            z1 = __dmul_rn(me.zi1, x0);
        }
        // in-place hand-off: all lanes walk the block in lockstep, reads of a group of eight before its writes (k_iir4_sys)
        if (__all(!active || cnt == WM_T)) {
            if (active) {
                double ea[8], eb[8], y8[8];
#pragma unroll
                for (int k = 0; k < 8; k++) ea[k] = src[k];
#pragma unroll 1
                for (int t0 = 0; t0 < WM_T; t0 += 16) {
#pragma unroll
                    for (int k = 0; k < 8; k++) eb[k] = src[t0 + 8 + k];
#pragma unroll
                    for (int k = 0; k < 8; k++) y8[k] = step(ea[k]);
#pragma unroll
                    for (int k = 0; k < 8; k++) dst[t0 + k] = y8[k];
                    if (t0 + 16 < WM_T) {
#pragma unroll
                        for (int k = 0; k < 8; k++) ea[k] = src[t0 + 16 + k];
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) y8[k] = step(eb[k]);
#pragma unroll
                    for (int k = 0; k < 8; k++) dst[t0 + 8 + k] = y8[k];
                }
            }
        } else {
            for (int t = 0; t < WM_T; t++) {
                if (t < cnt) {
                    const double e = src[t];
                    dst[t] = step(e);
                }
            }
        }
        if (active) {
            si = si + 1 == me.src_ring ? 0 : si + 1;
            di = di + 1 == me.dst_ring ? 0 : di + 1;
        }
        fused::lds_barrier();
    }
}

}  // namespace
#include "pss_wfm_fused.h"
namespace {

// bandpass_filter (signal_processing.py:34-42) on real float64 rows: scipy.signal.sosfilt, zero initial state, one lane per
// row (the recurrence is serial in time).  Used by the reference's decoders (decoders.py:100-101) on single audio buffers —
// a batch of rows is where a GPU makes sense; a single row runs on one lane and is there for interface completeness.
struct SosArg { Biquad s[8]; int nsec; };
__global__ __launch_bounds__(TILE) void k_sosfilt(const double *__restrict__ x, double *__restrict__ y, int n, long n_rows, SosArg c)
{
    const long r = (long)blockIdx.x * TILE + threadIdx.x;
    if (r >= n_rows) return;
    const double *xr = x + (size_t)r * n;
    double *yr = y + (size_t)r * n;
    double z[16];
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.0;
    for (int i = 0; i < n; i++) {
        double v = xr[i];
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++)
            if (s2 < c.nsec) v = biquad_step(c.s[s2], v, z[2 * s2], z[2 * s2 + 1]);
        yr[i] = v;
    }
}

// decode_afsk (decoders.py:94-112) after the two band-pass rows exist: energy of each band over every bit period (np.sum of
// the squares: float64 pairwise tree, 8192-element chunks added in order) and the comparison.  One thread per (row, bit).
__device__ double pairwise_sq_f64(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res = __dadd_rn(res, __dmul_rn(a[i], a[i]));
        return res;
    }
    if (n <= 128) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = __dmul_rn(a[j], a[j]);
        int i;
        for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] = __dadd_rn(r[j], __dmul_rn(a[i + j], a[i + j]));
        }
        double res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                               __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
        for (; i < n; i++) res = __dadd_rn(res, __dmul_rn(a[i], a[i]));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __dadd_rn(pairwise_sq_f64(a, n2), pairwise_sq_f64(a + n2, n - n2));
}
// ---- classify_signal (signal_processing.py:296-322 with `welch` bound to scipy.signal.welch) ----------
// Two kernels, one workgroup per frame.
// k_cls_modidx: estimate_modulation_index (:283-293), float32 exactly as NumPy evaluates it: np.abs / np.angle (SVML
//   atan2f model), np.unwrap's float32 arithmetic — its cumsum is a SEQUENTIAL float32 sum, so one lane walks it, chunk
//   by chunk through LDS — np.diff, and both np.var's on NumPy's own summation trees (frame_rsum).  The unwrapped phase
//   goes through a global scratch (float32 [frame][n]).
// k_cls_welch: Welch PSD (SciPy _spectral_helper: periodic Hann, 1024-sample segments every 512, per-segment mean
//   removed, two-sided, density scaling, mean over segments) with the segment FFT in float64 (the reference's is
//   pocketfft float32: its noise, ~1e-7 of the peak bin, is the tolerance of this row), then estimate_bandwidth
//   (:267-280, bins in FFT order), spectral flatness (:304) and the decision tree (:307-322).
constexpr int CLS_NP = 1024, CLS_STEP = 512, CLS_CH = 2048;

__device__ __forceinline__ float np_modf32(float a, float b)  // npy_remainderf
{
    float m = fmodf(a, b);
    if (m != 0.0f) { if ((b < 0.0f) != (m < 0.0f)) m = __fadd_rn(m, b); }
    else m = copysignf(0.0f, b);
    return m;
}

// LDS: [part: 8 * leaves floats][val][pbuf: CLS_CH + 1][cbuf: CLS_CH][carry][plans]
__global__ __launch_bounds__(256) void k_cls_modidx(const float2 *__restrict__ iq, int n, long n_frames, RedPlan rn, RedPlan rm,
                                                    int part_slots, int val_slots, float *__restrict__ up_all,
                                                    float *__restrict__ mi_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *part = reinterpret_cast<float *>(smem), *val = part + part_slots;
    float *pbuf = val + val_slots, *cbuf = pbuf + CLS_CH + 4, *carry = cbuf + CLS_CH;  // cbuf 16-byte aligned (val_slots % 4 == 0)
    {
        int *cur = reinterpret_cast<int *>(carry + 1);
        plan_to_lds(rn.full, cur);
        plan_to_lds(rn.tail, cur);
        plan_to_lds(rm.full, cur);
        plan_to_lds(rm.tail, cur);
        __syncthreads();
    }
    const int tid = threadIdx.x, T = blockDim.x;
    const float PI32 = (float)M_PI, TWOPI32 = (float)(2.0 * M_PI), NPI32 = (float)(-M_PI);
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *x = iq + (size_t)f * n;
        float *up = up_all + (size_t)f * n;
        const float fn = (float)n, fm = (float)(n - 1);
        // :286, :290  np.var(np.abs(samples))
        auto A = [&](int i) { const float2 v = x[i]; return cabsf_np(v.x, v.y); };
        const float amean = __fdiv_rn(frame_rsum(rn, part, val, A), fn);
        const float amp_var = __fdiv_rn(frame_rsum(rn, part, val, [&](int i) { const float d = __fsub_rn(A(i), amean); return __fmul_rn(d, d); }), fn);
        // :287  np.unwrap(np.angle(samples)) -> up[]
        if (tid == 0) *carry = 0.0f;
        for (int base = 0; base < n - 1; base += CLS_CH) {
            const int cnt = (n - 1 - base) < CLS_CH ? (n - 1 - base) : CLS_CH;
            for (int j = tid; j <= cnt; j += T) { const float2 v = x[base + j]; pbuf[j] = atan2f_svml(v.y, v.x); }
            __syncthreads();
            for (int j = tid; j < cnt; j += T) {
                const float dd = __fsub_rn(pbuf[j + 1], pbuf[j]);
                float ddmod = __fadd_rn(np_modf32(__fsub_rn(dd, NPI32), TWOPI32), NPI32);
                if (ddmod == NPI32 && dd > 0.0f) ddmod = PI32;
                float corr = __fsub_rn(ddmod, dd);
                if (fabsf(dd) < PI32) corr = 0.0f;
                cbuf[j] = corr;
            }
            __syncthreads();
            if (tid == 0) {  // ph_correct.cumsum(): sequential float32
                float cs = *carry;
                int j = 0;
                for (; j + 8 <= cnt; j += 8) {  // wide LDS accesses: the chain is then bound by the 8 dependent adds, not by LDS latency
                    float4 a = *reinterpret_cast<const float4 *>(cbuf + j), b = *reinterpret_cast<const float4 *>(cbuf + j + 4);
                    a.x = cs = __fadd_rn(cs, a.x); a.y = cs = __fadd_rn(cs, a.y); a.z = cs = __fadd_rn(cs, a.z); a.w = cs = __fadd_rn(cs, a.w);
                    b.x = cs = __fadd_rn(cs, b.x); b.y = cs = __fadd_rn(cs, b.y); b.z = cs = __fadd_rn(cs, b.z); b.w = cs = __fadd_rn(cs, b.w);
                    *reinterpret_cast<float4 *>(cbuf + j) = a;
                    *reinterpret_cast<float4 *>(cbuf + j + 4) = b;
                }
                for (; j < cnt; j++) { cs = __fadd_rn(cs, cbuf[j]); cbuf[j] = cs; }
                *carry = cs;
                if (base == 0) up[0] = pbuf[0];
            }
            __syncthreads();
            for (int j = tid; j < cnt; j += T) up[base + j + 1] = __fadd_rn(pbuf[j + 1], cbuf[j]);
            __syncthreads();
        }
        if (n == 1 && tid == 0) { const float2 v = x[0]; up[0] = atan2f_svml(v.y, v.x); }
        __threadfence_block();
        __syncthreads();
        // :291  np.var(np.diff(phase_env))
        float mi = NAN;  // np.var of an empty array
        if (n > 1) {
            auto D = [&](int i) { return __fsub_rn(up[i + 1], up[i]); };
            const float dmean = __fdiv_rn(frame_rsum(rm, part, val, D), fm);
            const float phase_var = __fdiv_rn(frame_rsum(rm, part, val, [&](int i) { const float d = __fsub_rn(D(i), dmean); return __fmul_rn(d, d); }), fm);
            mi = __fdiv_rn(phase_var, __fadd_rn(amp_var, (float)1e-10));  // :293
        }
        if (tid == 0) mi_out[f] = mi;
        __syncthreads();
    }
}

// The part of classify_signal behind the PSD (np bins in FFT order, float32, in LDS): estimate_bandwidth (:267-280), spectral
// flatness (:304), the decision tree (:307-322).  dbv: np floats, red: 16 doubles, redi: 8 ints of LDS.  All 256 threads.
__device__ __forceinline__ void cls_features(const float *psd, float *dbv, double *red, int *redi, int np, long f, double fs,
                                             const float *__restrict__ mi_in, int32_t *__restrict__ label,
                                             double *__restrict__ bw_out, float *__restrict__ flat_out,
                                             float *__restrict__ psd_out)
{
    const int tid = threadIdx.x;
    // estimate_bandwidth (:267-280): 10 log10(psd + 1e-10), bins above max - 20 dB, first / last in FFT order
    float mx = -INFINITY;
    bool nanv = false;
    double slog = 0.0, spsd = 0.0;
    for (int k = tid; k < np; k += 256) {
        const float v = __fadd_rn(psd[k], (float)1e-10);
        const float d = __fmul_rn(10.0f, log10f_np(v));               // np.log10 float32 = SVML (pss_device.h)
        dbv[k] = d;
        nanv = nanv || (d != d);
        mx = d > mx ? d : mx;
        slog += (double)(float)log((double)v);                       // :304 np.log(psd + 1e-10), float32
        spsd += (double)psd[k];
        if (psd_out) psd_out[(size_t)f * CLS_NP + k] = psd[k];
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
        nanv = nanv || __shfl_xor((int)nanv, off);
        slog += __shfl_xor(slog, off);
        spsd += __shfl_xor(spsd, off);
    }
    if ((tid & 63) == 0) { red[tid >> 6] = (double)mx; red[4 + (tid >> 6)] = nanv ? 1.0 : 0.0; red[8 + (tid >> 6)] = slog; red[12 + (tid >> 6)] = spsd; }
    __syncthreads();
    mx = (float)fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    nanv = (red[4] + red[5] + red[6] + red[7]) != 0.0;
    slog = (red[8] + red[9]) + (red[10] + red[11]);
    spsd = (red[12] + red[13]) + (red[14] + red[15]);
    const float thr = nanv ? NAN : __fadd_rn(mx, -20.0f);            // np.max propagates NaN: then nothing compares above it
    int first = np, last = -1;
    for (int k = tid; k < np; k += 256)
        if (dbv[k] > thr) { first = k < first ? k : first; last = k > last ? k : last; }
    for (int off = 32; off > 0; off >>= 1) {
        const int a = __shfl_xor(first, off), b = __shfl_xor(last, off);
        first = a < first ? a : first;
        last = b > last ? b : last;
    }
    if ((tid & 63) == 0) { redi[tid >> 6] = first; redi[4 + (tid >> 6)] = last; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 0; k < 4; k++) { first = redi[k] < first ? redi[k] : first; last = redi[4 + k] > last ? redi[4 + k] : last; }
        double bw = 0.0;
        if (last >= 0) {
            const double val = 1.0 / ((double)np * (1.0 / fs));      // np.fft.fftfreq(n, d) = integers * (1 / (n d))
            const int npos = (np - 1) / 2 + 1;                        // bins 0 .. npos - 1 are the non-negative frequencies
            bw = (double)(last < npos ? last : last - np) * val - (double)(first < npos ? first : first - np) * val;
        }
        const float gm = (float)exp((double)(float)(slog / (double)np));
        const float flat = __fdiv_rn(gm, (float)(spsd / (double)np));
        const float mi = mi_in[f];
        int lab = PSS_CLASS_UNKNOWN;                                  // :307-322; np.float32 vs Python float compares in float32
        if (bw > 150e3) lab = mi > 0.8f ? PSS_CLASS_FM_BROADCAST : PSS_CLASS_UNKNOWN;
        else if (8e3 <= bw && bw <= 16e3) lab = mi < 0.3f ? PSS_CLASS_NARROW_FM : PSS_CLASS_UNKNOWN;
        else if (2e3 <= bw && bw <= 3e3) lab = flat < 0.2f ? PSS_CLASS_SSB : PSS_CLASS_UNKNOWN;   // (:314 AM_BROADCAST is unreachable)
        else if (flat > 0.7f) lab = PSS_CLASS_DIGITAL;
        if (label) label[f] = lab;
        if (bw_out) bw_out[f] = bw;
        if (flat_out) flat_out[f] = flat;
    }
    __syncthreads();
}

// LDS: [buf: 1024 double2][tw: 512 double2][cpart][cval][plan]; psd (float) and the reduction slots alias buf afterwards
__global__ __launch_bounds__(256) void k_cls_welch(const float2 *__restrict__ iq, int n, long n_frames, double fs, RedPlan cp,
                                                   int part_slots, int val_slots, const float *__restrict__ win, float scale,
                                                   const float *__restrict__ mi_in, int32_t *__restrict__ label,
                                                   double *__restrict__ bw_out, float *__restrict__ flat_out,
                                                   float *__restrict__ psd_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *buf = reinterpret_cast<double2 *>(smem), *tw = buf + CLS_NP;
    float2 *cpart = reinterpret_cast<float2 *>(tw + CLS_NP / 2), *cval = cpart + part_slots;
    {
        int *cur = reinterpret_cast<int *>(cval + val_slots);
        plan_to_lds(cp.full, cur);
        plan_to_lds(cp.tail, cur);
    }
    const int tid = threadIdx.x;
    for (int k = tid; k < CLS_NP / 2; k += 256) {
        double sn, cs;
        sincospi(-2.0 * (double)k / (double)CLS_NP, &sn, &cs);
        tw[k] = make_double2(cs, sn);
    }
    float w[4];
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = win[tid + 256 * j];
    __syncthreads();
    const long nseg = (n - CLS_NP) / CLS_STEP + 1;
    const bool wave_tree = cp.tail.wave_tree && !cp.n_full;
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *x = iq + (size_t)f * n;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};  // positions tid + 256 j of the bit-reversed spectrum
        for (long sg = 0; sg < nseg; sg++) {
            const float2 *xs = x + sg * CLS_STEP;
            // detrend: data - mean(data), complex64.  1024 complex = NumPy's perfectly balanced tree: each wavefront folds it with
            // shuffles on its own (redundantly, no barrier); the general LDS walk remains for plans that are not (never here)
            const float2 m = wave_tree ? wg_csum<true>(cp.tail, cpart, cval, [&](int i) { return xs[i]; }, make_float2(0.0f, 0.0f), false)
                                       : frame_csum(cp, cpart, cval, [&](int i) { return xs[i]; });
            const float mr = __fdiv_rn(m.x, (float)CLS_NP), mim = __fdiv_rn(m.y, (float)CLS_NP);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float2 v = xs[tid + 256 * j];
                buf[tid + 256 * j] = make_double2((double)__fmul_rn(w[j], __fsub_rn(v.x, mr)), (double)__fmul_rn(w[j], __fsub_rn(v.y, mim)));
            }
            __syncthreads();
            // radix-2 decimation in frequency, natural order in, bit-reversed order out
            for (int st = 9; st >= 0; st--) {
                const int half = 1 << st;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int b = tid + 256 * r, grp = b >> st, pos = b & (half - 1);
                    const int i0 = (grp << (st + 1)) + pos, i1 = i0 + half;
                    const double2 a = buf[i0], c = buf[i1], t = tw[pos << (9 - st)];
                    const double dr = a.x - c.x, di = a.y - c.y;
                    buf[i0] = make_double2(a.x + c.x, a.y + c.y);
                    buf[i1] = make_double2(dr * t.x - di * t.y, dr * t.y + di * t.x);
                }
                __syncthreads();
            }
#pragma unroll
            for (int j = 0; j < 4; j++) { const double2 v = buf[tid + 256 * j]; acc[j] += v.x * v.x + v.y * v.y; }
            __syncthreads();
        }
        float *psd = reinterpret_cast<float *>(buf);                    // [1024], FFT order
        float *dbv = psd + CLS_NP;                                       // [1024]
        double *red = reinterpret_cast<double *>(dbv + CLS_NP);          // [4][4] wave partials
        int *redi = reinterpret_cast<int *>(red + 16);                   // [2][4]
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int bin = (int)(__brev((unsigned)(tid + 256 * j)) >> 22);
            psd[bin] = (float)(acc[j] / (double)nseg * (double)scale);
        }
        __syncthreads();
        cls_features(psd, dbv, red, redi, CLS_NP, f, fs, mi_in, label, bw_out, flat_out, psd_out);
    }
}

// Reads shorter than one Welch segment (n < 1024): SciPy takes nperseg = n (_spectral_py.py _triage_segments) — ONE segment,
// a Hann window of length n, an FFT of length n (any length).  Such reads are tiny (the sweep driver never produces them: it
// reads 0.1 s), so the transform is the plain length-n DFT in float64, one bin per thread and pass, twiddles exp(-2 pi i j / n)
// in LDS indexed by (i k) mod n.  LDS: [xw: 1024 double2][tw: 1024 double2][psd: 1024 f][dbv: 1024 f][red][redi][cpart][cval][plan]
__global__ __launch_bounds__(256) void k_cls_welch_short(const float2 *__restrict__ iq, int n, long n_frames, double fs, RedPlan cp,
                                                         int part_slots, int val_slots, const float *__restrict__ win, float scale,
                                                         const float *__restrict__ mi_in, int32_t *__restrict__ label,
                                                         double *__restrict__ bw_out, float *__restrict__ flat_out,
                                                         float *__restrict__ psd_out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *xw = reinterpret_cast<double2 *>(smem), *tw = xw + CLS_NP;
    float *psd = reinterpret_cast<float *>(tw + CLS_NP), *dbv = psd + CLS_NP;
    double *red = reinterpret_cast<double *>(dbv + CLS_NP);
    int *redi = reinterpret_cast<int *>(red + 16);
    float2 *cpart = reinterpret_cast<float2 *>(redi + 8), *cval = cpart + part_slots;
    {
        int *cur = reinterpret_cast<int *>(cval + val_slots);
        plan_to_lds(cp.full, cur);
        plan_to_lds(cp.tail, cur);
    }
    const int tid = threadIdx.x;
    for (int j = tid; j < n; j += 256) {
        double sn, cs;
        sincospi(-2.0 * (double)j / (double)n, &sn, &cs);
        tw[j] = make_double2(cs, sn);
    }
    __syncthreads();
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *x = iq + (size_t)f * n;
        const float2 m = frame_csum(cp, cpart, cval, [&](int i) { return x[i]; });   // detrend: data - mean(data), complex64
        const float mr = __fdiv_rn(m.x, (float)n), mim = __fdiv_rn(m.y, (float)n);
        for (int i = tid; i < n; i += 256) {
            const float2 v = x[i];
            const float w = win[i];
            xw[i] = make_double2((double)__fmul_rn(w, __fsub_rn(v.x, mr)), (double)__fmul_rn(w, __fsub_rn(v.y, mim)));
        }
        __syncthreads();
        for (int k = tid; k < n; k += 256) {
            double ar = 0.0, ai = 0.0;
            int idx = 0;
            for (int i = 0; i < n; i++) {
                const double2 a = xw[i], t = tw[idx];
                ar += a.x * t.x - a.y * t.y;
                ai += a.x * t.y + a.y * t.x;
                idx += k;
                idx = idx >= n ? idx - n : idx;
            }
            psd[k] = (float)((ar * ar + ai * ai) * (double)scale);      // mean over the one segment
        }
        __syncthreads();
        cls_features(psd, dbv, red, redi, n, f, fs, mi_in, label, bw_out, flat_out, psd_out);
    }
}

// ---- decode_morse, front half (decoders.py:149-165; called with threshold = -20 dB, pyspecsdr.py:573) ---------------------
// envelope = |x| / max|x| (float32); "20 log10(envelope + 1e-10) > -20" holds exactly for envelope + 1e-10 >= 0x3dcccccf under
// NumPy's SVML log10f (probed in the reference environment, see oracle/pss_oracle.c); the rise / fall indices of
// np.diff(signals) are compacted IN ORDER: each wavefront owns a contiguous quarter of the frame, counts its transitions
// (ballot + popcount), the four totals are scanned, then a second sweep writes them.  One workgroup per frame.
// generic: any threshold — the mask is 20 * np.log10(envelope + 1e-10) > float32(threshold) evaluated as NumPy does (float32 throughout, the
// SVML log10 model of pss_npf32.h: bit for bit on every positive float32); else the reference's own -20 dB, folded into one comparison with CUT,
// the smallest float32 whose dB value exceeds -20.
__global__ __launch_bounds__(256) void k_morse_edges(const float2 *__restrict__ iq, int n, long n_frames, int cap,
                                                     int32_t *__restrict__ rise, int32_t *__restrict__ fall,
                                                     int32_t *__restrict__ counts, float thr, int generic)
{
    __shared__ float red[4];
    __shared__ int rnan[4], cr[4], cf[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float CUT = __uint_as_float(0x3dcccccfu);
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *x = iq + (size_t)f * n;
        float mx = -INFINITY;
        int nanv = 0;
        for (int i = tid; i < n; i += 256) {
            const float2 v = x[i];
            const float e = cabsf_np(v.x, v.y);
            nanv |= (e != e);
            mx = e > mx ? e : mx;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(mx, off);
            mx = o > mx ? o : mx;
            nanv |= __shfl_xor(nanv, off);
        }
        if (lane == 0) { red[wave] = mx; rnan[wave] = nanv; }
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (rnan[0] | rnan[1] | rnan[2] | rnan[3]) mx = NAN;  // np.max propagates NaN: then nothing is above the threshold
        auto sig = [&](int i) {
            const float2 v = x[i];
            const float p = __fadd_rn(__fdiv_rn(cabsf_np(v.x, v.y), mx), (float)1e-10);
            return generic ? __fmul_rn(20.0f, log10f_np(p)) > thr : p >= CUT;
        };
        const int nt = n - 1;                                   // transitions i = 0 .. n-2 (between samples i and i+1)
        const int seg = ((nt + 3) / 4 + 63) & ~63;              // per wavefront, a multiple of 64
        const int lo = wave * seg, hi = (lo + seg) < nt ? (lo + seg) : nt;
        int32_t *rf = rise + (size_t)f * cap, *ff = fall + (size_t)f * cap;
        int nr = 0, nf = 0;
        for (int pass = 0; pass < 2; pass++) {
            int br = 0, bf = 0;
            if (pass == 1) {
                for (int w = 0; w < wave; w++) { br += cr[w]; bf += cf[w]; }
            }
            nr = 0; nf = 0;
            for (int b = lo; b < hi; b += 64) {
                const int i = b + lane;
                bool r = false, d = false;
                if (i < hi) {
                    const bool s0 = sig(i), s1 = sig(i + 1);
                    r = !s0 && s1;
                    d = s0 && !s1;
                }
                const unsigned long long mr = __ballot(r), mf = __ballot(d);
                if (pass == 1) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    if (r) { const int o = br + nr + __popcll(mr & below); if (o < cap) rf[o] = i; }
                    if (d) { const int o = bf + nf + __popcll(mf & below); if (o < cap) ff[o] = i; }
                }
                nr += __popcll(mr);
                nf += __popcll(mf);
            }
            if (pass == 0) {
                if (lane == 0) { cr[wave] = nr; cf[wave] = nf; }
                __syncthreads();
            }
        }
        if (tid == 0) { counts[2 * f] = cr[0] + cr[1] + cr[2] + cr[3]; counts[2 * f + 1] = cf[0] + cf[1] + cf[2] + cf[3]; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_afsk_bits(const double *__restrict__ f1, const double *__restrict__ f2, int n, int w,
                                                   int n_bits, long n_rows, uint8_t *__restrict__ bits)
{
    const long total = n_rows * n_bits;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / n_bits;
        const int b = (int)(idx - r * n_bits);
        const double *a1 = f1 + (size_t)r * n + (size_t)b * w, *a2 = f2 + (size_t)r * n + (size_t)b * w;
        double e1 = 0.0, e2 = 0.0;
        for (int st = 0; st < w; st += 8192) {
            const int len = (w - st) < 8192 ? (w - st) : 8192;
            const double c1 = pairwise_sq_f64(a1 + st, len), c2 = pairwise_sq_f64(a2 + st, len);
            e1 = st ? __dadd_rn(e1, c1) : c1;
            e2 = st ? __dadd_rn(e2, c2) : c2;
        }
        bits[idx] = e2 > e1;
    }
}

// samples / np.max(np.abs(samples)) on float64 rows (decode_aprs, decoders.py:126): one workgroup per row; max|x| is exact
// whatever the order, the quotient is IEEE division.  A row of zeros gives 0/0 = NaN, as NumPy does.
__global__ __launch_bounds__(256) void k_row_normalise(const double *__restrict__ x, double *__restrict__ y, int n, long n_rows)
{
    __shared__ double red[4];
    const int tid = threadIdx.x;
    for (long r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const double *a = x + (size_t)r * n;
        double m = 0.0;
        bool nan = false;
        for (int i = tid; i < n; i += 256) {
            const double v = fabs(a[i]);
            nan = nan || v != v;
            m = v > m ? v : m;
        }
        if (nan) m = __builtin_nan("");                      // np.max propagates NaN
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(m, off);
            m = (o != o || o > m) ? o : m;
        }
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        m = red[0];
        for (int w = 1; w < 4; w++) m = (red[w] != red[w] || red[w] > m) ? red[w] : m;
        double *o = y + (size_t)r * n;
        for (int i = tid; i < n; i += 256) o[i] = __ddiv_rn(a[i], m);
    }
}

// adjust_gain (pyspecsdr.py:898-919), sequential by nature.
// adjust_gain (pyspecsdr.py:898-919) over a series of power readings.  A step maps the gain index x to min(x + 1, n_gains - 1), max(x - 1, 0)
// or x, and WHICH of the three depends on the reading alone — so the series is a composition of maps of the form
// f(x) = min(max(x + a, lo), hi), a family closed under composition ((a1, lo1, hi1) then (a2, lo2, hi2) = (a1 + a2, clamp(lo1 + a2, lo2, hi2),
// clamp(hi1 + a2, lo2, hi2))): integer arithmetic, exact, and associative — one workgroup composes per-thread chunks, scans the 1024 chunk maps
// through LDS and replays each chunk from its true starting index (the single-thread loop took 0.9 ms for 8192 readings: one dependent
// global load per step).
struct AgcMap { long long a, lo, hi; };
constexpr long long AGC_INF = 1LL << 60;
__device__ __forceinline__ AgcMap agc_then(const AgcMap f, const AgcMap g)
{
    auto cl = [&](long long v) { return v < g.lo ? g.lo : (v > g.hi ? g.hi : v); };
    return AgcMap{f.a + g.a, cl(f.lo + g.a), cl(f.hi + g.a)};
}
__device__ __forceinline__ int agc_dir(float p)   // +1 / -1 / 0 exactly as the reference's comparisons fall (NaN readings step down)
{
    const float diff = __fsub_rn(-30.0f, p);
    if (fabsf(diff) < 2.0f) return 0;
    return diff > 0 ? 1 : -1;
}
constexpr int AGC_T = 1024;
__global__ __launch_bounds__(AGC_T) void k_agc(const float *__restrict__ power, long n, int idx, int n_gains, int *__restrict__ out)
{
    __shared__ AgcMap sm[2][AGC_T];
    const int t = threadIdx.x;
    const long per = (n + AGC_T - 1) / AGC_T, i0 = (long)t * per, i1 = i0 + per < n ? i0 + per : n;
    const AgcMap up{1, -AGC_INF, (long long)n_gains - 1}, down{-1, 0, AGC_INF};
    AgcMap m{0, -AGC_INF, AGC_INF};
    for (long i = i0; i < i1; i++) {
        const int d = agc_dir(power[i]);
        if (d) m = agc_then(m, d > 0 ? up : down);
    }
    sm[0][t] = m;
    __syncthreads();
    int cur = 0;
    for (int off = 1; off < AGC_T; off <<= 1) {      // inclusive scan of the chunk maps (earlier map first)
        AgcMap v = sm[cur][t];
        if (t >= off) v = agc_then(sm[cur][t - off], v);
        sm[cur ^ 1][t] = v;
        cur ^= 1;
        __syncthreads();
    }
    long long x = idx;
    if (t > 0) {
        const AgcMap pre = sm[cur][t - 1];
        x = x + pre.a;
        x = x < pre.lo ? pre.lo : (x > pre.hi ? pre.hi : x);
    }
    int xi = (int)x;
    for (long i = i0; i < i1; i++) {
        const int d = agc_dir(power[i]);
        if (d > 0) { xi += 1; if (xi > n_gains - 1) xi = n_gains - 1; }
        else if (d < 0) { xi -= 1; if (xi < 0) xi = 0; }
        out[i] = xi;
    }
}

// ---- host helpers --------------------------------------------------------------------------------

void plan_rec(int off, int n, std::vector<int> &lo, std::vector<int> &ll, std::vector<int> &nl, std::vector<int> &nr,
              std::vector<int> &lev, int &slot, int &level)
{
    if (n <= 128) {
        lo.push_back(off);
        ll.push_back(n);
        slot = (int)lo.size() - 1;  // leaf slot (internal nodes are renumbered later)
        level = 0;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    int sl, sr, l1, l2;
    plan_rec(off, n2, lo, ll, nl, nr, lev, sl, l1);
    plan_rec(off + n2, n - n2, lo, ll, nl, nr, lev, sr, l2);
    nl.push_back(sl);
    nr.push_back(sr);
    level = 1 + (l1 > l2 ? l1 : l2);
    lev.push_back(level);
    slot = -(int)nl.size();  // internal node id k encoded as -(k+1)
}

// Plan of ONE GROUP of `len` elements (at most RED_K ufunc chunks): a forest, one pairwise tree per 8192-element chunk,
// plus the list of roots in chunk order (numpy adds the chunk sums sequentially).
// cplx: the tree numpy walks for a complex64 reduce — same recursion over the 2 len interleaved FLOATS, chunks of 8192
// complex elements; leaf offsets/lengths are then in floats (always even).  Stored under key -len.
constexpr int RED_K = 8;
int get_plan(pss_ctx *ctx, int len, PssPairwisePlan **out, bool cplx = false)
{
    const int key = cplx ? -len : len;
    const int n = cplx ? 2 * len : len;
    auto it = ctx->plans.find(key);
    if (it == ctx->plans.end()) {
        std::vector<int> lo, ll, nl, nr, lev, roots;
        int level = 0;
        const int B = cplx ? 16384 : 8192;
        for (int st = 0; st < n; st += B) {
            int s2, l2;
            plan_rec(st, (n - st) < B ? (n - st) : B, lo, ll, nl, nr, lev, s2, l2);
            roots.push_back(s2);
            level = level > l2 ? level : l2;
        }
        const int nleaf = (int)lo.size(), nnode = (int)nl.size();
        // order internal nodes by level; remap ids
        std::vector<int> order(nnode ? nnode : 1), newid(nnode ? nnode : 1), lstart(level + 2, 0);
        int pos = 0;
        for (int lv = 1; lv <= level; lv++) {
            lstart[lv - 1] = pos;
            for (int k = 0; k < nnode; k++)
                if (lev[k] == lv) { order[pos] = k; newid[k] = pos; pos++; }
        }
        lstart[level] = pos;
        auto slot_of = [&](int sl) { return sl >= 0 ? sl : nleaf + newid[-sl - 1]; };
        std::vector<int> L(nnode ? nnode : 1), R(nnode ? nnode : 1);
        for (int k = 0; k < nnode; k++) { L[k] = slot_of(nl[order[k]]); R[k] = slot_of(nr[order[k]]); }
        for (auto &rt : roots) rt = slot_of(rt);
        PssPairwisePlan p;
        p.n_leaves = nleaf; p.n_nodes = nnode; p.n_levels = level; p.n_roots = (int)roots.size();
        {   // one wavefront can fold the tree with shuffles if 64 (leaf, accumulator) pairs cover equal 128-float leaves in order
            // and every node joins two adjacent, equally sized halves (true for 1024 floats and for 1024 complex)
            bool ok = roots.size() == 1 && nleaf * (cplx ? 4 : 8) == 64;
            for (int l = 0; ok && l < nleaf; l++) ok = ll[l] == 128 && lo[l] == 128 * l;
            std::function<int(int, int &)> span = [&](int slot, int &first) -> int {  // leaves under slot, -1 if not perfect
                if (slot < nleaf) { first = slot; return 1; }
                int fl, fr;
                const int a = span(L[slot - nleaf], fl), b = span(R[slot - nleaf], fr);
                if (a < 0 || b < 0 || a != b || fr != fl + a) return -1;
                first = fl;
                return a + b;
            };
            int first = 0;
            if (ok) ok = span(roots[0], first) == nleaf && first == 0;
            p.wave_tree = ok ? 1 : 0;
        }
        PSS_HIP(ctx, hipMalloc(&p.d_leaf_off, sizeof(int) * nleaf));
        PSS_HIP(ctx, hipMalloc(&p.d_leaf_len, sizeof(int) * nleaf));
        PSS_HIP(ctx, hipMalloc(&p.d_node_l, sizeof(int) * L.size()));
        PSS_HIP(ctx, hipMalloc(&p.d_node_r, sizeof(int) * R.size()));
        PSS_HIP(ctx, hipMalloc(&p.d_level_start, sizeof(int) * (level + 1)));
        PSS_HIP(ctx, hipMalloc(&p.d_roots, sizeof(int) * roots.size()));
        PSS_HIP(ctx, hipMemcpy(p.d_leaf_off, lo.data(), sizeof(int) * nleaf, hipMemcpyHostToDevice));
        PSS_HIP(ctx, hipMemcpy(p.d_leaf_len, ll.data(), sizeof(int) * nleaf, hipMemcpyHostToDevice));
        PSS_HIP(ctx, hipMemcpy(p.d_node_l, L.data(), sizeof(int) * L.size(), hipMemcpyHostToDevice));
        PSS_HIP(ctx, hipMemcpy(p.d_node_r, R.data(), sizeof(int) * R.size(), hipMemcpyHostToDevice));
        PSS_HIP(ctx, hipMemcpy(p.d_level_start, lstart.data(), sizeof(int) * (level + 1), hipMemcpyHostToDevice));
        PSS_HIP(ctx, hipMemcpy(p.d_roots, roots.data(), sizeof(int) * roots.size(), hipMemcpyHostToDevice));
        ctx->plans[key] = p;
    }
    *out = &ctx->plans[key];
    return PSS_OK;
}

PlanDev plan_dev(const PssPairwisePlan *p)
{
    if (!p) return PlanDev{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0};
    return PlanDev{p->d_leaf_off, p->d_leaf_len, p->d_node_l, p->d_node_r, p->d_level_start, p->d_roots, p->n_leaves, p->n_levels,
                   p->n_roots, p->n_nodes, p->wave_tree};
}

// Group decomposition of a frame of n elements; slots = the largest (leaves, leaves + nodes + 1) over its group plans.
size_t plan_lds_bytes(const RedPlan &rp)
{
    size_t ints = 0;
    for (const PlanDev *p : {&rp.full, &rp.tail})
        if (p->n_leaves) ints += (size_t)2 * p->n_leaves + 2 * p->n_nodes + (p->n_levels + 1) + p->n_roots;
    return ints * sizeof(int);
}

int get_red_plan(pss_ctx *ctx, int n, bool cplx, RedPlan *rp, int *max_leaves, int *max_vals)
{
    const int glen = RED_K * 8192;
    PssPairwisePlan *full = nullptr, *tail = nullptr;
    const int n_full = n > glen ? n / glen : 0, rem = n - n_full * glen;
    int r;
    if (n_full && (r = get_plan(ctx, glen, &full, cplx))) return r;
    if (rem && (r = get_plan(ctx, rem, &tail, cplx))) return r;
    rp->full = plan_dev(full);
    rp->tail = plan_dev(tail);
    rp->n_full = n_full;
    rp->glen = glen;
    *max_leaves = 0; *max_vals = 0;
    for (const PssPairwisePlan *p : {full, tail})
        if (p) {
            *max_leaves = p->n_leaves > *max_leaves ? p->n_leaves : *max_leaves;
            const int v = p->n_leaves + p->n_nodes + 1;
            *max_vals = v > *max_vals ? v : *max_vals;
        }
    return PSS_OK;
}

template <int KIND>
int launch_pairwise(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_out, float *d_env = nullptr)
{
    RedPlan rp;
    int leaves, vals;
    int r = get_red_plan(ctx, n, false, &rp, &leaves, &vals);
    if (r) return r;
    const int part_slots = 8 * leaves;
    const size_t lds = sizeof(float) * (size_t)(part_slots + vals) + plan_lds_bytes(rp);
    const int lanes = 8 * leaves;  // one lane per (leaf, accumulator) pair
    const int T = lanes <= 64 ? 64 : (lanes <= 128 ? 128 : 256);
    const bool wt = T == 64 && !rp.n_full && rp.tail.wave_tree;  // n = 1024: the whole tree folds inside one wavefront
    auto kern = wt ? k_pairwise<KIND, true> : k_pairwise<KIND, false>;
    if (lds > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
    long g = n_frames < 8192 ? n_frames : 8192;  // several frames per workgroup: the plan tables are copied to LDS once (a grid capped at what the CUs hold at once — 2048 workgroups — measured 20 % slower: the dispatcher's backfill of finished workgroups is the better balance)
    pss_kernel_begin(ctx, "k_pairwise");
    hipLaunchKernelGGL(kern, dim3((int)g), dim3(T), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), n, n_frames, rp,
                       part_slots, vals, d_out, d_env);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_pairwise launch");
}

int launch_pairwise2(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_power, float *d_mean, float *d_env)
{
    RedPlan rp;
    int leaves, vals;
    int r = get_red_plan(ctx, n, false, &rp, &leaves, &vals);
    if (r) return r;
    const int part_slots = 8 * leaves;
    const size_t lds = sizeof(float2) * (size_t)(part_slots + vals) + plan_lds_bytes(rp);
    const int lanes = 8 * leaves;
    const int T = lanes <= 64 ? 64 : (lanes <= 128 ? 128 : 256);
    const bool wt = T == 64 && !rp.n_full && rp.tail.wave_tree;
    auto kern = wt ? k_pairwise2<true> : k_pairwise2<false>;
    if (lds > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long g = n_frames < 8192 ? n_frames : 8192;
    pss_kernel_begin(ctx, "k_pairwise");
    hipLaunchKernelGGL(kern, dim3((int)g), dim3(T), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), n, n_frames, rp, part_slots, vals,
                       d_power, d_mean, d_env);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_pairwise2 launch");
}

int nfm_filters(pss_ctx *ctx, double fs, PssNfmFilt **out)
{
    auto it = ctx->nfm.find(fs);
    if (it == ctx->nfm.end()) {
        PssNfmFilt f;
        int q = (int)(fs / ctx->target_rate);
        if (q < 1) return pss_fail(ctx, PSS_E_ARG, "sample rate below the target rate: decimation factor int(fs / target_rate) is 0");
        int r = pss_design_firwin(65, 15000.0 / (fs / 2.0), f.taps);
        if (r) return pss_fail(ctx, r, "firwin: invalid cutoff frequency (15 kHz must be below fs/2)");
        r = pss_design_cheby1_sos(8, 0.05, 0.8 / q, f.sos);
        if (r) return pss_fail(ctx, r, "cheby1 design failed");
        pss_design_sosfilt_zi(f.sos, 4, f.zi);
        ctx->nfm[fs] = f;
    }
    *out = &ctx->nfm[fs];
    return PSS_OK;
}

// the reversed taps in device memory (fir_ring_asm's scalar loads); uploaded once per coefficient set
int nfm_dev_taps(pss_ctx *ctx, PssNfmFilt *f, const double **out)
{
    if (!f->d_rev) {
        double rev[80] = {0.0};   // 65 taps + 15 zero slots (the left-edge dots read up to 14 elements past their last tap)
        for (int j = 0; j < 65; j++) rev[j] = f->taps[64 - j];
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&f->d_rev), sizeof(rev));
        if (e == hipSuccess) e = hipMemcpy(f->d_rev, rev, sizeof(rev), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (f->d_rev) { hipFree(f->d_rev); f->d_rev = nullptr; }
            return pss_fail(ctx, PSS_E_HIP, std::string("device taps: ") + hipGetErrorString(e));
        }
    }
    *out = f->d_rev;
    return PSS_OK;
}

int wfm_filters(pss_ctx *ctx, double fs, PssWfmFilt **out)
{
    auto it = ctx->wfm.find(fs);
    if (it == ctx->wfm.end()) {
        PssWfmFilt f;
        const double nyq = fs / 2.0;
        int r = pss_design_butter_sos(5, 0.0, 15000.0 / nyq, f.lp, nullptr);
        if (!r) r = pss_design_butter_sos(5, (19000.0 - 200.0) / nyq, (19000.0 + 200.0) / nyq, f.pilot, nullptr);
        if (!r) r = pss_design_butter_sos(5, (38000.0 - 15000.0) / nyq, (38000.0 + 15000.0) / nyq, f.lmr, nullptr);
        // scipy: "Digital filter critical frequencies must be 0 < Wn < 1" (53 kHz must be below fs/2)
        if (r) return pss_fail(ctx, r, "butter: digital filter critical frequencies must be 0 < Wn < 1 (WFM needs fs > 106 kHz)");
        f.alpha = pss_np_exp(-1.0 / (75e-6 * fs));   // np.exp, not libm's (signal_processing.py:145)
        ctx->wfm[fs] = f;
    }
    *out = &ctx->wfm[fs];
    return PSS_OK;
}

int ssb_taps(pss_ctx *ctx, double fs, double **out)
{
    auto it = ctx->ssb.find(fs);
    if (it == ctx->ssb.end()) {
        std::array<double, 65> t;
        int r = pss_design_firwin(65, 3000.0 / fs, t.data());
        if (r) return pss_fail(ctx, r, "firwin: invalid cutoff frequency (3000/fs must be in (0,1))");
        ctx->ssb[fs] = t;
    }
    *out = ctx->ssb[fs].data();
    return PSS_OK;
}

TapsArg make_taps(const double *taps)
{
    TapsArg t;
    for (int j = 0; j < 65; j++) { t.fwd[j] = taps[j]; t.rev[j] = taps[64 - j]; }
    for (int k = 0; k < 4; k++)
        for (int h = 0; h < 2; h++) {
            double *r = t.ring + (2 * k + h) * 8;
            const int l = 2 * h;
            r[0] = t.rev[8 * k + l]; r[1] = t.rev[8 * k + l + 1]; r[2] = t.rev[8 * k + l + 4]; r[3] = t.rev[8 * k + l + 5];
            r[4] = t.rev[32 + 8 * k + l]; r[5] = t.rev[32 + 8 * k + l + 1]; r[6] = t.rev[32 + 8 * k + l + 4]; r[7] = t.rev[32 + 8 * k + l + 5];
        }
    return t;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int pss_demod_out_len_rate(int mode, int n, double fs, double target_rate)
{
    if (n <= 0) return 0;
    if (mode == PSS_MODE_NFM || mode == PSS_MODE_WFM) {
        if (!(target_rate > 0.0)) return PSS_E_ARG;
        int q = (int)(fs / target_rate);
        if (q < 1) return PSS_E_ARG;
        return (n - 1 + q - 1) / q;
    }
    if (mode == PSS_MODE_AM || mode == PSS_MODE_USB || mode == PSS_MODE_LSB) return n;
    return PSS_E_ARG;
}
extern "C" int pss_demod_out_len(int mode, int n, double fs) { return pss_demod_out_len_rate(mode, n, fs, 22050.0); }
extern "C" int pss_demod_out_len_ctx(pss_ctx *ctx, int mode, int n, double fs)
{
    return ctx ? pss_demod_out_len_rate(mode, n, fs, ctx->target_rate) : PSS_E_ARG;
}

// demodulate_nfm / demodulate_wfm's target_rate argument (signal_processing.py:91, :119; default 22050 = DEFAULT_SAMPLE_RATE): the decimation
// factor is int(sample_rate / target_rate).  Changing it drops the cached decimator designs (cheby1(8, 0.05, 0.8 / q) + zi per sample rate).
extern "C" int pss_set_target_rate(pss_ctx *ctx, double target_rate)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!(target_rate > 0.0)) return pss_fail(ctx, PSS_E_ARG, "pss_set_target_rate: the target rate must be positive");
    if (target_rate == ctx->target_rate) return PSS_OK;
    hipStreamSynchronize(ctx->stream);      // queued launches may still read the device copies of the taps
    for (auto &kv : ctx->nfm)
        if (kv.second.d_rev) hipFree(kv.second.d_rev);
    ctx->nfm.clear();
    ctx->target_rate = target_rate;
    return PSS_OK;
}

static int iq_correction_launch(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_out_iq, float *d_raw, float *d_scal);
extern "C" int pss_iq_correction(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_out_iq, float *d_raw)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || n <= 0 || (n_frames > 0 && (!d_iq || (!d_out_iq && !d_raw)))) return pss_fail(ctx, PSS_E_ARG, "pss_iq_correction: bad argument");
    if (n_frames == 0) return PSS_OK;
    return iq_correction_launch(ctx, d_iq, n_frames, n, d_out_iq, d_raw, nullptr);
}
// d_scal (nullable): float32 [n_frames][8], the frames' five correction scalars (IqcScal); d_out_iq / d_raw may then both be NULL
static int iq_correction_launch(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_out_iq, float *d_raw, float *d_scal)
{
    RedPlan rp, cp;
    int rl, rv, cl, cv;
    int r = get_red_plan(ctx, n, false, &rp, &rl, &rv);
    if (r) return r;
    r = get_red_plan(ctx, n, true, &cp, &cl, &cv);
    if (r) return r;
    // LDS: partial sums (8 floats | 4 float2 per leaf) + tree values, optionally the frame itself
    size_t part_slots = (size_t)(rl * 4 > cl * 4 ? rl * 4 : cl * 4);  // in float2
    size_t val_slots = (size_t)(rv > cv ? rv : cv);
    // staging pays while >= 2 workgroups fit a CU (measured: 0.59 vs 0.91 ms at 65536 x 1024, but 2.6 vs 1.8 ms at 8192 x 16384)
    const bool staged = n <= 8192;
    size_t lds = (part_slots + val_slots + (staged ? (size_t)n : 0)) * sizeof(float2) + plan_lds_bytes(rp) + plan_lds_bytes(cp);
    const int lanes = rl * 8 > cl * 4 ? rl * 8 : cl * 4;
    const int T = lanes <= 64 ? 64 : (lanes <= 128 ? 128 : 256);  // one lane per (leaf, accumulator) pair
    const bool wt = T == 64 && staged && !rp.n_full && !cp.n_full && rp.tail.wave_tree && cp.tail.wave_tree;  // n = 1024
    auto kern = wt ? k_iqcorr<true, true> : (staged ? k_iqcorr<true, false> : k_iqcorr<false, false>);
    if (lds > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const RedPlan &a = rp, &b = cp;
    long g = n_frames < 8192 ? n_frames : 8192;  // several frames per workgroup: the plan tables are copied to LDS once (a grid capped at what the CUs hold at once — 2048 workgroups — measured 20 % slower: the dispatcher's backfill of finished workgroups is the better balance)
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_iqcorr");
    hipLaunchKernelGGL(kern, dim3((int)g), dim3(T), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), n, n_frames, a, b,
                       (int)part_slots, (int)val_slots, reinterpret_cast<float2 *>(d_out_iq), d_raw, d_scal);
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_iqcorr launch");
}

extern "C" int pss_power_db(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_power)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n < 1 || n_frames < 0 || (n_frames > 0 && (!d_iq || !d_power))) return pss_fail(ctx, PSS_E_ARG, "bad power arguments");
    if (n_frames == 0) return PSS_OK;
    pss_time_begin(ctx);
    int r = launch_pairwise<0>(ctx, d_iq, n_frames, n, d_power);
    pss_time_end(ctx);
    return r;
}

extern "C" int pss_agc_steps(pss_ctx *ctx, const float *d_power, long n, int start_idx, int n_gains, int32_t *d_idx_out)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_power || !d_idx_out || n < 0 || n_gains < 1) return pss_fail(ctx, PSS_E_ARG, "bad agc arguments");
    if (n == 0) return PSS_OK;
    pss_kernel_begin(ctx, "k_agc");
    hipLaunchKernelGGL(k_agc, dim3(1), dim3(AGC_T), 0, PSS_STREAM(ctx), d_power, n, start_idx, n_gains, d_idx_out);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_agc launch");
}

extern "C" int pss_demod(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, int16_t *d_pcm,
                         double *d_audio)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || n < 1 || (n_frames > 0 && !d_iq)) return pss_fail(ctx, PSS_E_ARG, "bad demod arguments");
    if (n_frames > 0 && !d_pcm && !d_audio) return pss_fail(ctx, PSS_E_ARG, "both outputs are null");
    const long tiles = (n_frames + TILE - 1) / TILE;
    if (mode == PSS_MODE_NFM) {
        if (n - 1 <= EDGE)
            return pss_fail(ctx, PSS_E_PADLEN, "The length of the input vector x must be greater than padlen, which is 27.");
        PssNfmFilt *flt;
        int r = nfm_filters(ctx, fs, &flt);
        if (r) return r;
        if (n_frames == 0) return PSS_OK;
        const int q = (int)(fs / ctx->target_rate);
        const int n_out = (n - 1 + q - 1) / q;
        const long L = (long)(n - 1) + 2 * EDGE;
        const long Lp = (L + 1) & ~1L;  // even row stride -> 16-byte aligned rows of u
        const size_t szU = align256((size_t)n_frames * Lp * sizeof(double));
        const size_t szY = align256((size_t)tiles * L * TILE * sizeof(double));
        const size_t szA = align256((size_t)tiles * n_out * TILE * sizeof(double));
        double *U = nullptr, *Y = nullptr, *A = nullptr;  // three-kernel path only; each path sizes the (grow-only) scratch itself
        const TapsArg targ = make_taps(flt->taps);
        NfmCoef c;
        for (int s = 0; s < 4; s++) {
            const double *row = flt->sos + 6 * s;
            c.s[s] = Biquad{row[0], row[1], row[2], row[4], row[5]};
        }
        for (int i = 0; i < 8; i++) c.zi[i] = flt->zi[i];
        const float kscale = (float)(fs / (2.0 * M_PI));          // python float -> float32 scalar (:97)
        const int swapped = ((long)(n - 1) * 8 >= 262144) ? 1 : 0;  // NumPy temporary elision threshold
        bool b121 = true;  // sections 1..3 with numerator exactly [1, 2, 1] (always so for cheby1 low-pass SOS)
        for (int s = 1; s < 4; s++)
            b121 = b121 && flt->sos[6 * s] == 1.0 && flt->sos[6 * s + 1] == 2.0 && flt->sos[6 * s + 2] == 1.0;
        // a handful of long frames (the interactive loop: one 32768-sample buffer per call) cannot fill lane-per-frame
        // wavefronts: below this many frames the decimator runs as a 4-lane systolic array per frame instead
        const bool small_batch = !ctx->no_small_batch && n_frames <= ctx->small_batch_max;
        // (the fused kernels exist for decimator sections 1..3 with numerator exactly [1, 2, 1] — every cheby1 low-pass SOS; injected
        // tables of another shape take the three-kernel path)
        // (and for tiles of less than 2 GiB of IQ: the forward kernel addresses a tile of 64 frames through one buffer resource with 32-bit
        // offsets — frames of 4 Mi samples and more take the three-kernel path)
        if (n - 1 >= 128 && !ctx->no_fused && !small_batch && b121 && (long)TILE * n * (long)sizeof(float2) < (1L << 31)) {
            // fused path: u[] stays on chip; small L2-resident scratch for the irregular head / tail of u
            const size_t szH = align256((size_t)tiles * fused::HEAD * TILE * sizeof(double));
            const size_t szT = align256((size_t)tiles * (EDGE + 1) * TILE * sizeof(double));
            r = pss_ensure_scratch(ctx, szY + szA + szH + szT);
            if (r) return r;
            const double *d_rev = nullptr;
            r = nfm_dev_taps(ctx, flt, &d_rev);
            if (r) return r;
            const unsigned ncu = ctx->n_cus > 0 ? (unsigned)ctx->n_cus : 256u;
            if (!ctx->prog) {    // progress words of the forward kernel's workgroups (16 bytes per CU: up to four workgroups share one)
                PSS_HIP(ctx, hipMalloc(&ctx->prog, 16 * (size_t)ncu));
                PSS_HIP(ctx, hipMemsetAsync(ctx->prog, 0, 16 * (size_t)ncu, PSS_STREAM(ctx)));   // ordered before the first launch
            }
            ctx->prog_epoch = (ctx->prog_epoch + 1u) & 0xffffu;   // every launch ranks only against words of its own epoch (never reset)
            if (ctx->prog_epoch == 0) ctx->prog_epoch = 1;
            char *base = reinterpret_cast<char *>(ctx->scratch);
            double *Yf = reinterpret_cast<double *>(base), *Af = reinterpret_cast<double *>(base + szY);
            double *Uh = reinterpret_cast<double *>(base + szY + szA), *Ut = reinterpret_cast<double *>(base + szY + szA + szH);
            pss_time_begin(ctx);
            pss_kernel_begin(ctx, "k_nfm_fwd");
            {
                auto kf = swapped ? fused::k_nfm_fwd<true, true> : fused::k_nfm_fwd<true, false>;
                hipLaunchKernelGGL(kf, dim3((unsigned)tiles), dim3(fused::WG), fused::LDS_BYTES, PSS_STREAM(ctx),
                                   reinterpret_cast<const float2 *>(d_iq), Yf, Uh, Ut, n, n_frames, c, kscale, d_rev,
                                   reinterpret_cast<unsigned *>(ctx->prog), ncu, ctx->prog_epoch, 0L);
            }
            pss_kernel_end(ctx);
            auto launch_bwd = [=]() -> int {
                pss_kernel_begin(ctx, "k_nfm_bwd");
                hipLaunchKernelGGL(fused::k_nfm_bwd<true>, dim3((unsigned)tiles), dim3(TILE), 0, PSS_STREAM(ctx), Yf, Af, n, q,
                                   n_out, n_frames, c, d_pcm, d_audio);
                pss_kernel_end(ctx);
                return pss_hip_check(ctx, hipGetLastError(), "k_nfm_bwd launch");
            };
            if (ctx->defer_bwd) {  // pss_frame_pipeline_nfm places the backward pass itself (beside the post-process)
                ctx->pending_bwd = launch_bwd;
                pss_time_end(ctx);
                return pss_hip_check(ctx, hipGetLastError(), "nfm fused launch");
            }
            if (ctx->fork_after_fwd) {  // pss_spectrum_nfm overlaps the rest
                PSS_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
                ctx->did_fork = true;
            }
            r = launch_bwd();
            pss_time_end(ctx);
            if (r) return r;
            return pss_hip_check(ctx, hipGetLastError(), "nfm fused launch");
        }
        const int cpf = (n - 1 + FIR_CH - 1) / FIR_CH;
        const long items = n_frames * cpf;
        const long g1 = items < 256L * 64 ? items : 256L * 64;
        if (!small_batch) {
            r = pss_ensure_scratch(ctx, szU + szY + szA);
            if (r) return r;
            U = reinterpret_cast<double *>(ctx->scratch);
            Y = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->scratch) + szU);
            A = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->scratch) + szU + szY);
        }
        if (small_batch) {
            const size_t szY2 = align256((size_t)n_frames * L * sizeof(double));
            const size_t szA2 = align256((size_t)n_frames * n_out * sizeof(double));
            r = pss_ensure_scratch(ctx, szU + szY2 + szA2 + align256((size_t)n_frames * sizeof(double)));
            if (r) return r;
            U = reinterpret_cast<double *>(ctx->scratch);
        }
        pss_time_begin(ctx);
        pss_kernel_begin(ctx, "k_nfm_front");
        hipLaunchKernelGGL(k_nfm_front, dim3((unsigned)g1), dim3(FIR_T), 0, PSS_STREAM(ctx),
                           reinterpret_cast<const float2 *>(d_iq), U, n, n_frames, cpf, Lp, kscale, swapped, targ);
        pss_kernel_end(ctx);
        pss_kernel_begin(ctx, "k_nfm_edge");
        hipLaunchKernelGGL(k_nfm_edge, dim3((unsigned)n_frames), dim3(128), 0, PSS_STREAM(ctx),
                           reinterpret_cast<const float2 *>(d_iq), U, n, n_frames, Lp, kscale, swapped, targ);
        pss_kernel_end(ctx);
        if (small_batch) {
            // few frames: the four sections of a frame on four lanes (16 frames per wavefront), one launch per direction
            char *b2 = reinterpret_cast<char *>(ctx->scratch);
            const size_t szY2 = align256((size_t)n_frames * L * sizeof(double));
            const size_t szA2 = align256((size_t)n_frames * n_out * sizeof(double));
            double *Y2 = reinterpret_cast<double *>(b2 + szU), *A2 = reinterpret_cast<double *>(b2 + szU + szY2);
            double *MX = reinterpret_cast<double *>(b2 + szU + szY2 + szA2);
            const unsigned gs = (unsigned)((n_frames + IS_G - 1) / IS_G);
            pss_kernel_begin(ctx, "k_iir4_sys");
            hipLaunchKernelGGL(k_iir4_sys, dim3(gs), dim3(128), 0, PSS_STREAM(ctx), U, Lp, 0, L, L, c, Y2, L, q, n_out, nullptr, n_frames);
            pss_kernel_end(ctx);
            pss_kernel_begin(ctx, "k_iir4_sys");
            hipLaunchKernelGGL(k_iir4_sys, dim3(gs), dim3(128), 0, PSS_STREAM(ctx), Y2, L, 1, L, L - EDGE, c, A2, (long)n_out, q, n_out, MX,
                               n_frames);
            pss_kernel_end(ctx);
            size_t tot = (size_t)n_frames * n_out;
            size_t g2 = (tot + TPB - 1) / TPB;
            if (g2 > 16384) g2 = 16384;
            pss_kernel_begin(ctx, "k_finalize");
            hipLaunchKernelGGL(k_finalize, dim3((unsigned)g2), dim3(TPB), 0, PSS_STREAM(ctx), A2, MX, n_out, n_frames, d_pcm, d_audio);
            pss_kernel_end(ctx);
            pss_time_end(ctx);
            return pss_hip_check(ctx, hipGetLastError(), "nfm small-batch launch");
        }
        pss_kernel_begin(ctx, "k_nfm_iir");
        if (b121)
            hipLaunchKernelGGL(k_nfm_iir<true>, dim3((unsigned)tiles), dim3(TILE), 0, PSS_STREAM(ctx), U, Y, A, n, q, n_out,
                               n_frames, Lp, c, d_pcm, d_audio);
        else
            hipLaunchKernelGGL(k_nfm_iir<false>, dim3((unsigned)tiles), dim3(TILE), 0, PSS_STREAM(ctx), U, Y, A, n, q, n_out,
                               n_frames, Lp, c, d_pcm, d_audio);
        pss_kernel_end(ctx);
        pss_time_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "nfm launch");
    }
    if (mode == PSS_MODE_AM) {
        if (n_frames == 0) return PSS_OK;
        const size_t szY = align256((size_t)n_frames * n * sizeof(double));
        const size_t szM = align256((size_t)n_frames * sizeof(double));
        const size_t szMu = align256((size_t)n_frames * sizeof(float));
        const size_t szE = align256((size_t)n_frames * n * sizeof(float));
        int r = pss_ensure_scratch(ctx, szY + szM + szMu + szE);
        if (r) return r;
        char *base = reinterpret_cast<char *>(ctx->scratch);
        double *Yf = reinterpret_cast<double *>(base);
        double *mx = reinterpret_cast<double *>(base + szY);
        float *mu = reinterpret_cast<float *>(base + szY + szM);
        float *env = reinterpret_cast<float *>(base + szY + szM + szMu);  // |samples| float32, written once by the mean pass
        double sos[30];
        pss_am_bandpass_sos(sos);
        AmCoef c;
        for (int s = 0; s < 5; s++) c.s[s] = Biquad{sos[6 * s], sos[6 * s + 1], sos[6 * s + 2], sos[6 * s + 4], sos[6 * s + 5]};
        pss_time_begin(ctx);
        // (pss_demod_power: measure_signal_power of the same frames comes out of the same pass over the IQ)
        r = ctx->power_out ? launch_pairwise2(ctx, d_iq, n_frames, n, ctx->power_out, mu, env) : launch_pairwise<1>(ctx, d_iq, n_frames, n, mu, env);
        ctx->power_out = nullptr;
        if (r) { pss_time_end(ctx); return r; }
        pss_kernel_begin(ctx, "k_am_grp");
        hipLaunchKernelGGL(k_am_grp, dim3((unsigned)((n_frames + GRP_G - 1) / GRP_G)), dim3(256), 0, PSS_STREAM(ctx), env, mu, Yf, mx, n, n_frames, c);
        pss_kernel_end(ctx);
        size_t total = (size_t)n_frames * n;
        size_t g = (total + TPB - 1) / TPB;
        if (g > 16384) g = 16384;
        pss_kernel_begin(ctx, "k_finalize");
        hipLaunchKernelGGL(k_finalize, dim3((unsigned)g), dim3(TPB), 0, PSS_STREAM(ctx), Yf, mx, n, n_frames, d_pcm, d_audio);
        pss_kernel_end(ctx);
        pss_time_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "am launch");
    }
    if (mode == PSS_MODE_USB || mode == PSS_MODE_LSB) {
        double *taps;
        int r = ssb_taps(ctx, fs, &taps);
        if (r) return r;
        if (n_frames == 0) return PSS_OK;
        const size_t szY = align256((size_t)n_frames * n * sizeof(double));
        const size_t szM = align256((size_t)n_frames * sizeof(double));
        r = pss_ensure_scratch(ctx, szY + szM);
        if (r) return r;
        char *base = reinterpret_cast<char *>(ctx->scratch);
        double *Yf = reinterpret_cast<double *>(base);
        unsigned long long *mxb = reinterpret_cast<unsigned long long *>(base + szY);
        const TapsArg targ = make_taps(taps);
        pss_time_begin(ctx);
        PSS_HIP(ctx, hipMemsetAsync(mxb, 0, (size_t)n_frames * sizeof(double), PSS_STREAM(ctx)));
        if (ctx->ssb_hilbert && pss_ssb_fused_supported(n) && !ctx->ssb_unfused && !ctx->hilbert_exact && !ctx->iq_c128) {
            // frames of 8192 / 16 384 samples: FIR, hilbert() round trip, normalisation and PCM in ONE kernel (no float64 round trip of
            // the FIR output through HBM)
            r = pss_ssb_hilbert_fused(ctx, d_iq, n_frames, n, taps, d_audio, d_pcm);
            pss_time_end(ctx);
            return r ? r : pss_hip_check(ctx, hipGetLastError(), "ssb launch");
        }
        const int cpf = (n + 1023) / 1024;
        long total = n_frames * cpf;
        long g = total < 16384 ? total : 16384;
        // (iq_c128: d_iq points at complex128 frames — pss_demod_ssb_c128; the same kernels behind a float64 loader)
        pss_kernel_begin(ctx, "k_ssb_fir");
        if (ctx->iq_c128)
            hipLaunchKernelGGL(k_ssb_fir<double2>, dim3((unsigned)g), dim3(TPB), 0, PSS_STREAM(ctx), reinterpret_cast<const double2 *>(d_iq),
                               Yf, mxb, n, n_frames, cpf, targ);
        else
            hipLaunchKernelGGL(k_ssb_fir<float2>, dim3((unsigned)g), dim3(TPB), 0, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq),
                               Yf, mxb, n, n_frames, cpf, targ);
        pss_kernel_end(ctx);
        pss_kernel_begin(ctx, "k_ssb_edge");
        if (ctx->iq_c128)
            hipLaunchKernelGGL(k_ssb_edge<double2>, dim3((unsigned)n_frames), dim3(128), 0, PSS_STREAM(ctx),
                               reinterpret_cast<const double2 *>(d_iq), Yf, mxb, n, n_frames, targ);
        else
            hipLaunchKernelGGL(k_ssb_edge<float2>, dim3((unsigned)n_frames), dim3(128), 0, PSS_STREAM(ctx),
                               reinterpret_cast<const float2 *>(d_iq), Yf, mxb, n, n_frames, targ);
        pss_kernel_end(ctx);
        // hilbert(np.real(analytical)) and np.real of it again (signal_processing.py:205-213): the FFT round trip of the
        // reference, executed where a register transform exists for the frame length; numerically the identity on the
        // real part up to the transforms' rounding (~1e-16), so skipping it (option "ssb_hilbert" = 0, and every other
        // frame length) changes no int16 sample
        if (ctx->ssb_hilbert && pss_hilbert_supported(n) && n <= 16384) {
            // ... with the normalisation and the int16 conversion in the same kernel (the frame is in registers when the frame
            // peak becomes known): no float64 round trip through HBM, no k_finalize pass
            r = pss_hilbert_rows(ctx, Yf, n_frames, n, d_audio, 2, nullptr, d_pcm);
            pss_time_end(ctx);
            return r ? r : pss_hip_check(ctx, hipGetLastError(), "ssb launch");
        }
        if (ctx->ssb_hilbert && pss_hilbert_supported(n)) {
            // longer read buffers (the reference's default is 32768 samples): the round trip goes through a spectrum in HBM, in
            // place on Yf, and leaves the frame peak of its real part for k_finalize
            PSS_HIP(ctx, hipMemsetAsync(mxb, 0, (size_t)n_frames * sizeof(double), PSS_STREAM(ctx)));
            r = pss_hilbert_rows(ctx, Yf, n_frames, n, Yf, 1, mxb, nullptr);
            if (r) { pss_time_end(ctx); return r; }
        }
        size_t tot = (size_t)n_frames * n;
        size_t g2 = (tot + TPB - 1) / TPB;
        if (g2 > 16384) g2 = 16384;
        pss_kernel_begin(ctx, "k_finalize");
        hipLaunchKernelGGL(k_finalize, dim3((unsigned)g2), dim3(TPB), 0, PSS_STREAM(ctx), Yf,
                           reinterpret_cast<const double *>(mxb), n, n_frames, d_pcm, d_audio);
        pss_kernel_end(ctx);
        pss_time_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "ssb launch");
    }
    if (mode == PSS_MODE_WFM) {
        const int q = (int)(fs / ctx->target_rate);
        if (q < 1) return pss_fail(ctx, PSS_E_ARG, "WFM: sample rate below the target rate");
        const bool q1 = q == 1;     // the reference skips its decimate() stage (:152-155): the plain kernels, normalisation of the de-emphasised rows
        if (n - 1 <= EDGE)
            return pss_fail(ctx, PSS_E_PADLEN, "The length of the input vector x must be greater than padlen, which is 27.");
        PssWfmFilt *wf;
        int r = wfm_filters(ctx, fs, &wf);
        if (r) return r;
        PssNfmFilt *flt;  // the decimator (cheby1 + zi) is the NFM one
        r = nfm_filters(ctx, fs, &flt);
        if (r) return r;
        if (n_frames == 0) return PSS_OK;
        const long rows = 2 * n_frames, tiles2 = (rows + TILE - 1) / TILE;
        const int n_out = (n - 1 + q - 1) / q;
        const long L = (long)(n - 1) + 2 * EDGE;
        const long Lp = (L + 1) & ~1L;
        const size_t szU = align256((size_t)rows * Lp * sizeof(double));
        const long T2 = 2 * tiles;  // the fused path keeps the channels of a tile in two separate row blocks (T2 >= tiles2)
        const size_t szY = align256((size_t)T2 * L * TILE * sizeof(double));
        const size_t szA = align256((size_t)T2 * n_out * TILE * sizeof(double));
        const size_t szM = align256((size_t)T2 * TILE * sizeof(double));
        bool b121_dec = true;  // decimator sections 1..3 with numerator exactly [1, 2, 1]: the fused kernels' shape
        for (int s1 = 1; s1 < 4; s1++)
            b121_dec = b121_dec && flt->sos[6 * s1] == 1.0 && flt->sos[6 * s1 + 1] == 2.0 && flt->sos[6 * s1 + 2] == 1.0;
        r = pss_ensure_scratch(ctx, ((ctx->no_wfm_fused || !b121_dec || q1) ? szU : 0) + szY + szA + szM);
        if (r) return r;
        char *base = reinterpret_cast<char *>(ctx->scratch);
        double *U = reinterpret_cast<double *>(base), *Y = reinterpret_cast<double *>(base + szU);
        double *A = reinterpret_cast<double *>(base + szU + szY), *MX = reinterpret_cast<double *>(base + szU + szY + szA);
        WfmCoef wc;
        auto fill = [](Biquad *dst, const double *sos, int ns) {
            for (int s2 = 0; s2 < ns; s2++) dst[s2] = Biquad{sos[6 * s2], sos[6 * s2 + 1], sos[6 * s2 + 2], sos[6 * s2 + 4], sos[6 * s2 + 5]};
        };
        fill(wc.lp, wf->lp, 3); fill(wc.pil, wf->pilot, 5); fill(wc.lmr, wf->lmr, 5);
        wc.b0d = 1.0 - wf->alpha;
        wc.a1d = -wf->alpha;
        NfmCoef c;
        fill(c.s, flt->sos, 4);
        for (int i = 0; i < 8; i++) c.zi[i] = flt->zi[i];
        bool b121 = true;
        for (int s2 = 1; s2 < 4; s2++)
            b121 = b121 && flt->sos[6 * s2] == 1.0 && flt->sos[6 * s2 + 1] == 2.0 && flt->sos[6 * s2 + 2] == 1.0;
        const int swapped = ((long)(n - 1) * 8 >= 262144) ? 1 : 0;
        // SciPy's zero pairing gives every Butterworth SOS the same numerator shapes; anything else takes the generic steps
        auto is_num = [](const double *row, double b0, double b1, double b2) { return row[0] == b0 && row[1] == b1 && row[2] == b2; };
        bool spec = is_num(wf->lp + 6, 1, 2, 1) && is_num(wf->lp + 12, 1, 1, 0);
        for (const double *bp : {wf->pilot, wf->lmr})
            spec = spec && is_num(bp + 6, 1, 2, 1) && is_num(bp + 12, 1, 0, -1) && is_num(bp + 18, 1, -2, 1) && is_num(bp + 24, 1, -2, 1);
        pss_time_begin(ctx);
        // demodulate_signal's dispatcher (pss_demod_signal, pss_frame_pipeline): d_iq holds the frames as read and iq_correction comes first
        // (signal_processing.py:222-225).  The fused forward kernel corrects the samples as it loads them, from a pre-pass that leaves five
        // scalars per frame (no corrected copy of the batch: 8 bytes per sample neither written nor read back); the other kernel families
        // read a corrected copy from the context's scratch.
        const bool correct = ctx->wfm_correct;
        ctx->wfm_correct = false;
        ctx->wfm_scal = nullptr;
        if (correct) {
            const bool fused_path = !(!ctx->no_small_batch && n_frames <= ctx->wfm_small_batch_max) && !ctx->no_wfm_fused && b121 && !ctx->wfm_corr_copy && !q1;
            if (fused_path) {
                r = pss_ensure_buffer(ctx, &ctx->scratch_iqc, &ctx->scratch_iqc_bytes, (size_t)n_frames * 8 * sizeof(float), "iq_correction scalars");
                if (!r) r = iq_correction_launch(ctx, d_iq, n_frames, n, nullptr, nullptr, reinterpret_cast<float *>(ctx->scratch_iqc));
                ctx->wfm_scal = reinterpret_cast<const float *>(ctx->scratch_iqc);
            } else {
                r = pss_ensure_buffer(ctx, &ctx->scratch_iqc, &ctx->scratch_iqc_bytes, (size_t)n_frames * n * sizeof(float2), "iq_correction scratch");
                if (!r) r = iq_correction_launch(ctx, d_iq, n_frames, n, reinterpret_cast<float *>(ctx->scratch_iqc), nullptr, nullptr);
                d_iq = reinterpret_cast<const float *>(ctx->scratch_iqc);
            }
            if (r) { pss_time_end(ctx); return r; }
        }
        if (!q1 && !ctx->no_small_batch && n_frames <= ctx->wfm_small_batch_max) {
            // a handful of frames: one lane per filter SECTION instead of one lane per frame (k_wfm_casc, k_iir4_sys)
            const int M = n - 1;
            const size_t szR = align256((size_t)n_frames * M * sizeof(double));
            const size_t szY2 = align256((size_t)rows * L * sizeof(double));
            const size_t szA2 = align256((size_t)rows * n_out * sizeof(double));
            r = pss_ensure_scratch(ctx, 3 * szR + szU + szY2 + szA2 + align256((size_t)rows * sizeof(double)));
            if (r) return r;
            char *b2 = reinterpret_cast<char *>(ctx->scratch);
            double *Aa = reinterpret_cast<double *>(b2), *Pp = reinterpret_cast<double *>(b2 + szR), *Mm = reinterpret_cast<double *>(b2 + 2 * szR);
            double *U2 = reinterpret_cast<double *>(b2 + 3 * szR), *Y2 = reinterpret_cast<double *>(b2 + 3 * szR + szU);
            double *A2 = reinterpret_cast<double *>(b2 + 3 * szR + szU + szY2), *MX2 = reinterpret_cast<double *>(b2 + 3 * szR + szU + szY2 + szA2);
#ifdef PSS_EXP_CASC_SAMPLE
            CascArg a1, a2;
            const Biquad idle{0.0, 0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < 16; i++) a1.lane[i] = a2.lane[i] = CascLane{idle, 1, 0, 0, -1};
            for (int i = 0; i < 3; i++) a1.lane[i] = CascLane{wc.lp[i], i == 0, 0, 0, i == 2 ? 0 : -1};             // a
            for (int i = 0; i < 5; i++) a1.lane[3 + i] = CascLane{wc.pil[i], i == 0, 0, 0, -1};                       // pilot band-pass
            a1.lane[8] = CascLane{Biquad{1.0, 0.0, 0.0, -0.99, 0.0}, 0, 1, 0, 1};                                     // lfilter([1],[1,-0.99]) -> y
            for (int i = 0; i < 5; i++) a1.lane[9 + i] = CascLane{wc.lmr[i], i == 0, 0, 0, i == 4 ? 2 : -1};          // m
            for (int ch = 0; ch < 2; ch++) {
                for (int i = 0; i < 3; i++) a2.lane[4 * ch + i] = CascLane{wc.lp[i], i == 0, 0, 0, -1};               // :137
                a2.lane[4 * ch + 3] = CascLane{Biquad{wc.b0d, 0.0, 0.0, wc.a1d, 0.0}, 0, 1, ch ? -1 : 1, ch};         // matrix + de-emphasis
            }
            const unsigned gc = (unsigned)((n_frames + 3) / 4);
            pss_kernel_begin(ctx, "k_wfm_casc");
            constexpr int CASC_THREADS = 128;
            hipLaunchKernelGGL(k_wfm_casc<1>, dim3(gc), dim3(CASC_THREADS), 0, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), Aa, Pp, Mm,
                               U2, n, n_frames, Lp, swapped, a1);
            pss_kernel_end(ctx);
            pss_kernel_begin(ctx, "k_wfm_casc");
            hipLaunchKernelGGL(k_wfm_casc<2>, dim3(gc), dim3(CASC_THREADS), 0, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), Aa, Pp, Mm,
                               U2, n, n_frames, Lp, swapped, a2);
            pss_kernel_end(ctx);
#else
#ifdef PSS_EXP_WFM_TWOPASS
            BlkArg a1, a2;
            const Biquad idle{0.0, 0.0, 0.0, 0.0, 0.0};
            for (int i = 0; i < 16; i++) a1.lane[i] = a2.lane[i] = BlkLane{idle, 0, 0, -1, 0};
            // pass 1 (rows: 0,1 d | 2,3 | 4,5 a | 6..10 | 11,12 y | 13..16 | 17,18 m)
            for (int i = 0; i < 3; i++) a1.lane[i] = BlkLane{wc.lp[i], i == 0 ? 0 : 1 + i, i == 2 ? 4 : 2 + i, i, 0};               // a: rows 0 -> 2 -> 3 -> 4
            for (int i = 0; i < 5; i++) a1.lane[3 + i] = BlkLane{wc.pil[i], i == 0 ? 0 : 5 + i, 6 + i, i, 0};                          // pilot band-pass: 0 -> 6 .. 10
            a1.lane[8] = BlkLane{Biquad{1.0, 0.0, 0.0, -0.99, 0.0}, 10, 11, 5, 1};                                                       // lfilter([1],[1,-0.99]) -> y
            for (int i = 0; i < 5; i++) a1.lane[9 + i] = BlkLane{wc.lmr[i], i == 0 ? 0 : 12 + i, i == 4 ? 17 : 13 + i, i, 0};           // m: 0 -> 13 .. 16 -> 17
            // pass 2 (rows: 0,1 m*2p | 2,3 | 4,5 lp | 6,7 l | 8,9 r | 10,11 u_l | 12,13 u_r)
            for (int i = 0; i < 3; i++) a2.lane[i] = BlkLane{wc.lp[i], i == 0 ? 0 : 1 + i, i == 2 ? 4 : 2 + i, i, 0};               // :137
            a2.lane[3] = BlkLane{Biquad{wc.b0d, 0.0, 0.0, wc.a1d, 0.0}, 6, 10, 4, 1};                                                    // de-emphasis, left
            a2.lane[4] = BlkLane{Biquad{wc.b0d, 0.0, 0.0, wc.a1d, 0.0}, 8, 12, 4, 1};                                                    // de-emphasis, right
            const unsigned gc = (unsigned)((n_frames + WB_G - 1) / WB_G);
            pss_kernel_begin(ctx, "k_wfm_casc");
            hipLaunchKernelGGL(k_wfm_blk<1>, dim3(gc), dim3(128), 0, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), Aa, Pp, Mm,
                               U2, n, n_frames, Lp, swapped, a1);
            pss_kernel_end(ctx);
            pss_kernel_begin(ctx, "k_wfm_casc");
            hipLaunchKernelGGL(k_wfm_blk<2>, dim3(gc), dim3(128), 0, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), Aa, Pp, Mm,
                               U2, n, n_frames, Lp, swapped, a2);
            pss_kernel_end(ctx);
#else
            {
                MrgArg am;
                const Biquad idle{0.0, 0.0, 0.0, 0.0, 0.0};
                for (int i = 0; i < 27; i++) am.lane[i] = MrgLane{idle, 0, 1, 0, 1, -1, 0, 0.0, 0.0, -1};
                // pass 1: a (rows D1 -> 2 -> 3 -> A ring), pilot band-pass (D1 -> 13..17) + 1-pole (17 -> Y), m (D1 -> 20..23 -> Mo ring)
                for (int i = 0; i < 3; i++)
                    am.lane[i] = MrgLane{wc.lp[i], i == 0 ? WM_D1 : 1 + i, i == 0 ? 2 : 1, i == 2 ? WM_A : 2 + i, i == 2 ? 9 : 1, i, 0, 0.0, 0.0, -1};
                for (int i = 0; i < 5; i++) am.lane[3 + i] = MrgLane{wc.pil[i], i == 0 ? WM_D1 : 12 + i, i == 0 ? 2 : 1, 13 + i, 1, i, 0, 0.0, 0.0, -1};
                am.lane[8] = MrgLane{Biquad{1.0, 0.0, 0.0, -0.99, 0.0}, 17, 1, WM_Y, 2, 5, 1, 0.0, 0.0, -1};            // lfilter([1],[1,-0.99]) -> y
                for (int i = 0; i < 5; i++)
                    am.lane[9 + i] = MrgLane{wc.lmr[i], i == 0 ? WM_D1 : 19 + i, i == 0 ? 2 : 1, i == 4 ? WM_MO : 20 + i, i == 4 ? 3 : 1, i, 0, 0.0, 0.0, -1};
                // pass 2: LP15k on m * 2p (D2 -> 29 -> 30 -> LP), de-emphasis per channel (DL -> UL, DR -> UR)
                for (int i = 0; i < 3; i++)
                    am.lane[14 + i] = MrgLane{wc.lp[i], i == 0 ? WM_D2 : 28 + i, i == 0 ? 2 : 1, i == 2 ? WM_LP : 29 + i, i == 2 ? 2 : 1, 7 + i, 0, 0.0, 0.0, -1};
                am.lane[17] = MrgLane{Biquad{wc.b0d, 0.0, 0.0, wc.a1d, 0.0}, WM_DL, 2, WM_UL, 3, 11, 1, 0.0, 0.0, -1};
                am.lane[18] = MrgLane{Biquad{wc.b0d, 0.0, 0.0, wc.a1d, 0.0}, WM_DR, 2, WM_UR, 3, 11, 1, 0.0, 0.0, -1};
#ifdef PSS_EXP_WFM_NOFWD
                const unsigned gm = (unsigned)((n_frames + 2) / 3);
                pss_kernel_begin(ctx, "k_wfm_casc");
                hipLaunchKernelGGL(k_wfm_mrg<false>, dim3(gm), dim3(128), 0, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), U2, n, n_frames, Lp,
                                   swapped, am);
                pss_kernel_end(ctx);
#else
                // forward half of the zero-phase decimator, per channel (EL -> 47 -> 48 -> 49 -> YL, ER -> 50 -> 51 -> 52 -> YR)
                for (int ch = 0; ch < 2; ch++)
                    for (int i = 0; i < 4; i++)
                        am.lane[19 + 4 * ch + i] = MrgLane{c.s[i], i == 0 ? (ch ? WM_ER : WM_EL) : 46 + 3 * ch + i, i == 0 ? 2 : 1,
                                                           i == 3 ? (ch ? WM_YR : WM_YL) : 47 + 3 * ch + i, i == 3 ? 2 : 1, 13 + i, 0, c.zi[2 * i], c.zi[2 * i + 1], ch};
                const unsigned gm = (unsigned)((n_frames + 1) / 2);
                pss_kernel_begin(ctx, "k_wfm_casc");
                hipLaunchKernelGGL(k_wfm_mrg<true>, dim3(gm), dim3(128), 0, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), Y2, n, n_frames, L,
                                   swapped, am);
                pss_kernel_end(ctx);
#endif
            }
#endif
#endif
            const unsigned gs = (unsigned)((rows + IS_G - 1) / IS_G);
#if defined(PSS_EXP_CASC_SAMPLE) || defined(PSS_EXP_WFM_TWOPASS) || defined(PSS_EXP_WFM_NOFWD)
            pss_kernel_begin(ctx, "k_iir4_sys");
            hipLaunchKernelGGL(k_iir4_sys, dim3(gs), dim3(128), 0, PSS_STREAM(ctx), U2, Lp, 0, L, L, c, Y2, L, q, n_out, nullptr, rows);
            pss_kernel_end(ctx);
#endif
            pss_kernel_begin(ctx, "k_iir4_sys");
            hipLaunchKernelGGL(k_iir4_sys, dim3(gs), dim3(128), 0, PSS_STREAM(ctx), Y2, L, 1, L, L - EDGE, c, A2, (long)n_out, q, n_out, MX2, rows);
            pss_kernel_end(ctx);
            size_t tot = (size_t)n_frames * n_out;
            size_t g2 = (tot + TPB - 1) / TPB;
            if (g2 > 16384) g2 = 16384;
            pss_kernel_begin(ctx, "k_wfm_finalize");
            hipLaunchKernelGGL(k_wfm_finalize, dim3((unsigned)g2), dim3(TPB), 0, PSS_STREAM(ctx), A2, MX2, n_out, n_frames, 2, d_pcm, d_audio);
            pss_kernel_end(ctx);
            pss_time_end(ctx);
            return pss_hip_check(ctx, hipGetLastError(), "wfm small-batch launch");
        }
        if (!q1 && !ctx->no_wfm_fused && b121) {
            // fused path (decimator sections 1..3 with numerator [1, 2, 1]): forward decimator pass inside the front kernel, y_fwd planar-transposed, u[] never stored
            double *Yf = reinterpret_cast<double *>(base), *Af = reinterpret_cast<double *>(base + szY);
            double *MXf = reinterpret_cast<double *>(base + szY + szA);
            // ctx->wfm_scal (pss_demod_signal / pss_frame_pipeline, WFM): d_iq holds the frames as read, corrected on the fly by the kernel
            const float *scal = ctx->wfm_scal;
            ctx->wfm_scal = nullptr;
            auto kf = scal ? (spec ? wfmf::k_wfm_fwd<true, true, true> : wfmf::k_wfm_fwd<false, true, true>)
                           : (spec ? wfmf::k_wfm_fwd<true, true> : wfmf::k_wfm_fwd<false, true>);
            pss_kernel_begin(ctx, "k_wfm_fwd");
            hipLaunchKernelGGL(kf, dim3((unsigned)tiles), dim3(TILE), wfmf::LDS_BYTES, PSS_STREAM(ctx),
                               reinterpret_cast<const float2 *>(d_iq), Yf, n, n_frames, swapped, wc, c, scal);
            pss_kernel_end(ctx);
            auto launch_bwd = [=]() -> int {
                pss_kernel_begin(ctx, "k_nfm_bwd");
                hipLaunchKernelGGL((fused::k_nfm_bwd<true, true>), dim3((unsigned)tiles), dim3(2 * TILE), 0, PSS_STREAM(ctx), Yf, Af,
                                   n, q, n_out, n_frames, c, d_pcm, d_audio);      // both channels of a tile: joint normalisation + stereo PCM inside
                pss_kernel_end(ctx);
                return pss_hip_check(ctx, hipGetLastError(), "wfm fused launch (backward)");
            };
            if (ctx->defer_bwd) {  // pss_frame_pipeline places the backward pass + the L / R normalisation itself (beside the display chain)
                ctx->pending_bwd = launch_bwd;
                pss_time_end(ctx);
                return pss_hip_check(ctx, hipGetLastError(), "wfm fused launch");
            }
            const int rb = launch_bwd();
            pss_time_end(ctx);
            return rb;
        }
        pss_kernel_begin(ctx, "k_wfm_front");
        hipLaunchKernelGGL(spec ? k_wfm_front<true> : k_wfm_front<false>, dim3((unsigned)tiles), dim3(TILE), 0, PSS_STREAM(ctx),
                           reinterpret_cast<const float2 *>(d_iq), U, n, n_frames, Lp, swapped, wc);
        pss_kernel_end(ctx);
        if (q1) {
            // no decimate() stage: the de-emphasised channels are normalised as they are (n_out = n - 1 samples per channel)
            pss_kernel_begin(ctx, "k_wfm_rows_q1");
            hipLaunchKernelGGL(k_wfm_rows_q1, dim3((unsigned)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192)), dim3(256), 0, PSS_STREAM(ctx), U, Lp, n - 1, rows, A, MX);
            pss_kernel_end(ctx);
            size_t tot1 = (size_t)n_frames * n_out;
            size_t g1 = (tot1 + TPB - 1) / TPB;
            if (g1 > 16384) g1 = 16384;
            pss_kernel_begin(ctx, "k_wfm_finalize");
            hipLaunchKernelGGL(k_wfm_finalize, dim3((unsigned)g1), dim3(TPB), 0, PSS_STREAM(ctx), A, MX, n_out, n_frames, 2, d_pcm, d_audio);
            pss_kernel_end(ctx);
            pss_time_end(ctx);
            return pss_hip_check(ctx, hipGetLastError(), "wfm (decimation factor 1) launch");
        }
        pss_kernel_begin(ctx, "k_nfm_iir");
        if (b121)
            hipLaunchKernelGGL((k_nfm_iir<true, true>), dim3((unsigned)tiles2), dim3(TILE), 0, PSS_STREAM(ctx), U, Y, A, n, q,
                               n_out, rows, Lp, c, nullptr, MX);
        else
            hipLaunchKernelGGL((k_nfm_iir<false, true>), dim3((unsigned)tiles2), dim3(TILE), 0, PSS_STREAM(ctx), U, Y, A, n, q,
                               n_out, rows, Lp, c, nullptr, MX);
        pss_kernel_end(ctx);
        size_t tot = (size_t)n_frames * n_out;
        size_t g2 = (tot + TPB - 1) / TPB;
        if (g2 > 16384) g2 = 16384;
        pss_kernel_begin(ctx, "k_wfm_finalize");
        hipLaunchKernelGGL(k_wfm_finalize, dim3((unsigned)g2), dim3(TPB), 0, PSS_STREAM(ctx), A, MX, n_out, n_frames, 0, d_pcm,
                           d_audio);
        pss_kernel_end(ctx);
        pss_time_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "wfm launch");
    }
    return pss_fail(ctx, PSS_E_ARG, "unknown demodulation mode");
}

// demodulate_signal (signal_processing.py:220-240): the voice modes go straight to their demodulator, every other mode
// is IQ-corrected first (:222-225).
extern "C" int pss_demod_signal(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, int16_t *d_pcm,
                                double *d_audio)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (mode != PSS_MODE_WFM) return pss_demod(ctx, mode, d_iq, n_frames, n, fs, d_pcm, d_audio);
    if (n_frames < 0 || n < 1 || (n_frames > 0 && !d_iq)) return pss_fail(ctx, PSS_E_ARG, "bad demod arguments");
    if (n_frames == 0) return pss_demod(ctx, mode, d_iq, n_frames, n, fs, d_pcm, d_audio);
    PssFlagScope corr(ctx->wfm_correct, true);    // the WFM branch of pss_demod runs iq_correction itself (scalars pre-pass or a corrected copy)
    return pss_demod(ctx, mode, d_iq, n_frames, n, fs, d_pcm, d_audio);
}

// measure_signal_power + demodulate of the same read buffers, as the main loop runs them back to back (pyspecsdr.py:2251, :2262): d_power
// float32 [n_frames] = 10 log10(mean |x|^2 + 1e-10) (pss_power_db's bits), then pss_demod.  AM: both np.mean reductions (|x|^2 and |x|) come
// out of ONE pass over the IQ (k_pairwise2); the other modes: the two calls in order.
extern "C" int pss_demod_power(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, int16_t *d_pcm, double *d_audio,
                               float *d_power)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_power && n_frames > 0) return pss_fail(ctx, PSS_E_ARG, "pss_demod_power: d_power is null");
    if (mode == PSS_MODE_AM && n_frames > 0 && n >= 1 && d_iq) {
        PssScoped<float *> want(ctx->power_out, d_power);
        return pss_demod(ctx, mode, d_iq, n_frames, n, fs, d_pcm, d_audio);
    }
    pss_time_begin(ctx);
    int r = pss_power_db(ctx, d_iq, n_frames, n, d_power);
    if (!r) r = pss_demod(ctx, mode, d_iq, n_frames, n, fs, d_pcm, d_audio);
    pss_time_end(ctx);
    return r;
}

extern "C" int pss_sosfilt(pss_ctx *ctx, const double *d_x, long n_rows, int n, const double *sos, int nsec, double *d_y)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_x || !d_y || !sos || n_rows < 0 || n < 0 || nsec < 1 || nsec > 8) return pss_fail(ctx, PSS_E_ARG, "pss_sosfilt: bad argument");
    if (n_rows == 0 || n == 0) return PSS_OK;
    SosArg a;
    a.nsec = nsec;
    for (int s2 = 0; s2 < 8; s2++) {
        const double *row = sos + 6 * (s2 < nsec ? s2 : 0);
        a.s[s2] = Biquad{row[0], row[1], row[2], row[4], row[5]};
    }
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_sosfilt");
    hipLaunchKernelGGL(k_sosfilt, dim3((unsigned)((n_rows + TILE - 1) / TILE)), dim3(TILE), 0, PSS_STREAM(ctx), d_x, d_y, n, n_rows, a);
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_sosfilt launch");
}

// ---- complex128 read buffers: demodulate_am in float64 from the first statement (signal_processing.py:179-195) ---------------------------------
// The reference's SDR buffer is complex64, but handed complex128 its functions compute in float64: np.abs(complex128) is the same scaled hypot
// as the complex64 loop in float64 (mx * sqrt(fma(r, r, 1)), r = mn / mx; probed bit for bit), np.mean the same pairwise tree over float64
// (8192-element chunks added in order; blocks of 128 with 8 accumulators), the subtraction float64; sosfilt and the normalisation are float64
// anyway.  Plain kernels, one workgroup per frame, the mean by one thread — a conformance path, not a throughput path.
namespace {
__device__ double cabs_np64(double re, double im)
{
    const double a = fabs(re), b = fabs(im);
    const double mx = a > b ? a : b, mn = a > b ? b : a;
    if (mx == 0.0) return 0.0;
    const double r = __ddiv_rn(mn, mx);
    return __dmul_rn(mx, __dsqrt_rn(__fma_rn(r, r, 1.0)));
}
// numpy's DOUBLE_pairwise_sum over a[0 .. n), n <= 8192: the recursion as an explicit post-order walk (depth <= 7)
__device__ double pairwise_chunk_f64(const double *a, int n)
{
    struct Fr { int off, len, st; double left; };
    Fr stk[12];
    int sp = 0;
    double ret = 0.0;
    stk[sp++] = Fr{0, n, 0, 0.0};
    while (sp > 0) {
        Fr &fr = stk[sp - 1];
        if (fr.len <= 128) {
            const double *x = a + fr.off;
            const int len = fr.len;
            double res;
            if (len < 8) {
                res = 0.0;
                for (int i = 0; i < len; i++) res = __dadd_rn(res, x[i]);
            } else {
                double r[8];
                for (int j = 0; j < 8; j++) r[j] = x[j];
                int i = 8;
                for (; i < len - (len % 8); i += 8)
                    for (int j = 0; j < 8; j++) r[j] = __dadd_rn(r[j], x[i + j]);
                res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])), __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
                for (; i < len; i++) res = __dadd_rn(res, x[i]);
            }
            ret = res;
            sp--;
        } else if (fr.st == 0) {
            int n2 = fr.len / 2;
            n2 -= n2 % 8;
            fr.st = 1;
            stk[sp] = Fr{fr.off, n2, 0, 0.0};
            sp++;
        } else if (fr.st == 1) {
            int n2 = fr.len / 2;
            n2 -= n2 % 8;
            fr.left = ret;
            fr.st = 2;
            stk[sp] = Fr{fr.off + n2, fr.len - n2, 0, 0.0};
            sp++;
        } else {
            ret = __dadd_rn(fr.left, ret);
            sp--;
        }
    }
    return ret;
}
// envelope |x| (float64), its mean, X[f][i] = |x[i]| - mean
__global__ __launch_bounds__(256) void k_am_env_c128(const double2 *__restrict__ iq, int n, long n_frames, double *__restrict__ X)
{
    __shared__ double mu_s;
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const double2 *x = iq + (size_t)f * n;
        double *e = X + (size_t)f * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) e[i] = cabs_np64(x[i].x, x[i].y);
        __syncthreads();
        if (threadIdx.x == 0) {
            double acc = pairwise_chunk_f64(e, n < 8192 ? n : 8192);
            for (int st = 8192; st < n; st += 8192) acc = __dadd_rn(acc, pairwise_chunk_f64(e + st, (n - st) < 8192 ? (n - st) : 8192));
            mu_s = __ddiv_rn(acc, (double)n);
        }
        __syncthreads();
        const double mu = mu_s;
        for (int i = threadIdx.x; i < n; i += blockDim.x) e[i] = __dsub_rn(e[i], mu);
        __syncthreads();
    }
}
// measure_signal_power's array part for complex128 frames (:327): P[f] = np.mean(np.abs(x) ** 2) in float64; E: n doubles of scratch per frame
__global__ __launch_bounds__(256) void k_power_c128(const double2 *__restrict__ iq, int n, long n_frames, double *__restrict__ E, double *__restrict__ P)
{
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const double2 *x = iq + (size_t)f * n;
        double *e = E + (size_t)f * n;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double a = cabs_np64(x[i].x, x[i].y);
            e[i] = __dmul_rn(a, a);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double acc = pairwise_chunk_f64(e, n < 8192 ? n : 8192);
            for (int st = 8192; st < n; st += 8192) acc = __dadd_rn(acc, pairwise_chunk_f64(e + st, (n - st) < 8192 ? (n - st) : 8192));
            P[f] = __ddiv_rn(acc, (double)n);
        }
        __syncthreads();
    }
}
// audio / np.max(np.abs(audio)) * 0.95 (:194; np.max propagates a NaN), mono float64 and / or int16 stereo
__global__ __launch_bounds__(256) void k_norm_rows_f64(const double *__restrict__ Y, int n, long n_frames, int16_t *__restrict__ pcm, double *__restrict__ audio)
{
    __shared__ double red[256];
    __shared__ int nan_s[256];
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const double *y = Y + (size_t)f * n;
        double mx = 0.0;
        int has_nan = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double a = fabs(y[i]);
            has_nan |= a != a;
            mx = a > mx ? a : mx;
        }
        red[threadIdx.x] = mx;
        nan_s[threadIdx.x] = has_nan;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) {
                red[threadIdx.x] = red[threadIdx.x + st] > red[threadIdx.x] ? red[threadIdx.x + st] : red[threadIdx.x];
                nan_s[threadIdx.x] |= nan_s[threadIdx.x + st];
            }
            __syncthreads();
        }
        const double peak = nan_s[0] ? __builtin_nan("") : red[0];
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const double a = __dmul_rn(__ddiv_rn(y[i], peak), 0.95);
            if (audio) audio[(size_t)f * n + i] = a;
            if (pcm) {
                const uint16_t s16 = (uint16_t)pcm16(a);
                reinterpret_cast<uint32_t *>(pcm)[(size_t)f * n + i] = (uint32_t)s16 | ((uint32_t)s16 << 16);
            }
        }
        __syncthreads();
    }
}
}  // namespace

// demodulate_am of complex128 frames: d_iq interleaved float64 (re, im) [n_frames][n]; d_pcm int16 [n_frames][n][2] and / or d_audio float64
// [n_frames][n] (mono, before stereo duplication).  Same results as the reference handed a complex128 buffer (tests/golden/c128.npz).
extern "C" int pss_demod_am_c128(pss_ctx *ctx, const double *d_iq, long n_frames, int n, int16_t *d_pcm, double *d_audio)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || n < 1 || (n_frames > 0 && (!d_iq || (!d_pcm && !d_audio)))) return pss_fail(ctx, PSS_E_ARG, "pss_demod_am_c128: bad argument");
    if (n_frames == 0) return PSS_OK;
    const size_t rows = align256((size_t)n_frames * n * sizeof(double));
    int r = pss_ensure_scratch(ctx, 2 * rows);
    if (r) return r;
    double *X = reinterpret_cast<double *>(ctx->scratch), *Y = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->scratch) + rows);
    double sos[30];
    pss_am_bandpass_sos(sos);
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_am_env_c128");
    hipLaunchKernelGGL(k_am_env_c128, dim3((unsigned)(n_frames < 4096 ? n_frames : 4096)), dim3(256), 0, PSS_STREAM(ctx), reinterpret_cast<const double2 *>(d_iq), n,
                       n_frames, X);
    pss_kernel_end(ctx);
    r = pss_hip_check(ctx, hipGetLastError(), "k_am_env_c128 launch");
    if (!r) r = pss_sosfilt(ctx, X, n_frames, n, sos, 5, Y);
    if (!r) {
        pss_kernel_begin(ctx, "k_norm_rows_f64");
        hipLaunchKernelGGL(k_norm_rows_f64, dim3((unsigned)(n_frames < 4096 ? n_frames : 4096)), dim3(256), 0, PSS_STREAM(ctx), Y, n, n_frames, d_pcm, d_audio);
        pss_kernel_end(ctx);
        r = pss_hip_check(ctx, hipGetLastError(), "k_norm_rows_f64 launch");
    }
    pss_time_end(ctx);
    return r;
}

// demodulate_ssb (signal_processing.py:198-217) of complex128 frames: lfilter's complex128 convolution on the samples as they are (for a complex64 buffer
// the reference widens first — the same kernels behind a float64 loader), the hilbert() round trip, normalisation, int16.  d_iq: interleaved float64
// (re, im) [n_frames][n]; d_pcm int16 [n_frames][n][2] and / or d_audio float64 [n_frames][n] (mono).  tests/golden/c128.npz keys ssb_*.
extern "C" int pss_demod_ssb_c128(pss_ctx *ctx, int lower, const double *d_iq, long n_frames, int n, double fs, int16_t *d_pcm, double *d_audio)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || n < 1 || (n_frames > 0 && (!d_iq || (!d_pcm && !d_audio)))) return pss_fail(ctx, PSS_E_ARG, "pss_demod_ssb_c128: bad argument");
    PssFlagScope wide(ctx->iq_c128, true);
    return pss_demod(ctx, lower ? PSS_MODE_LSB : PSS_MODE_USB, reinterpret_cast<const float *>(d_iq), n_frames, n, fs, d_pcm, d_audio);
}

// measure_signal_power (signal_processing.py:325-328) of complex128 frames, the array part: d_power[f] = np.mean(np.abs(x) ** 2) in float64 as the
// reference computes it for such a buffer.  The scalar that follows — 10 * log10(power + 1e-10), NumPy's float64 log10 — is the caller's (the Python
// shim applies NumPy's own; tests/golden/c128.npz keys mp_* / pw_*).
extern "C" int pss_mean_power_c128(pss_ctx *ctx, const double *d_iq, long n_frames, int n, double *d_power)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || n < 1 || (n_frames > 0 && (!d_iq || !d_power))) return pss_fail(ctx, PSS_E_ARG, "pss_mean_power_c128: bad argument");
    if (n_frames == 0) return PSS_OK;
    int r = pss_ensure_scratch(ctx, align256((size_t)n_frames * n * sizeof(double)));
    if (r) return r;
    pss_kernel_begin(ctx, "k_power_c128");
    hipLaunchKernelGGL(k_power_c128, dim3((unsigned)(n_frames < 4096 ? n_frames : 4096)), dim3(256), 0, PSS_STREAM(ctx), reinterpret_cast<const double2 *>(d_iq), n,
                       n_frames, reinterpret_cast<double *>(ctx->scratch), d_power);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_power_c128 launch");
}

extern "C" int pss_afsk_n_bits(int n, double fs)
{
    const int w = (int)(fs / 1200.0);
    if (w < 1 || n - w <= 0) return 0;
    return (n - w + w - 1) / w;  // len(range(0, n - w, w))
}

extern "C" int pss_morse_edges(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double threshold_db, int cap,
                               int32_t *d_rise, int32_t *d_fall, int32_t *d_counts)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || n < 1 || cap < 0) return pss_fail(ctx, PSS_E_ARG, "pss_morse_edges: bad argument");
    if (n_frames == 0) return PSS_OK;
    if (!d_iq || !d_counts || (cap > 0 && (!d_rise || !d_fall))) return pss_fail(ctx, PSS_E_ARG, "pss_morse_edges: null buffer");
    const long g = n_frames < 16384 ? n_frames : 16384;
    pss_kernel_begin(ctx, "k_morse_edges");
    hipLaunchKernelGGL(k_morse_edges, dim3((unsigned)g), dim3(256), 0, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), n,
                       n_frames, cap, d_rise, d_fall, d_counts, (float)threshold_db, threshold_db != -20.0 ? 1 : 0);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_morse_edges launch");
}

extern "C" const char *pss_class_name(int label)
{
    static const char *names[] = {"UNKNOWN", "FM_BROADCAST", "NARROW_FM", "AM_BROADCAST", "SSB", "DIGITAL"};
    return (label >= 0 && label < 6) ? names[label] : "UNKNOWN";
}

extern "C" int pss_classify(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double fs, int32_t *d_label, double *d_bw,
                            float *d_mi, float *d_flat, float *d_psd)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || !(fs > 0.0)) return pss_fail(ctx, PSS_E_ARG, "pss_classify: bad argument");
    if (n < 1) return pss_fail(ctx, PSS_E_ARG, "pss_classify: empty read (welch raises on it)");
    if (n_frames == 0) return PSS_OK;
    if (!d_iq) return pss_fail(ctx, PSS_E_ARG, "pss_classify: null input");
    // Hann window as scipy.signal.get_window('hann', np) builds it (general_cosine over np.linspace(-pi, pi, np + 1); ones for
    // np <= 1), cast to float32; scale = 1 / (fs * sum(win * win)) in complex64 arithmetic with zero imaginary parts.
    // np = 1024, or the read length when that is shorter (one segment: SciPy's nperseg = n fallback)
    const int np = n < CLS_NP ? n : CLS_NP;
    float *&d_win = np == CLS_NP ? ctx->d_hann : ctx->d_hann_short;
    float &win_sum = np == CLS_NP ? ctx->hann_sum : ctx->hann_short_sum;
    if (!d_win || (np != CLS_NP && ctx->hann_short_n != np)) {
        std::vector<float> w(np);
        const double start = -M_PI, step = (M_PI - (-M_PI)) / (double)np;
        for (int i = 0; i < np; i++) w[i] = np == 1 ? 1.0f : (float)(0.5 + 0.5 * cos((double)i * step + start));
        // (win * win).sum(): numpy's pairwise sum over the 2 np interleaved floats of the complex64 array (imaginary lanes all
        // zero): blocks of <= 128 floats with 8 accumulators striding the array, a plain loop below 8 floats
        std::vector<float> z(2 * np, 0.0f);
        for (int i = 0; i < np; i++) z[2 * i] = w[i] * w[i];
        std::function<float(const float *, int)> pw = [&](const float *a, int nn) -> float {  // real part of numpy's pairwise complex sum
            if (nn < 8) {
                float res = 0.0f;
                for (int i = 0; i < nn; i += 2) res += a[i];
                return res;
            }
            if (nn <= 128) {
                float r[8];
                for (int k = 0; k < 8; k++) r[k] = a[k];
                int i;
                for (i = 8; i < nn - (nn % 8); i += 8)
                    for (int k = 0; k < 8; k++) r[k] += a[i + k];
                float res = ((r[0] + r[2]) + (r[4] + r[6]));
                for (; i < nn; i += 2) res += a[i];
                return res;
            }
            int n2 = nn / 2;
            n2 -= n2 % 8;
            return pw(a, n2) + pw(a + n2, nn - n2);
        };
        win_sum = pw(z.data(), 2 * np);
        if (!d_win) PSS_HIP(ctx, hipMalloc(&d_win, sizeof(float) * CLS_NP));
        PSS_HIP(ctx, hipMemcpyAsync(d_win, w.data(), sizeof(float) * np, hipMemcpyHostToDevice, PSS_STREAM(ctx)));
        PSS_HIP(ctx, hipStreamSynchronize(PSS_STREAM(ctx)));   // w is a local; also orders the upload behind earlier launches reading the old window
        if (np != CLS_NP) ctx->hann_short_n = np;
    }
    const float scale = 1.0f / ((float)fs * win_sum);
    RedPlan rn, rm, cp;
    int l1, v1, l2, v2, lc, vc;
    int r = get_red_plan(ctx, n, false, &rn, &l1, &v1);
    if (!r) r = get_red_plan(ctx, n - 1, false, &rm, &l2, &v2);
    if (!r) r = get_red_plan(ctx, np, true, &cp, &lc, &vc);
    if (r) return r;
    const size_t szUp = align256((size_t)n_frames * n * sizeof(float)), szMi = align256((size_t)n_frames * sizeof(float));
    r = pss_ensure_scratch(ctx, szUp + szMi);
    if (r) return r;
    float *up = reinterpret_cast<float *>(ctx->scratch);
    float *mi = d_mi ? d_mi : reinterpret_cast<float *>(reinterpret_cast<char *>(ctx->scratch) + szUp);
    const int part1 = 8 * (l1 > l2 ? l1 : l2), val1 = ((v1 > v2 ? v1 : v2) + 3) & ~3;
    const size_t lds1 = sizeof(float) * ((size_t)part1 + val1 + (CLS_CH + 4) + CLS_CH + 1) + plan_lds_bytes(rn) + plan_lds_bytes(rm);
    if (lds1 > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_cls_modidx), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    const int partc = 4 * lc;
    const size_t lds2 = sizeof(double2) * (CLS_NP + CLS_NP / 2) + sizeof(float2) * ((size_t)partc + vc) + plan_lds_bytes(cp);
    const long g = n_frames < 16384 ? n_frames : 16384;
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_cls_modidx");
    hipLaunchKernelGGL(k_cls_modidx, dim3((unsigned)g), dim3(256), lds1, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), n,
                       n_frames, rn, rm, part1, val1, up, mi);
    pss_kernel_end(ctx);
    pss_kernel_begin(ctx, "k_cls_welch");
    if (np == CLS_NP)
        hipLaunchKernelGGL(k_cls_welch, dim3((unsigned)g), dim3(256), lds2, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), n,
                           n_frames, fs, cp, partc, vc, d_win, scale, mi, d_label, d_bw, d_flat, d_psd);
    else {
        const size_t lds3 = sizeof(double2) * 2 * CLS_NP + sizeof(float) * 2 * CLS_NP + sizeof(double) * 16 + sizeof(int) * 8 +
                            sizeof(float2) * ((size_t)partc + vc) + plan_lds_bytes(cp);
        hipLaunchKernelGGL(k_cls_welch_short, dim3((unsigned)g), dim3(256), lds3, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq),
                           n, n_frames, fs, cp, partc, vc, d_win, scale, mi, d_label, d_bw, d_flat, d_psd);
    }
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "classify launch");
}

extern "C" int pss_afsk_bits(pss_ctx *ctx, const double *d_audio, long n_rows, int n, double fs, const double *sos1200,
                             const double *sos2200, int nsec, uint8_t *d_bits)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_rows < 0 || n < 0 || !(fs >= 1200.0)) return pss_fail(ctx, PSS_E_ARG, "pss_afsk_bits: bad argument");
    double t1[48], t2[48];
    if (!sos1200 || !sos2200) {  // design butter(5) band-passes as bandpass_filter does (signal_processing.py:41)
        const double nyq = fs / 2.0;
        int r = pss_design_butter_sos(5, 1100.0 / nyq, 1300.0 / nyq, t1, &nsec);
        if (!r) r = pss_design_butter_sos(5, 2100.0 / nyq, 2300.0 / nyq, t2, &nsec);
        if (r) return pss_fail(ctx, r, "butter: digital filter critical frequencies must be 0 < Wn < 1");
        sos1200 = t1; sos2200 = t2;
    }
    const int n_bits = pss_afsk_n_bits(n, fs);
    if (n_rows == 0 || n_bits == 0) return PSS_OK;
    if (!d_audio || !d_bits) return pss_fail(ctx, PSS_E_ARG, "pss_afsk_bits: null buffer");
    const size_t szF = align256((size_t)n_rows * n * sizeof(double));
    int r = pss_ensure_scratch(ctx, 2 * szF);
    if (r) return r;
    double *f1 = reinterpret_cast<double *>(ctx->scratch), *f2 = reinterpret_cast<double *>(reinterpret_cast<char *>(ctx->scratch) + szF);
    pss_time_begin(ctx);
    r = pss_sosfilt(ctx, d_audio, n_rows, n, sos1200, nsec, f1);
    if (!r) r = pss_sosfilt(ctx, d_audio, n_rows, n, sos2200, nsec, f2);
    if (r) { pss_time_end(ctx); return r; }
    const long total = n_rows * n_bits;
    pss_kernel_begin(ctx, "k_afsk_bits");
    hipLaunchKernelGGL(k_afsk_bits, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0,
                       PSS_STREAM(ctx), f1, f2, n, (int)(fs / 1200.0), n_bits, n_rows, d_bits);
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_afsk_bits launch");
}

// The float32 building blocks on their own — np.arctan2 / np.log10 / np.abs(complex64) as NumPy's AVX512_SKX loops evaluate
// them — so that the device models can be pinned element by element against NumPy's outputs (tests/golden/atan2f.npz,
// log10f.npz), not only through the demodulators that use them.
__global__ __launch_bounds__(256) void k_np_f32(int op, const float *__restrict__ a, const float *__restrict__ b, long n,
                                                float *__restrict__ out)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float r;
        if (op == PSS_NP_ARCTAN2) r = atan2f_svml(a[i], b[i]);
        else if (op == PSS_NP_LOG10) r = log10f_np(a[i]);
        else r = cabsf_np(a[i], b[i]);
        out[i] = r;
    }
}

extern "C" int pss_np_f32(pss_ctx *ctx, int op, const float *d_a, const float *d_b, long n, float *d_out)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (op < PSS_NP_ARCTAN2 || op > PSS_NP_ABS || n < 0) return pss_fail(ctx, PSS_E_ARG, "pss_np_f32: bad argument");
    if (n == 0) return PSS_OK;
    if (!d_a || !d_out || (op != PSS_NP_LOG10 && !d_b)) return pss_fail(ctx, PSS_E_ARG, "pss_np_f32: null pointer");
    const long blocks = (n + 255) / 256;
    pss_kernel_begin(ctx, "k_np_f32");
    hipLaunchKernelGGL(k_np_f32, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, PSS_STREAM(ctx), op, d_a, d_b, n, d_out);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_np_f32 launch");
}

extern "C" int pss_row_normalise(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_y)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_rows < 0 || n < 0 || (n_rows > 0 && n > 0 && (!d_x || !d_y))) return pss_fail(ctx, PSS_E_ARG, "pss_row_normalise: bad argument");
    if (n_rows == 0 || n == 0) return PSS_OK;
    pss_kernel_begin(ctx, "k_row_normalise");
    hipLaunchKernelGGL(k_row_normalise, dim3((unsigned)(n_rows < 8192 ? n_rows : 8192)), dim3(256), 0, PSS_STREAM(ctx), d_x, d_y, n, n_rows);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_row_normalise launch");
}

// decode_afsk on one HOST buffer of real float64 audio, optionally normalised first the way decode_aprs does
// (decoders.py:126): upload, [normalise,] two band-passes, per-bit energy compare, download — no torch involved.
extern "C" int pss_h_afsk_bits(pss_ctx *ctx, const double *h_audio, int n, double fs, int normalise, const double *sos1200,
                               const double *sos2200, int nsec, uint8_t *h_bits)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n < 0 || !(fs >= 1200.0) || (n > 0 && !h_audio)) return pss_fail(ctx, PSS_E_ARG, "pss_h_afsk_bits: bad argument");
    const int n_bits = pss_afsk_n_bits(n, fs);
    if (n_bits <= 0) return PSS_OK;
    if (!h_bits) return pss_fail(ctx, PSS_E_ARG, "pss_h_afsk_bits: null bit buffer");
    const size_t o_n = align256(sizeof(double) * (size_t)n), o_b = 2 * o_n;
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_b + align256((size_t)n_bits), "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    double *d_x = reinterpret_cast<double *>(base), *d_nrm = reinterpret_cast<double *>(base + o_n);
    PSS_HIP(ctx, hipMemcpyAsync(d_x, h_audio, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    if (normalise) {
        r = pss_row_normalise(ctx, d_x, 1, n, d_nrm);
        if (r) return r;
    }
    r = pss_afsk_bits(ctx, normalise ? d_nrm : d_x, 1, n, fs, sos1200, sos2200, nsec, reinterpret_cast<uint8_t *>(base + o_b));
    if (r) return r;
    PSS_HIP(ctx, hipMemcpyAsync(h_bits, base + o_b, (size_t)n_bits, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSS_OK;
}

extern "C" int pss_set_wfm_filters(pss_ctx *ctx, double fs, const double *lp3x6, const double *pilot5x6, const double *lmr5x6,
                                   double alpha)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!lp3x6 || !pilot5x6 || !lmr5x6) return pss_fail(ctx, PSS_E_ARG, "null coefficient table");
    PssWfmFilt f;
    memcpy(f.lp, lp3x6, sizeof(f.lp));
    memcpy(f.pilot, pilot5x6, sizeof(f.pilot));
    memcpy(f.lmr, lmr5x6, sizeof(f.lmr));
    f.alpha = alpha;
    ctx->wfm[fs] = f;
    return PSS_OK;
}

extern "C" int pss_get_wfm_filters(pss_ctx *ctx, double fs, double *lp3x6, double *pilot5x6, double *lmr5x6, double *alpha)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    PssWfmFilt *f;
    int r = wfm_filters(ctx, fs, &f);
    if (r) return r;
    if (lp3x6) memcpy(lp3x6, f->lp, sizeof(f->lp));
    if (pilot5x6) memcpy(pilot5x6, f->pilot, sizeof(f->pilot));
    if (lmr5x6) memcpy(lmr5x6, f->lmr, sizeof(f->lmr));
    if (alpha) *alpha = f->alpha;
    return PSS_OK;
}


extern "C" int pss_spectrum_nfm(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double fs, float *d_db,
                                int16_t *d_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    // The backward IIR pass runs one wavefront per SIMD and is latency-bound; the spectrum kernel (HBM-bound, high
    // occupancy) is launched on a side stream right behind the forward kernel so that the two share the machine
    // (fork / join with events).  Only the fused large-batch NFM path honours fork_after_fwd, and reports it in did_fork.
    pss_time_begin(ctx);  // nested begin/end pairs inside the two calls are no-ops
    int r2;
    {
        PssFlagScope fork(ctx->fork_after_fwd, true);
        ctx->did_fork = false;
        r2 = pss_demod(ctx, PSS_MODE_NFM, d_iq, n_frames, n, fs, d_pcm, nullptr);
    }
    int r;
    if (ctx->did_fork) {
        ctx->did_fork = false;
        r = pss_hip_check(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0), "hipStreamWaitEvent(fork)");
        if (!r && !r2) {
            PssStreamScope side(ctx->cur, ctx->stream2);
            r = pss_spectrum_db(ctx, d_iq, n_frames, n, d_db);
        }
        // the join is attempted whatever happened above: the main stream must never run ahead of the side stream
        int rj = pss_hip_check(ctx, hipEventRecord(ctx->ev_join, ctx->stream2), "hipEventRecord(join)");
        if (!rj) rj = pss_hip_check(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0), "hipStreamWaitEvent(join)");
        if (!r) r = rj;
    } else {
        r = r2 ? r2 : pss_spectrum_db(ctx, d_iq, n_frames, n, d_db);
    }
    pss_time_end(ctx);
    return r2 ? r2 : r;
}

// One iteration of the reference's main loop for a whole batch of read buffers (pyspecsdr.py:2262-2283 + the display call):
// demodulate_signal(samples, fs, mode) -> int16; compute_fft -> dB row; smoothing + median clamp; display accumulator line.
// Rows of either type through the SAME schedule: TR = float (pss_frame_pipeline[_nfm]: float32 dB rows, the contract of the spectrum
// output) or TR = double (pss_frame_pipeline_nfm_f64: the reference's own row type from IQ to cells — compute_fft returns float64 and the
// caller smooths, clamps and draws float64: these are the reference's cells).
namespace {
inline int pipe_spectrum(pss_ctx *ctx, const float *d_iq, long nf, int n, float *d_db) { return pss_spectrum_db(ctx, d_iq, nf, n, d_db); }
inline int pipe_spectrum(pss_ctx *ctx, const float *d_iq, long nf, int n, double *d_db) { return pss_spectrum_db_f64(ctx, d_iq, nf, n, d_db); }
inline int pipe_post(pss_ctx *ctx, const float *d_db, long nf, int n, float *d_post, float *lo, float *hi) { return pss_spectrum_post_extremes(ctx, d_db, nf, n, d_post, lo, hi); }
inline int pipe_post(pss_ctx *ctx, const double *d_db, long nf, int n, double *d_post, double *lo, double *hi) { return pss_spectrum_post_f64(ctx, d_db, nf, n, d_post, lo, hi); }
inline int pipe_lines(pss_ctx *ctx, int display, const float *d_post, long nf, int len, const float *lo, const float *hi, int n_halo, int window, int disp_h,
                      int disp_w, int8_t *a, int8_t *b)
{
    return display ? pss_persistence_rows(ctx, d_post, nf, len, lo, hi, n_halo, window, disp_h, disp_w, a)
                   : pss_waterfall_rows(ctx, d_post, nf, len, lo, hi, n_halo, window, disp_w, a, b);
}
inline int pipe_lines(pss_ctx *ctx, int display, const double *d_post, long nf, int len, const double *lo, const double *hi, int n_halo, int window, int disp_h,
                      int disp_w, int8_t *a, int8_t *b)
{
    return display ? pss_persistence_rows_f64(ctx, d_post, nf, len, lo, hi, n_halo, window, disp_h, disp_w, a)
                   : pss_waterfall_rows_f64(ctx, d_post, nf, len, lo, hi, n_halo, window, disp_w, a, b);
}
inline int pipe_chain_vals(pss_ctx *ctx, const float *d_db, long nf, int n, float *lo, float *hi, int n_halo, int window, int display, int disp_h, int disp_w,
                           int8_t *a, int8_t *b, double *vals)
{
    return pss_chain_vals_f32(ctx, d_db, nf, n, lo, hi, n_halo, window, display, disp_h, disp_w, a, b, vals);
}
inline int pipe_chain_vals(pss_ctx *ctx, const double *d_db, long nf, int n, double *lo, double *hi, int n_halo, int window, int display, int disp_h, int disp_w,
                           int8_t *a, int8_t *b, double *vals)
{
    return pss_chain_vals_f64(ctx, d_db, nf, n, lo, hi, n_halo, window, display, disp_h, disp_w, a, b, vals);
}
}  // namespace

// display: 0 = the waterfall accumulator's newest line (d_glyph, d_colour), 1 = the persistence accumulator's newest trace (d_glyph = row
// index per column, d_colour unused).
// d_post == NULL: the post-processed rows are not materialised.  Rows the register select serves (a multiple of 4 points, up to 32 772 /
// float64: 16 388): ONE pass over the dB rows leaves per row the extremes and the row resampled to the display width (disp_w float64
// values: what the accumulators normalise and quantise), and the lines are quantised from those — the same bytes as from materialised rows.
// Other lengths go through a context-owned scratch copy of the rows.
template <class TR>
static int frame_pipeline_impl(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, TR *d_db, TR *d_post,
                               TR *d_row_lo, TR *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w,
                               int8_t *d_glyph, int8_t *d_colour, int16_t *d_pcm, float *d_db32, bool demodulate);
// d_db32 (float64 rows only; NULL otherwise): the dB rows ALSO (or, with d_db == NULL, ONLY) as float32 — compute_fft's float64 value rounded once
// demodulate = false: the display half alone (pss_spectrum_cells): no demodulator, d_pcm unused
template <class TR>
static int frame_pipeline(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, TR *d_db, TR *d_post,
                          TR *d_row_lo, TR *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w,
                          int8_t *d_glyph, int8_t *d_colour, int16_t *d_pcm, float *d_db32 = nullptr, bool demodulate = true)
{
    pss_time_begin(ctx);     // one bracket around the whole call (the nested pairs inside are no-ops)
    const int r = frame_pipeline_impl<TR>(ctx, mode, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, n_halo, window, display, disp_h, disp_w,
                                          d_glyph, d_colour, d_pcm, d_db32, demodulate);
    pss_time_end(ctx);
    return r;
}

__global__ __launch_bounds__(256) void k_rows_f64_to_f32(const double *__restrict__ src, float *__restrict__ dst, long count)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}

template <class TR>
static int frame_pipeline_impl(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, TR *d_db, TR *d_post,
                               TR *d_row_lo, TR *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w,
                               int8_t *d_glyph, int8_t *d_colour, int16_t *d_pcm, float *d_db32, bool demodulate)
{
    constexpr bool F64 = sizeof(TR) == 8;
    if (n_frames < 0 || n_halo < 0 || window < 1 || disp_w < 1 || (display != 0 && display != 1) || (display == 1 && (disp_h < 1 || disp_h > 127)))
        return pss_fail(ctx, PSS_E_ARG, "pss_frame_pipeline: bad frame count, halo, window or display geometry");
    if (n < 8) return pss_fail(ctx, PSS_E_ARG, "pss_frame_pipeline: frames of fewer than 8 samples have no post-processed row to draw");
    if (n_frames > 0 && (!d_iq || (!d_db && !d_db32) || !d_row_lo || !d_row_hi || !d_glyph || (!d_colour && display == 0) || (!d_pcm && demodulate)))
        return pss_fail(ctx, PSS_E_ARG, "pss_frame_pipeline: null buffer");
    double *d_vals = nullptr;
    if (n_frames > 0 && !d_post) {
        const bool direct = pss_post_sel_serves(ctx, n, F64) && !(F64 && ctx->f64_plain);
        const size_t need = direct ? (size_t)n_frames * disp_w * sizeof(double) : (size_t)n_frames * (n - 4) * sizeof(TR);
        int rq = pss_ensure_buffer(ctx, &ctx->scratch_post, &ctx->scratch_post_bytes, need, "post-process scratch");
        if (rq) return rq;
        if (direct) d_vals = reinterpret_cast<double *>(ctx->scratch_post);
        else d_post = reinterpret_cast<TR *>(ctx->scratch_post);
    }
    // 1024-point frames, float64 rows, rows not materialised: the transform and the post-process are ONE kernel (pss_spec_post.h; option
    // "fuse_post" = 0: the two kernels) — the float64 rows never go through HBM unless the caller asks for them (d_db)
    bool fused = false;
    if constexpr (F64) fused = ctx->fuse_post && d_vals && pss_spec_post_serves(ctx, n);
    if (n_frames > 0 && !d_db && !fused) {     // float32 rows only, but this path needs the float64 rows in memory: the context's scratch
        int rq = pss_ensure_buffer(ctx, &ctx->scratch_db64, &ctx->scratch_db64_bytes, (size_t)n_frames * n * sizeof(TR), "float64 dB rows");
        if (rq) return rq;
        d_db = reinterpret_cast<TR *>(ctx->scratch_db64);
    }
    // compute_fft of every frame and the display chain behind it, on whichever stream it is queued
    auto spectrum_and_chain = [&]() -> int {
        if constexpr (F64) {
            if (fused)
                return pss_spec_post_chain(ctx, d_iq, n_frames, n, d_db32, d_db, d_row_lo, d_row_hi, n_halo, window, display, disp_h, disp_w, d_glyph,
                                           d_colour, d_vals);
        }
        int q = pipe_spectrum(ctx, d_iq, n_frames, n, d_db);     // compute_fft sees the samples as read (pyspecsdr.py:2275), not the corrected ones
        if (q) return q;
        if (d_db32 && n_frames > 0) {
            const long count = n_frames * (long)n;
            pss_kernel_begin(ctx, "k_rows_f64_to_f32");
            hipLaunchKernelGGL(k_rows_f64_to_f32, dim3((unsigned)((count + 255) / 256 < 16384 ? (count + 255) / 256 : 16384)), dim3(256), 0, PSS_STREAM(ctx),
                               reinterpret_cast<const double *>(d_db), d_db32, count);
            pss_kernel_end(ctx);
            q = pss_hip_check(ctx, hipGetLastError(), "k_rows_f64_to_f32 launch");
            if (q) return q;
        }
        if (d_vals) return pipe_chain_vals(ctx, d_db, n_frames, n, d_row_lo, d_row_hi, n_halo, window, display, disp_h, disp_w, d_glyph, d_colour, d_vals);
        q = pipe_post(ctx, d_db, n_frames, n, d_post, d_row_lo + n_halo, d_row_hi + n_halo);
        if (!q) q = pipe_lines(ctx, display, d_post, n_frames, n - 4, d_row_lo, d_row_hi, n_halo, window, disp_h, disp_w, d_glyph, d_colour);
        return q;
    };
    if (!demodulate) return spectrum_and_chain();
    // WFM: demodulate_signal's dispatcher semantics — the frames are IQ-corrected first (signal_processing.py:222-225); pss_demod's WFM
    // branch does it (wfm_correct): a scalars pre-pass in front of the forward kernel, alone on the machine
    const float *d_in = d_iq;
    PssFlagScope corr(ctx->wfm_correct, mode == PSS_MODE_WFM && n_frames > 0);
    if (mode != PSS_MODE_NFM && mode != PSS_MODE_WFM) {
        // AM / USB / LSB: neither demodulator has the two-phase shape of the FM paths, so the display chain simply runs on the side stream
        // beside the whole demodulator (AM's recurrence kernel keeps two thirds of the SIMDs busy with one wavefront each — the HBM-bound
        // chain fits in beside it).
        pss_time_begin(ctx);
        int r = pss_hip_check(ctx, hipEventRecord(ctx->ev_fork, ctx->stream), "hipEventRecord(fork)");
        if (!r) r = pss_hip_check(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0), "hipStreamWaitEvent(fork)");
        if (r) { pss_time_end(ctx); return r; }
        const int rd = pss_demod(ctx, mode, d_in, n_frames, n, fs, d_pcm, nullptr);   // main stream
        int rc;
        {
            PssStreamScope side(ctx->cur, ctx->stream2);
            rc = spectrum_and_chain();
        }
        int rj = pss_hip_check(ctx, hipEventRecord(ctx->ev_join, ctx->stream2), "hipEventRecord(join)");
        if (!rj) rj = pss_hip_check(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0), "hipStreamWaitEvent(join)");
        pss_time_end(ctx);
        return rd ? rd : (rc ? rc : rj);
    }
    pss_time_begin(ctx);
    // NFM and WFM.  Schedule: forward kernel (VALU-bound, fills the machine) ->
    //   { backward pass (latency-bound, one wavefront per SIMD)  ||  spectrum -> post-process -> display lines }.
    // (WFM until round 4: the chain beside the whole demodulator.  k_wfm_fwd's four workgroups per CU hold 150 of a CU's 160 KB of LDS, so the
    // spectrum kernel's 70 KB workgroups only ran as forward workgroups retired: 1.2 ms for a 0.17 ms kernel, and the chain was the critical path.)
    // Measured alternatives (rounds 2 - 4, NOTEBOOK.md R4-08 and A5; the code of those experiments left the tree in round 5): the spectrum in
    // front of the fork (+2 %); the whole display chain on the side stream from the start (-5 % when the forward kernel reaches the dispatcher
    // first, +8 % when it does not); the two streams on disjoint CU masks (hipExtStreamCreateWithCUMask, 128..240 of 256 CUs for the forward
    // kernel: +5 % at best — both halves of the step scale with the CUs they get); the spectrum kernel handing discriminator rows to the
    // forward kernel (+2 %); everything in order on one stream; forward -> spectrum -> { backward || post-process -> lines }.
    int r2;
    ctx->pending_bwd = nullptr;
    {
        PssFlagScope defer(ctx->defer_bwd, true);
        r2 = pss_demod(ctx, mode, d_in, n_frames, n, fs, d_pcm, nullptr);
    }
    int r = r2;
    if (ctx->pending_bwd) {
        auto bwd = ctx->pending_bwd;
        ctx->pending_bwd = nullptr;
        // the side stream ALWAYS waits for the main stream here: the chain reads d_iq and writes d_db / the scratch, all ordered on ctx->stream
        if (!r) r = pss_hip_check(ctx, hipEventRecord(ctx->ev_fork, ctx->stream), "hipEventRecord(fork)");
        if (!r) r = pss_hip_check(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0), "hipStreamWaitEvent(fork)");
        if (!r) {
            PssStreamScope side(ctx->cur, ctx->stream2);
            r = spectrum_and_chain();
        }
        const int rb = bwd();                      // main stream; launched whatever happened above (the PCM must be produced)
        int rj = pss_hip_check(ctx, hipEventRecord(ctx->ev_join, ctx->stream2), "hipEventRecord(join)");
        if (!rj) rj = pss_hip_check(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0), "hipStreamWaitEvent(join)");
        if (!r) r = rb ? rb : rj;
    } else if (!r) {
        r = spectrum_and_chain();                  // the demodulator took a path without a separate backward kernel
    }
    pss_time_end(ctx);
    return r;
}

// The iteration with the reference's own row type: float64 dB rows, float64 post-processed rows (d_post may be NULL: not materialised) and
// extremes, the waterfall line quantised from those — the cells the reference draws from this IQ, not those of the float32 rows.  The same
// schedule and the same kernel families as the float32 call (register transform with a float64 dB evaluation and 8-byte stores, register
// select on 64-bit keys); option "f64_plain" = 1: the plain round-3 kernels.  n: a power of two in [16, 65536].
extern "C" int pss_frame_pipeline_nfm_f64(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double fs, double *d_db, double *d_post,
                                          double *d_row_lo, double *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph,
                                          int8_t *d_colour, int16_t *d_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n < 16 || n > 65536 || (n & (n - 1))) return pss_fail(ctx, PSS_E_ARG, "pss_frame_pipeline_nfm_f64: n must be a power of two in [16, 65536]");
    return frame_pipeline<double>(ctx, PSS_MODE_NFM, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, n_halo, window, 0, 0, disp_w, d_glyph,
                                  d_colour, d_pcm);
}

// ... in ANY demodulation mode and for either batched display accumulator (pss_frame_pipeline's arguments, float64 rows)
extern "C" int pss_frame_pipeline_f64(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, double *d_db, double *d_post,
                                      double *d_row_lo, double *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w,
                                      int8_t *d_line_a, int8_t *d_line_b, int16_t *d_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (mode < PSS_MODE_NFM || mode > PSS_MODE_WFM) return pss_fail(ctx, PSS_E_ARG, "unknown demodulation mode");
    if (n < 16 || n > 65536 || (n & (n - 1))) return pss_fail(ctx, PSS_E_ARG, "pss_frame_pipeline_f64: n must be a power of two in [16, 65536]");
    return frame_pipeline<double>(ctx, mode, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, n_halo, window, display, disp_h, disp_w, d_line_a,
                                  d_line_b, d_pcm);
}

// The cell-exact iteration with the dB rows materialised as float32 (compute_fft's float64 value rounded once: the spectrum output's own contract),
// and as float64 too if d_db64 != NULL; float64 from the IQ to the cells either way.
extern "C" int pss_frame_pipeline_cells(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, float *d_db32, double *d_db64,
                                        double *d_row_lo, double *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w, int8_t *d_line_a,
                                        int8_t *d_line_b, int16_t *d_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (mode < PSS_MODE_NFM || mode > PSS_MODE_WFM) return pss_fail(ctx, PSS_E_ARG, "unknown demodulation mode");
    if (n < 16 || n > 65536 || (n & (n - 1))) return pss_fail(ctx, PSS_E_ARG, "pss_frame_pipeline_cells: n must be a power of two in [16, 65536]");
    if (n_frames > 0 && !d_db32) return pss_fail(ctx, PSS_E_ARG, "pss_frame_pipeline_cells: d_db32 is null");
    return frame_pipeline<double>(ctx, mode, d_iq, n_frames, n, fs, d_db64, nullptr, d_row_lo, d_row_hi, n_halo, window, display, disp_h, disp_w, d_line_a,
                                  d_line_b, d_pcm, d_db32);
}

// ... and its display half alone: compute_fft -> post-process -> display line of every frame, no demodulator
extern "C" int pss_spectrum_cells(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_db32, double *d_db64, double *d_row_lo,
                                  double *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w, int8_t *d_line_a, int8_t *d_line_b)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n < 16 || n > 65536 || (n & (n - 1))) return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_cells: n must be a power of two in [16, 65536]");
    if (n_frames > 0 && !d_db32 && !d_db64) return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_cells: no row buffer");
    return frame_pipeline<double>(ctx, PSS_MODE_NFM, d_iq, n_frames, n, 0.0, d_db64, nullptr, d_row_lo, d_row_hi, n_halo, window, display, disp_h, disp_w,
                                  d_line_a, d_line_b, nullptr, d_db32, false);
}

extern "C" int pss_frame_pipeline_nfm(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double fs, float *d_db, float *d_post,
                                      float *d_row_lo, float *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph,
                                      int8_t *d_colour, int16_t *d_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return frame_pipeline<float>(ctx, PSS_MODE_NFM, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, n_halo, window, 0, 0, disp_w, d_glyph,
                                 d_colour, d_pcm);
}

extern "C" int pss_frame_pipeline(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, float *d_db, float *d_post,
                                  float *d_row_lo, float *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w,
                                  int8_t *d_line_a, int8_t *d_line_b, int16_t *d_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (mode < PSS_MODE_NFM || mode > PSS_MODE_WFM) return pss_fail(ctx, PSS_E_ARG, "unknown demodulation mode");
    return frame_pipeline<float>(ctx, mode, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, n_halo, window, display, disp_h, disp_w, d_line_a,
                                 d_line_b, d_pcm);
}

extern "C" int pss_set_nfm_filters(pss_ctx *ctx, double fs, const double *taps65, const double *sos4x6, const double *zi4x2)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!taps65 || !sos4x6 || !zi4x2) return pss_fail(ctx, PSS_E_ARG, "null coefficient table");
    PssNfmFilt f;
    memcpy(f.taps, taps65, sizeof(f.taps));
    memcpy(f.sos, sos4x6, sizeof(f.sos));
    memcpy(f.zi, zi4x2, sizeof(f.zi));
    auto old = ctx->nfm.find(fs);
    if (old != ctx->nfm.end() && old->second.d_rev) {   // the device copy of the replaced taps may still be read by queued launches
        hipStreamSynchronize(ctx->stream);
        hipFree(old->second.d_rev);
    }
    ctx->nfm[fs] = f;
    return PSS_OK;
}

extern "C" int pss_set_ssb_taps(pss_ctx *ctx, double fs, const double *taps65)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!taps65) return pss_fail(ctx, PSS_E_ARG, "null coefficient table");
    std::array<double, 65> t;
    memcpy(t.data(), taps65, sizeof(double) * 65);
    ctx->ssb[fs] = t;
    return PSS_OK;
}

extern "C" int pss_get_nfm_filters(pss_ctx *ctx, double fs, double *taps65, double *sos4x6, double *zi4x2)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    PssNfmFilt *f;
    int r = nfm_filters(ctx, fs, &f);
    if (r) return r;
    if (taps65) memcpy(taps65, f->taps, sizeof(f->taps));
    if (sos4x6) memcpy(sos4x6, f->sos, sizeof(f->sos));
    if (zi4x2) memcpy(zi4x2, f->zi, sizeof(f->zi));
    return PSS_OK;
}

extern "C" int pss_get_ssb_taps(pss_ctx *ctx, double fs, double *taps65)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    double *t;
    int r = ssb_taps(ctx, fs, &t);
    if (r) return r;
    if (taps65) memcpy(taps65, t, sizeof(double) * 65);
    return PSS_OK;
}

#ifdef PSS_UBENCH   // role micro-benchmarks of the fused NFM forward kernel: variant builds only (tools/build_variant.py ubench -DPSS_UBENCH)
#include "pss_ubench.h"
#endif
