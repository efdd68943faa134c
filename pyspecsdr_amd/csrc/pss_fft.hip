// pss_fft.hip — spectrum-side kernels for gfx950: batched windowed FFT -> dB (compute_fft), the inline
// scanner slice, the caller's smoothing/median clamp, and the waterfall / persistence quantisers.
//
// Reference lines replaced: signal_processing.py:243-264 (compute_fft), pyspecsdr.py:2278-2283 (post),
// pyspecsdr.py:2542-2552 (scanner), pyspecsdr.py:1342-1406 / 1512-1564 (display accumulators).
//
// The FFT runs float64 butterflies in LDS (the reference is float64 pocketfft; float64 keeps every bin
// within 1e-4 relative of it even 100 dB below the peak), one workgroup per frame:
//   N <= 4096 : whole frame in LDS (16 B/point), in-place radix-4 DIF passes (+ one radix-2 when log2 N
//               is odd), digit-reversed read-out.
//   N  > 4096 : one radix-R DIF pre-pass (R = N/4096) folded into the load, then R sub-transforms of
//               4096 points, one after the other in the same LDS buffer.
// dB values are staged in LDS (4 B/point, when N <= 16384) so the global write is a coalesced float4 stream
// in fftshift order regardless of the digit reversal.  This path is tolerance-checked (1e-4 relative),
// so floating-point contraction is left on here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "pss_ctx.h"
#include "pss_fft_r16.h"
#include "pss_fft_xl.h"
#include "pss_post.h"
#include "pss_spec_post.h"
#include "pss_hilbert.h"
#include "pss_hilbert_pf.h"

namespace {

constexpr int TPB = 256;

// wave-uniform switches of the spectrum kernels (pss_fft_r16.h FLAG_*)
inline int spec_flags(const pss_ctx *ctx) { return (ctx->scan_exact ? pss_r16::FLAG_SCAN_EXACT : 0) | (ctx->db_exact ? pss_r16::FLAG_DB_EXACT : 0); }
constexpr int LOG_NSUB_MAX = 12;  // 4096 points * 16 B = 64 KiB of LDS
constexpr int STAGE_MAX_N = 16384;
constexpr int PT = 1024;  // k_post threads per row

__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// In-place decimation-in-frequency FFT of n = 2^logn points in LDS.  tw[k * twstride] = exp(-2 pi i k / n).
// Output X[k] lands at the digit-reversed position (see digit_reverse()).
__device__ void fft_dif_inplace(double2 *s, int logn, const double2 *__restrict__ tw, int twstride, int tid)
{
    const int n = 1 << logn;
    int L = n, logL = logn;
    if (logn & 1) {
        const int half = L >> 1;
        for (int b = tid; b < half; b += TPB) {
            double2 a0 = s[b], a1 = s[b + half];
            s[b] = cadd(a0, a1);
            s[b + half] = cmul(csub(a0, a1), tw[(size_t)b * twstride]);
        }
        __syncthreads();
        L = half;
        logL--;
    }
    while (L >= 4) {
        const int quarter = L >> 2, logq = logL - 2;
        const size_t twm = (size_t)twstride * (n >> logL);  // W_L^j = tw[j * twm]
        for (int b = tid; b < (n >> 2); b += TPB) {
            const int g = b >> logq, j = b & (quarter - 1);
            double2 *p = s + ((size_t)g << logL) + j;
            double2 a0 = p[0], a1 = p[quarter], a2 = p[2 * quarter], a3 = p[3 * quarter];
            double2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
            double2 t3 = make_double2(d.y, -d.x);  // -i * (a1 - a3)
            double2 y0 = cadd(t0, t2), y2 = csub(t0, t2), y1 = cadd(t1, t3), y3 = csub(t1, t3);
            if (quarter > 1) {
                y1 = cmul(y1, tw[(size_t)j * twm]);
                y2 = cmul(y2, tw[(size_t)2 * j * twm]);
                y3 = cmul(y3, tw[(size_t)3 * j * twm]);
            }
            p[0] = y0; p[quarter] = y1; p[2 * quarter] = y2; p[3 * quarter] = y3;
        }
        __syncthreads();
        L = quarter;
        logL -= 2;
    }
}

// Frequency index held at LDS position p after fft_dif_inplace (mixed radix 2,4,4,...).
__device__ __forceinline__ int digit_reverse(int p, int logn)
{
    int rem = logn, k = 0, mult = 0;
    if (logn & 1) {
        int m = p >> (rem - 1);
        p -= m << (rem - 1);
        k = m;
        mult = 1;
        rem -= 1;
    }
    while (rem >= 2) {
        int m = p >> (rem - 2);
        p -= m << (rem - 2);
        k += m << mult;
        mult += 2;
        rem -= 2;
    }
    return k;
}

using pss_r16::db_of;   // 10 log10(pw) to float64 accuracy, rounded once to float32 (pss_fft_r16.h)

// D64: the row type of the reference itself — db points at float64 rows, 10 log10(|X|^2 + 1e-10) evaluated in float64 as compute_fft does
// (np.abs = hypot, squared, + 1e-10, log10); for callers that need the display cells of the float64 rows (pss_spectrum_db_f64).
// IN64 (with D64): `iq` points at complex128 frames (interleaved float64): `samples * window` is then a float64 product of float64 samples (:247)
template <bool SCAN, bool EXACT = false, bool D64 = false, bool IN64 = false>
__global__ __launch_bounds__(TPB) void k_spectrum(const float2 *__restrict__ iq, float *__restrict__ db,
                                                  const double2 *__restrict__ tw, const double *__restrict__ win,
                                                  int N, int logNsub, int R, long n_frames, int staged,
                                                  float *__restrict__ peak, double *__restrict__ bw,
                                                  int *__restrict__ count, double bin_hz, int flags)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int Nsub = 1 << logNsub;
    double2 *s = reinterpret_cast<double2 *>(smem);
    float *stage = reinterpret_cast<float *>(smem + (size_t)Nsub * sizeof(double2));
    __shared__ float red_f[TPB / 64];
    __shared__ int red_i[TPB / 64];
    const int tid = threadIdx.x;
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *x = iq + (size_t)f * N * (IN64 ? 2 : 1);
        float *out = db ? db + (size_t)f * N : nullptr;
        auto sample = [&](int idx) -> double2 {
            if constexpr (IN64) return reinterpret_cast<const double2 *>(x)[idx];
            else { const float2 v = x[idx]; return make_double2((double)v.x, (double)v.y); }
        };
        for (int r = 0; r < R; r++) {
            for (int n = tid; n < Nsub; n += TPB) {
                double2 acc;
                if (R == 1) {
                    double2 v = sample(n);
                    double w = SCAN ? 1.0 : win[n];
                    acc = make_double2(v.x * w, v.y * w);
                } else {
                    acc = make_double2(0.0, 0.0);
                    for (int q = 0; q < R; q++) {
                        int idx = n + Nsub * q;
                        double2 v = sample(idx);
                        double w = SCAN ? 1.0 : win[idx];
                        double2 a = make_double2(v.x * w, v.y * w);
                        int e = (q * r) & (R - 1);  // W_R^{qr} = tw[((q r) mod R) * Nsub]
                        acc = cadd(acc, cmul(a, tw[(size_t)e * Nsub]));
                    }
                    acc = cmul(acc, tw[(size_t)n * r]);  // W_N^{n r}, n r < N
                }
                s[n] = acc;
            }
            __syncthreads();
            fft_dif_inplace(s, logNsub, tw, R, tid);
            for (int p = tid; p < Nsub; p += TPB) {
                double2 v = s[p];
                double pw = v.x * v.x + v.y * v.y + 1e-10;
                int k = R * digit_reverse(p, logNsub) + r;
                int o = (k + (N >> 1)) & (N - 1);  // fftshift
                if constexpr (D64) {
                    const double a = hypot(v.x, v.y);
                    reinterpret_cast<double *>(db)[(size_t)f * N + o] = 10.0 * log10(a * a + 1e-10);
                    continue;
                }
                float d;
                if constexpr (SCAN) d = (flags & pss_r16::FLAG_SCAN_EXACT) ? pss::scan_db_np(v.x, v.y) : pss_r16::db_of_fast(pw);
                else d = EXACT ? pss_r16::db_of_exact(pw) : pss_r16::db_of_fast(pw);   // scanner slice: NumPy's complex64 spectrum + float32 chain
                if (staged) stage[o] = d;
                else if (out) out[o] = d;
            }
            __syncthreads();
        }
        if (staged) {
            if (out) {
                float4 *o4 = reinterpret_cast<float4 *>(out);
                const float4 *s4 = reinterpret_cast<const float4 *>(stage);
                for (int i = tid; i < (N >> 2); i += TPB) o4[i] = s4[i];
            }
            if (SCAN) {
                float m = -INFINITY;
                for (int i = tid; i < N; i += TPB) m = fmaxf(m, stage[i]);
                for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
                if ((tid & 63) == 0) red_f[tid >> 6] = m;
                __syncthreads();
                m = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
                const float thr = m - 20.0f;  // power_db > (peak_power - 20), float32
                int c = 0;
                for (int i = tid; i < N; i += TPB) c += stage[i] > thr;
                for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
                if ((tid & 63) == 0) red_i[tid >> 6] = c;
                __syncthreads();
                if (tid == 0) {
                    c = red_i[0] + red_i[1] + red_i[2] + red_i[3];
                    peak[f] = m;
                    if (bw) bw[f] = (double)c * bin_hz;
                    if (count) count[f] = c;
                }
            }
            __syncthreads();
        }
    }
}

// pyspecsdr.py:2278-2283 — 5-tap moving average ('valid'), then everything below median-10 is raised to it.
// One workgroup per frame; the median comes from a bitonic sort of the smoothed row in LDS.
__global__ __launch_bounds__(PT) void k_post(const float *__restrict__ db, float *__restrict__ post, int N, int P,
                                              long n_frames)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *srt = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, m = N - 4, nthr = blockDim.x;
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float *row = db + (size_t)f * N;
        for (int i = tid; i < P; i += nthr) {
            float v = INFINITY;
            if (i < m) {
                double acc = 0.0;
                for (int k = 0; k < 5; k++) acc += (double)row[i + k] * 0.2;
                v = (float)acc;
            }
            srt[i] = v;
        }
        __syncthreads();
        for (int k = 2; k <= P; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < P; i += nthr) {
                    int ixj = i ^ j;
                    if (ixj > i) {
                        float a = srt[i], b = srt[ixj];
                        bool up = (i & k) == 0;
                        if ((a > b) == up) { srt[i] = b; srt[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        double med = (m & 1) ? (double)srt[m >> 1] : 0.5 * ((double)srt[(m >> 1) - 1] + (double)srt[m >> 1]);
        double thr = med - 10.0;
        __syncthreads();
        float *o = post + (size_t)f * m;
        for (int i = tid; i < m; i += nthr) {
            double acc = 0.0;
            for (int k = 0; k < 5; k++) acc += (double)row[i + k] * 0.2;
            o[i] = (float)(acc < thr ? thr : acc);
        }
        __syncthreads();
    }
}

// The same for rows that do not fit an LDS sort (N > 32768, up to the reference's 2^20-sample buffers): the two middle
// order statistics come from a most-significant-digit-first radix SELECT over the order-preserving integer image of the
// float32 smoothed values (4 passes of 8 bits per rank, 256-bin LDS histogram).  dB rows cluster in a few top-digit bins,
// so lanes that hit the same bin are combined with a ballot before the LDS atomic.
__device__ __forceinline__ unsigned f2ord(float v)
{
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

__global__ __launch_bounds__(1024) void k_post_select(const float *__restrict__ db, float *__restrict__ post, int N, long n_frames)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned sel_prefix, sel_k;
    const int tid = threadIdx.x, m = N - 4, nthr = blockDim.x;
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float *row = db + (size_t)f * N;
        auto smooth = [&](int i) {
            double acc = 0.0;
            for (int k = 0; k < 5; k++) acc += (double)row[i + k] * 0.2;
            return acc;
        };
        auto select = [&](unsigned k) {  // k-th smallest (0-based) of float32(smooth(i)), as its ordered-integer image
            unsigned prefix = 0, mask = 0;
            for (int shift = 24; shift >= 0; shift -= 8) {
                for (int b = tid; b < 256; b += nthr) hist[b] = 0;
                __syncthreads();
                for (int i0 = 0; i0 < m; i0 += nthr) {
                    const int i = i0 + tid;
                    bool live = i < m;
                    unsigned bin = 0;
                    if (live) {
                        const unsigned o = f2ord((float)smooth(i));
                        live = (o & mask) == prefix;
                        bin = (o >> shift) & 255u;
                    }
                    // combine equal bins of this wavefront: one atomic per distinct bin
                    unsigned long long todo = __ballot(live);
                    while (todo) {
                        const int leader = __ffsll((long long)todo) - 1;
                        const unsigned b0 = __shfl(bin, leader);
                        const unsigned long long same = __ballot(live && bin == b0) & todo;
                        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b0], (unsigned)__popcll(same));
                        todo &= ~same;
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    unsigned cum = 0, b = 0;
                    for (; b < 256; b++) {
                        if (cum + hist[b] > k) break;
                        cum += hist[b];
                    }
                    sel_prefix = prefix | (b << shift);
                    sel_k = k - cum;
                }
                __syncthreads();
                prefix = sel_prefix;
                k = sel_k;
                mask |= 255u << shift;
                __syncthreads();
            }
            return prefix;
        };
        double med;
        if (m & 1) med = (double)ord2f(select((unsigned)(m >> 1)));
        else med = 0.5 * ((double)ord2f(select((unsigned)(m >> 1) - 1)) + (double)ord2f(select((unsigned)(m >> 1))));
        const double thr = med - 10.0;
        float *o = post + (size_t)f * m;
        for (int i = tid; i < m; i += nthr) {
            const double acc = smooth(i);
            o[i] = (float)(acc < thr ? thr : acc);
        }
        __syncthreads();
    }
}

// The caller's post-process on float64 rows (the reference's own row type, pyspecsdr.py:2278-2283): np.convolve(row, ones(5) / 5,
// 'valid') — five products by 0.2 summed left to right, no fused multiply-adds —, np.median of the float64 values (mean of the two
// middle order statistics for an even count; NaN if the row holds one), everything below median - 10 raised to it.  Any row length:
// the order statistics come from a most-significant-byte-first radix select over the order-preserving integer image of the doubles.
__device__ __forceinline__ unsigned long long d2ord(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord2d(unsigned long long o)
{
    return __longlong_as_double((long long)((o >> 63) ? (o & 0x7fffffffffffffffull) : ~o));
}

__global__ __launch_bounds__(256) void k_post_f64(const double *__restrict__ db, double *__restrict__ post, int N, long n_frames)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sel_prefix;
    __shared__ unsigned sel_k;
    const int tid = threadIdx.x, m = N - 4;
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const double *row = db + (size_t)f * N;
        double *o = post + (size_t)f * m;
        int nan_here = 0;
        for (int i = tid; i < m; i += 256) {
            double acc = 0.0;
            {
#pragma clang fp contract(off)   // this unit is compiled with contraction on, and HIP's __dmul_rn / __dadd_rn are plain operators
                for (int k = 0; k < 5; k++) acc = acc + row[i + k] * 0.2;
            }
            o[i] = acc;
            nan_here |= acc != acc;
        }
        const int has_nan = __syncthreads_or(nan_here);      // also orders the stores above before the reads below
        auto select = [&](unsigned k) {                      // k-th smallest (0-based) of the smoothed row, as its ordered-integer image
            unsigned long long prefix = 0, mask = 0;
            for (int shift = 56; shift >= 0; shift -= 8) {
                hist[tid] = 0;
                __syncthreads();
                for (int i = tid; i < m; i += 256) {
                    const unsigned long long q = d2ord(o[i]);
                    if ((q & mask) == prefix) atomicAdd(&hist[(unsigned)(q >> shift) & 255u], 1u);
                }
                __syncthreads();
                if (tid == 0) {
                    unsigned cum = 0, b = 0;
                    for (; b < 255; b++) {
                        if (cum + hist[b] > k) break;
                        cum += hist[b];
                    }
                    sel_prefix = prefix | ((unsigned long long)b << shift);
                    sel_k = k - cum;
                }
                __syncthreads();
                prefix = sel_prefix;
                k = sel_k;
                mask |= 255ull << shift;
                __syncthreads();
            }
            return prefix;
        };
        if (!has_nan && m > 0) {
            double med;
            if (m & 1) med = ord2d(select((unsigned)(m >> 1)));
            else med = 0.5 * (ord2d(select((unsigned)(m >> 1) - 1)) + ord2d(select((unsigned)(m >> 1))));
            const double thr = med - 10.0;
            for (int i = tid; i < m; i += 256) {
                const double v = o[i];
                if (v < thr) o[i] = thr;
            }
        }
        __syncthreads();
    }
}

// np.interp(np.linspace(0, len-1, W), np.arange(len), row)[i]
template <class T>
__device__ __forceinline__ double interp_row(const T *row, int len, int W, int i)
{
    double stop = (double)(len - 1), x;
    if (W == 1) x = 0.0;
    else {
        double step = stop / (double)(W - 1);
        x = (i == W - 1) ? stop : (double)i * step;
    }
    if (x >= stop) return (double)row[len - 1];
    int j = (int)x;
    double slope = ((double)row[j + 1] - (double)row[j]) / ((double)(j + 1) - (double)j);
    return slope * (x - (double)j) + (double)row[j];
}

// draw_spectrogram (pyspecsdr.py:398-498) for one post-processed dB row per workgroup: noise floor = np.percentile(., 20)
// (method 'linear': the order statistics floor((n-1)*0.2) and the next one, found by an MSD radix select over the
// order-preserving 64-bit image of the values, then numpy's _lerp), display range (:424-427), clip + x**0.7 (:442-445),
// np.interp to the display width (:448-452), bar height int(value*H) and the glyph / colour of every cell (:455-490).
// glyph: 0 '.', 1 '-', 2 '=', 3 '#', 4 ' '; colour: curses pair (1 = cleared cell); -1: column not drawn (non-finite).

template <class T>
__global__ __launch_bounds__(1024) void k_spectrogram(const T *__restrict__ rows, long n_rows, int len, int disp_h, int disp_w,
                                                      int8_t *__restrict__ glyph, int8_t *__restrict__ colour,
                                                      double *__restrict__ range_out)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sel_prefix;
    __shared__ unsigned sel_k;
    __shared__ double red[16];
    __shared__ unsigned redn[16];
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (long f = blockIdx.x; f < n_rows; f += gridDim.x) {
        const T *row = rows + (size_t)f * len;
        int8_t *gl = glyph + (size_t)f * disp_h * disp_w, *co = colour + (size_t)f * disp_h * disp_w;
        // max and count of the finite values
        double mx = -INFINITY;
        unsigned cnt = 0;
        for (int i = tid; i < len; i += nthr) {
            const double v = (double)row[i];
            if (isfinite(v)) { mx = v > mx ? v : mx; cnt++; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(mx, off);
            mx = o > mx ? o : mx;
            cnt += __shfl_xor(cnt, off);
        }
        if ((tid & 63) == 0) { red[tid >> 6] = mx; redn[tid >> 6] = cnt; }
        __syncthreads();
        mx = red[0]; cnt = redn[0];
        for (int w = 1; w < nthr / 64; w++) { mx = red[w] > mx ? red[w] : mx; cnt += redn[w]; }
        __syncthreads();
        for (int i = tid; i < disp_h * disp_w; i += nthr) { gl[i] = -1; co[i] = -1; }
        if (cnt == 0) { __syncthreads(); continue; }
        auto select = [&](unsigned k) {  // k-th smallest finite value (0-based)
            unsigned long long prefix = 0, mask = 0;
            for (int shift = 56; shift >= 0; shift -= 8) {
                for (int b = tid; b < 256; b += nthr) hist[b] = 0;
                __syncthreads();
                for (int i0 = 0; i0 < len; i0 += nthr) {
                    const int i = i0 + tid;
                    bool live = false;
                    unsigned bin = 0;
                    if (i < len) {
                        const double v = (double)row[i];
                        const unsigned long long o = d2ord(v);
                        live = isfinite(v) && (o & mask) == prefix;
                        bin = (unsigned)(o >> shift) & 255u;
                    }
                    unsigned long long todo = __ballot(live);
                    while (todo) {  // one LDS atomic per distinct bin of the wavefront
                        const int leader = __ffsll((long long)todo) - 1;
                        const unsigned b0 = __shfl(bin, leader);
                        const unsigned long long same = __ballot(live && bin == b0) & todo;
                        if ((tid & 63) == leader) atomicAdd(&hist[b0], (unsigned)__popcll(same));
                        todo &= ~same;
                    }
                }
                __syncthreads();
                if (tid == 0) {
                    unsigned cum = 0, b = 0;
                    for (; b < 256; b++) {
                        if (cum + hist[b] > k) break;
                        cum += hist[b];
                    }
                    sel_prefix = prefix | ((unsigned long long)b << shift);
                    sel_k = k - cum;
                }
                __syncthreads();
                prefix = sel_prefix;
                k = sel_k;
                mask |= 255ull << shift;
                __syncthreads();
            }
            return ord2d(prefix);
        };
        // np.percentile(fin, 20), method 'linear'
        const double vi = (double)(cnt - 1) * 0.2;
        unsigned lo = (unsigned)floor(vi), hi = lo + 1;
        if (vi >= (double)(cnt - 1)) { lo = cnt - 1; hi = cnt - 1; }
        if (hi > cnt - 1) hi = cnt - 1;
        const double g = vi - floor(vi), a = select(lo), b = (hi == lo) ? a : select(hi), dba = b - a;
        double noise = a + dba * g;
        if (g >= 0.5) noise = b - dba * (1 - g);
        const double range = mx - noise;
        const double dmin = noise - (range * 0.1), dmax = mx + (range * 0.05);
        if (range_out && tid == 0) { range_out[2 * f] = dmin; range_out[2 * f + 1] = dmax; }
        auto shaped = [&](int j) {
            double v = ((double)row[j] - dmin) / (dmax - dmin);
            v = v < 0 ? 0 : (v > 1 ? 1 : v);
            return pow(v, 0.7);
        };
        for (int x = tid; x < disp_w; x += nthr) {
            // np.interp(np.linspace(0, len-1, W), np.arange(len), shaped)[x]
            const double stop = (double)(len - 1);
            double xp;
            if (disp_w == 1) xp = 0.0;
            else {
                const double step = stop / (double)(disp_w - 1);
                xp = (x == disp_w - 1) ? stop : (double)x * step;
            }
            double value;
            if (xp >= stop) value = shaped(len - 1);
            else {
                const int j = (int)xp;
                const double p0 = shaped(j), p1 = shaped(j + 1);
                const double slope = (p1 - p0) / ((double)(j + 1) - (double)j);
                value = slope * (xp - (double)j) + p0;
            }
            if (!isfinite(value)) continue;
            int height = (int)(value * disp_h);
            if (height > disp_h) height = disp_h;
            for (int y = 0; y < disp_h; y++) {
                int gch = 4, col = 1;
                if (y >= disp_h - height) {
                    const double rel = height > 0 ? (double)(y - (disp_h - height)) / (double)height : 0.0;
                    if (value > 0.8) { gch = rel > 0.5 ? 3 : 2; col = 14; }
                    else if (value > 0.4) { gch = rel > 0.5 ? 2 : 1; col = 13; }
                    else if (value > 0.2) { gch = rel > 0.5 ? 1 : 0; col = 12; }
                    else if (rel > 0.7) { gch = 0; col = 11; }
                    else { gch = 4; col = 10; }
                }
                gl[y * disp_w + x] = (int8_t)gch;
                co[y * disp_w + x] = (int8_t)col;
            }
        }
        __syncthreads();
    }
}

// MODE 0: waterfall (pyspecsdr.py:1342-1406)   MODE 1: persistence (pyspecsdr.py:1512-1564)
// MODE 2: gradient waterfall (pyspecsdr.py:1640-1716): zero-range guard, glyph = int(norm*8) into ' ._-=+*#@', colour int(norm*5)
// MODE 3: surface plot (pyspecsdr.py:1567-1616) of ONE row on the whole screen grid [disp_h][disp_w] = [max_h][max_w]
template <class T, int MODE>
__global__ __launch_bounds__(1024) void k_cells(const T *__restrict__ rows, int n_rows, int len, int disp_h, int disp_w,
                                               int8_t *__restrict__ glyph, int8_t *__restrict__ colour, int start, int cap)
{
    // rows live in a ring of `cap` rows; logical row i (0 = oldest) is physical row (start + i) mod cap
    constexpr int CT = 1024;  // threads: the min/max scan over the ring is latency-bound, so go wide and unroll
    __shared__ double red_lo[CT / 64], red_hi[CT / 64];
    const int tid = threadIdx.x;
    auto rowp = [&](int i) { return rows + (size_t)((start + i) % cap) * len; };
    double lo = INFINITY, hi = -INFINITY;
    const int total = n_rows * len;
#pragma unroll 8
    for (int e = tid; e < total; e += CT) {
        const int r = e / len, i = e - r * len;
        double v = (double)rowp(r)[i];
        if (isfinite(v)) { lo = fmin(lo, v); hi = fmax(hi, v); }
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, off));
        hi = fmax(hi, __shfl_xor(hi, off));
    }
    if ((tid & 63) == 0) { red_lo[tid >> 6] = lo; red_hi[tid >> 6] = hi; }
    __syncthreads();
    lo = red_lo[0]; hi = red_hi[0];
#pragma unroll
    for (int k = 1; k < CT / 64; k++) { lo = fmin(lo, red_lo[k]); hi = fmax(hi, red_hi[k]); }
    const int cells = disp_h * disp_w;
    if (MODE == 0 || MODE == 2) {
        double range2 = hi - lo;
        if (MODE == 2 && range2 == 0) range2 = 1;  // the plain waterfall has no zero-range guard
        for (int c = tid; c < cells; c += CT) {
            int y = c / disp_w, x = c - y * disp_w;
            int8_t g = -1, ci = -1;
            if (y < n_rows) {
                double v = interp_row(rowp(n_rows - 1 - y), len, disp_w, x);
                if (isfinite(v)) {
                    double nv = (v - lo) / range2;
                    ci = (int8_t)(int)(nv * 5);
                    if (MODE == 0) g = nv > 0.75 ? 3 : nv > 0.5 ? 2 : nv > 0.25 ? 1 : 0;
                    else g = (int8_t)(int)(nv * 8);
                }
            }
            glyph[c] = g;
            colour[c] = ci;
        }
    } else if (MODE == 3) {
        // '#' cells march up-left at 45 degrees from every column; later (x, y) overwrite earlier ones, so each cell keeps
        // the LARGEST (x, y) key that hits it (atomicMax on an int grid in global memory, then decoded to the colour pair)
        const double COS45 = 0x1.6a09e667f3bcdp-1, SIN45 = 0x1.6a09e667f3bccp-1;  // np.cos / np.sin(np.radians(45))
        double range = hi - lo;
        if (range == 0) range = 1;
        int *keys = reinterpret_cast<int *>(glyph);  // scratch: [disp_h][disp_w] ints supplied by the host wrapper
        for (int c = tid; c < cells; c += CT) keys[c] = -1;
        __syncthreads();
        const int w = disp_w - 8;
        const T *row = rowp(0);
        for (int x = tid; x < w; x += CT) {
            // np.interp over the NORMALISED row: normalisation is affine and applied per sample before interpolating
            const double stop = (double)(len - 1);
            double xp;
            if (w == 1) xp = 0.0;
            else {
                const double step = stop / (double)(w - 1);
                xp = (x == w - 1) ? stop : (double)x * step;
            }
            double value;
            if (xp >= stop) value = ((double)row[len - 1] - lo) / range;
            else {
                const int j = (int)xp;
                const double p0 = ((double)row[j] - lo) / range, p1 = ((double)row[j + 1] - lo) / range;
                const double slope = (p1 - p0) / ((double)(j + 1) - (double)j);
                value = slope * (xp - (double)j) + p0;
            }
            if (!isfinite(value)) continue;
            const int mag = (int)(value * 20);
            for (int y = 0; y < mag; y++) {
                const int sx = (int)((double)x - (double)y * COS45) + 8;
                const int sy = (int)((double)(disp_h - 2) - (double)y * SIN45);
                if (sx >= 0 && sx < disp_w && sy >= 2 && sy < disp_h - 1) atomicMax(&keys[sy * disp_w + sx], x * 32 + y);
            }
        }
        __syncthreads();
        for (int c = tid; c < cells; c += CT) colour[c] = keys[c] < 0 ? 0 : (int8_t)(1 + (keys[c] & 31) % 5);
    } else {
        double range = hi - lo;
        if (range == 0) range = 1;
        for (int c = tid; c < cells; c += CT) colour[c] = 0;
        __syncthreads();
        // traces are drawn oldest first and later ones overwrite: a column is owned by one thread
        for (int x = tid; x < disp_w; x += CT)
            for (int i = 0; i < n_rows; i++) {
                double alpha = pow(0.7, (double)(10 - i));
                int cp = (int)(1 + (5 * (1 - alpha)));
                double v = interp_row(rowp(i), len, disp_w, x);
                if (!isfinite(v)) continue;
                double nv = (v - lo) / range;
                int y = (int)((1 - nv) * (disp_h - 1));
                if (y >= 0 && y < disp_h) colour[y * disp_w + x] = (int8_t)cp;
            }
    }
}

int ilog2(int n)
{
    int l = 0;
    while ((1 << l) < n) l++;
    return l;
}

int grid_for(long n_frames, int per_cu)
{
    long g = n_frames;
    long cap = 256L * per_cu * 4;
    return (int)(g < cap ? g : cap);
}

template <int LOG_R3, bool SCAN>
int launch_r16(pss_ctx *ctx, const float *d_iq, long n_frames, float *d_db, const double2 *tw, const double *win,
               float *d_peak, double *d_bw, int32_t *d_count, double bin_hz)
{
    using C = pss_r16::Cfg<LOG_R3>;
    // component-wise LDS exchanges (half the LDS, twice the barriers) pay only at N = 256, where the plain kernel fits a
    // single 80 KB workgroup per CU: 0.29 -> 0.20 ms for 262144 frames; at 512...2048 they measured 20 % slower
    // next-frame prefetch, A/B in one process: N = 1024: 0.198 -> 0.185 ms (65536 frames), 2048: 0.234 -> 0.226 ms; 512: no change in round 2,
    // 0.163 -> 0.160 ms with the round-3 kernels (on since then);
    // 4096: 0.27 -> 0.35 ms (274 VGPRs: one wavefront per SIMD) in round 2; since the stage-2 twiddles are fetched from LDS a few products
    // ahead the kernel has 238 VGPRs with the prefetch, and with conflict-free exchanges it is the fastest 4096-point kernel:
    // 0.200 ms per 2^26 samples against 0.210 without the prefetch and 0.233 for k_spectrum_xl<0>; 256: the split kernel wins
    // "db_exact" (compute_fft rows only): its own instantiations (244 VGPRs with the prefetch: no spill)
    const bool exact = !SCAN && ctx->db_exact;
    int fpw = C::FPW;
#ifdef PSS_VARIANTS   // every combination, steered by the options "fft_split" / "fft_prefetch" / "fft_two_per_wg" (A/B builds)
    const bool split = ctx->fft_split >= 0 ? ctx->fft_split != 0 : LOG_R3 == 0;
    const bool prefetch = !split && (ctx->fft_prefetch >= 0 ? ctx->fft_prefetch != 0 : (LOG_R3 >= 1 && !(LOG_R3 == 4 && SCAN)));
    auto kern = exact ? (split ? pss_r16::k_spectrum_r16<LOG_R3, false, true, false, true>
                         : prefetch ? pss_r16::k_spectrum_r16<LOG_R3, false, false, true, true> : pss_r16::k_spectrum_r16<LOG_R3, false, false, false, true>)
                : split ? pss_r16::k_spectrum_r16<LOG_R3, SCAN, true, false>
                : prefetch ? pss_r16::k_spectrum_r16<LOG_R3, SCAN, false, true> : pss_r16::k_spectrum_r16<LOG_R3, SCAN, false, false>;
    size_t lds = split ? (size_t)C::FPW * C::EX * sizeof(double) + (size_t)C::TW2 * sizeof(double2) : C::LDS;
    if constexpr (LOG_R3 == 3 && !SCAN) {
        // N = 2048 (two wavefronts per frame): one frame per 128-thread workgroup
        if (!split && !ctx->fft_two_per_wg) {
            kern = exact ? (prefetch ? pss_r16::k_spectrum_r16<3, false, false, true, true, true> : pss_r16::k_spectrum_r16<3, false, false, false, true, true>)
                         : (prefetch ? pss_r16::k_spectrum_r16<3, false, false, true, false, true> : pss_r16::k_spectrum_r16<3, false, false, false, false, true>);
            fpw = 1;
            lds = (size_t)C::EX * sizeof(double2) + (size_t)C::TW2 * sizeof(double2);
        }
    }
#else                 // the product library carries the measured winner per length only (12 instantiations instead of 49)
    constexpr bool split = LOG_R3 == 0, prefetch = LOG_R3 >= 1 && !(LOG_R3 == 4 && SCAN), one = LOG_R3 == 3 && !SCAN;
    auto kern = exact ? pss_r16::k_spectrum_r16<LOG_R3, false, split, prefetch, true, one> : pss_r16::k_spectrum_r16<LOG_R3, SCAN, split, prefetch, false, one>;
    if (one) fpw = 1;
    const size_t lds = split ? (size_t)C::FPW * C::EX * sizeof(double) + (size_t)C::TW2 * sizeof(double2)
                             : (size_t)fpw * C::EX * sizeof(double2) + (size_t)C::TW2 * sizeof(double2);
#endif
    if (lds > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long groups = (n_frames + fpw - 1) / fpw;
    const int wg_threads = fpw * C::T;
    int per_cu = (int)((160 * 1024) / (lds + 256));
    const int vgpr_cap = (split ? 4 : 2) * 256 / wg_threads;   // 124 / 240 VGPRs: at most four / two 256-thread workgroups' worth of wavefronts per CU
    if (per_cu > vgpr_cap) per_cu = vgpr_cap;
    if (per_cu < 1) per_cu = 1;
    const long cap = 256L * per_cu * 2;
    const int grid = (int)(groups < cap ? groups : cap);
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_spectrum");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(wg_threads), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), d_db, tw,
                       win, n_frames, d_peak, d_bw, d_count, bin_hz, spec_flags(ctx));
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_spectrum_r16 launch");
}

// float64 dB rows (the reference's own row type) from the register kernels: the measured winner per length, float64 evaluation, 8-byte stores
template <int LOG_R3>
int launch_r16_f64(pss_ctx *ctx, const float *d_iq, long n_frames, double *d_db, const double2 *tw, const double *win)
{
    using C = pss_r16::Cfg<LOG_R3>;
    constexpr bool split = LOG_R3 == 0, prefetch = LOG_R3 >= 1, one = LOG_R3 == 3;
    auto kern = pss_r16::k_spectrum_r16<LOG_R3, false, split, prefetch, true, one, true>;
    const int fpw = one ? 1 : C::FPW;
    const size_t lds = split ? (size_t)C::FPW * C::EX * sizeof(double) + (size_t)C::TW2 * sizeof(double2)
                             : (size_t)fpw * C::EX * sizeof(double2) + (size_t)C::TW2 * sizeof(double2);
    if (lds > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long groups = (n_frames + fpw - 1) / fpw;
    const int wg_threads = fpw * C::T;
    int per_cu = (int)((160 * 1024) / (lds + 256));
    const int vgpr_cap = (split ? 4 : 2) * 256 / wg_threads;
    if (per_cu > vgpr_cap) per_cu = vgpr_cap;
    if (per_cu < 1) per_cu = 1;
    const long cap = 256L * per_cu * 2;
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_spectrum");
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups < cap ? groups : cap)), dim3(wg_threads), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq),
                       reinterpret_cast<float *>(d_db), tw, win, n_frames, (float *)nullptr, (double *)nullptr, (int *)nullptr, 0.0, spec_flags(ctx));
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_spectrum_r16 (float64 rows) launch");
}

template <bool SCAN>
int launch_spectrum(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, float *d_db, float *d_peak, double *d_bw,
                    int32_t *d_count, double bin_hz)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || (n_frames > 0 && !d_iq)) return pss_fail(ctx, PSS_E_ARG, "null iq / negative n_frames");
    if (n_fft < 16 || n_fft > (1 << 20) || (n_fft & (n_fft - 1)))
        return pss_fail(ctx, PSS_E_ARG, "n_fft must be a power of two in [16, 1048576]");
    if (n_frames == 0) return PSS_OK;
    const double2 *tw;
    const double *win;
    int r = pss_fft_tables(ctx, n_fft, &tw, &win);
    if (r) return r;
    // register-resident radix-16 kernel for the sizes that fit one workgroup (256 <= N <= 4096)
    switch (n_fft) {
    case 256: return launch_r16<0, SCAN>(ctx, d_iq, n_frames, d_db, tw, win, d_peak, d_bw, d_count, bin_hz);
    case 512: return launch_r16<1, SCAN>(ctx, d_iq, n_frames, d_db, tw, win, d_peak, d_bw, d_count, bin_hz);
    case 1024: return launch_r16<2, SCAN>(ctx, d_iq, n_frames, d_db, tw, win, d_peak, d_bw, d_count, bin_hz);
    case 2048: return launch_r16<3, SCAN>(ctx, d_iq, n_frames, d_db, tw, win, d_peak, d_bw, d_count, bin_hz);
    case 4096:
        // compute_fft rows: the three-stage kernel with complex exchanges (two workgroups per CU, next frame prefetched): 4.0 TB/s since
        // round 3 (conflict-free layouts, fused multiply-adds); scanner slices: the component-wise-exchange kernel below (128 VGPRs,
        // 35 KB LDS: four workgroups per CU), which hides the exact float32 chain's dependent arithmetic better.
        // "fft_xl4096" = 0 / 1 (variant builds): one of the two for both.
#ifdef PSS_VARIANTS
        if ((ctx->fft_xl4096 >= 0 ? !ctx->fft_xl4096 : !SCAN) || ctx->fft_big_scratch)
            return launch_r16<4, SCAN>(ctx, d_iq, n_frames, d_db, tw, win, d_peak, d_bw, d_count, bin_hz);
#else
        if constexpr (!SCAN) return launch_r16<4, SCAN>(ctx, d_iq, n_frames, d_db, tw, win, d_peak, d_bw, d_count, bin_hz);
#endif
        break;
    default: break;
    }
    if (((!SCAN && (n_fft == 8192 || n_fft == 16384)) || n_fft == 4096) && !ctx->fft_big_scratch) {
        // N = 16 x 16 x 16 x R4 in registers + LDS: the frame is read once, the dB row written once
        auto go = [&](auto kern, size_t lds, int threads, int per_cu) -> int {
            PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const long cap = 256L * per_cu;
            pss_time_begin(ctx);
            pss_kernel_begin(ctx, "k_spectrum");
            hipLaunchKernelGGL(kern, dim3((unsigned)(n_frames < cap ? n_frames : cap)), dim3(threads), lds, PSS_STREAM(ctx),
                               reinterpret_cast<const float2 *>(d_iq), d_db, tw, win, n_frames, d_peak, d_bw, d_count, bin_hz, spec_flags(ctx));
            pss_kernel_end(ctx);
            pss_time_end(ctx);
            return pss_hip_check(ctx, hipGetLastError(), "k_spectrum_xl launch");
        };
        if (!SCAN && ctx->db_exact) {
#ifdef PSS_VARIANTS
            if (n_fft == 4096) return go(pss_xl::k_spectrum_xl<0, true, false, true>, pss_xl::CfgX<0>::LDS, 256, 4);
#endif
            if (n_fft == 8192) return go(pss_xl::k_spectrum_xl<1, true, false, true>, pss_xl::CfgX<1>::LDS, 512, 2);
            return go(pss_xl::k_spectrum_xl<2, true, false, true>, pss_xl::CfgX<2>::LDS, 1024, 1);
        }
#ifdef PSS_VARIANTS
        if (n_fft == 4096) return go(pss_xl::k_spectrum_xl<0, !SCAN, SCAN>, pss_xl::CfgX<0>::LDS, 256, 4);
#else
        if constexpr (SCAN) return go(pss_xl::k_spectrum_xl<0, false, true>, pss_xl::CfgX<0>::LDS, 256, 4);
#endif
        if (n_fft == 8192) return go(pss_xl::k_spectrum_xl<1, true>, pss_xl::CfgX<1>::LDS, 512, 2);
        return go(pss_xl::k_spectrum_xl<2, true>, pss_xl::CfgX<2>::LDS, 1024, 1);
    }
    if (!SCAN && n_fft >= 8192 && n_fft <= 65536) {
        // N = R * 4096: radix-R pre-pass + register-resident 4096-point transforms, one workgroup per frame
        using C = pss_r16::Cfg<4>;
        const long cap = 512;  // workgroups (each owns N complex float64 of L2-resident scratch)
        const int grid = (int)(n_frames < cap ? n_frames : cap);
        r = pss_ensure_buffer(ctx, &ctx->scratch_fft, &ctx->scratch_fft_bytes, (size_t)grid * n_fft * sizeof(double2),
                              "spectrum scratch");
        if (r) return r;
        double2 *scr = reinterpret_cast<double2 *>(ctx->scratch_fft);
        void (*kern)(const float2 *, float *, const double2 *, const double *, long, double2 *, int) =
            n_fft == 8192 ? pss_r16::k_spectrum_r16_big<1, true>
            : n_fft == 16384 ? pss_r16::k_spectrum_r16_big<2, true>
            : n_fft == 32768 ? pss_r16::k_spectrum_r16_big<3, true> : pss_r16::k_spectrum_r16_big<4, true>;
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::LDS));
        pss_time_begin(ctx);
        pss_kernel_begin(ctx, "k_spectrum");
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::LDS, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), d_db,
                           tw, win, n_frames, scr, spec_flags(ctx));
        pss_kernel_end(ctx);
        pss_time_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "k_spectrum_r16_big launch");
    }
    if (!SCAN && n_fft >= (1 << 17)) {
        // N = 256 * NS: 256-point column transforms into a float64 scratch, then NS-point row transforms + dB
        const int NS = n_fft >> 8;
        r = pss_ensure_buffer(ctx, &ctx->scratch_fft, &ctx->scratch_fft_bytes, (size_t)n_frames * n_fft * sizeof(double2),
                              "spectrum scratch");
        if (r) return r;
        double2 *Y = reinterpret_cast<double2 *>(ctx->scratch_fft);
        using C0 = pss_r16::Cfg<0>;
        const size_t lds1 = (size_t)16 * (C0::EX + 1) * sizeof(double2);
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(pss_r16::k_huge_p1),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        const long total1 = n_frames * (NS / 16);
        pss_time_begin(ctx);
        pss_kernel_begin(ctx, "k_spectrum_p1");
        hipLaunchKernelGGL(pss_r16::k_huge_p1, dim3((unsigned)(total1 < 8192 ? total1 : 8192)), dim3(256), lds1, PSS_STREAM(ctx),
                           reinterpret_cast<const float2 *>(d_iq), tw, win, Y, NS, n_frames);
        pss_kernel_end(ctx);
        const long rows = n_frames * 256;
        auto launch2 = [&](auto kern, size_t lds2, int fpw) {
            if (lds2 > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            const long groups = rows / fpw;
            hipLaunchKernelGGL(kern, dim3((unsigned)(groups < 8192 ? groups : 8192)), dim3(256), lds2, PSS_STREAM(ctx), Y, d_db, tw, rows, spec_flags(ctx));
        };
        pss_kernel_begin(ctx, "k_spectrum");
        switch (NS) {
        case 512: launch2(pss_r16::k_huge_p2<1>, pss_r16::Cfg<1>::LDS, pss_r16::Cfg<1>::FPW); break;
        case 1024: launch2(pss_r16::k_huge_p2<2>, pss_r16::Cfg<2>::LDS, pss_r16::Cfg<2>::FPW); break;
        case 2048: launch2(pss_r16::k_huge_p2<3>, pss_r16::Cfg<3>::LDS, pss_r16::Cfg<3>::FPW); break;
        default: launch2(pss_r16::k_huge_p2<4>, pss_r16::Cfg<4>::LDS, pss_r16::Cfg<4>::FPW); break;
        }
        pss_kernel_end(ctx);
        pss_time_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "k_huge launch");
    }
    int logn = ilog2(n_fft);
    int logNsub = logn < LOG_NSUB_MAX ? logn : LOG_NSUB_MAX;
    int R = n_fft >> logNsub;
    int staged = n_fft <= STAGE_MAX_N;
    if (SCAN && !staged) return pss_fail(ctx, PSS_E_ARG, "scanner slices support n_fft <= 16384");
    size_t lds = ((size_t)1 << logNsub) * sizeof(double2) + (staged ? (size_t)n_fft * sizeof(float) : 0);
    auto kern = (!SCAN && ctx->db_exact) ? k_spectrum<false, true> : k_spectrum<SCAN>;
    if (lds > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / (lds + 64));
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    int grid = grid_for(n_frames, per_cu);
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_spectrum_generic");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(TPB), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), d_db, tw,
                       win, n_fft, logNsub, R, n_frames, staged, d_peak, d_bw, d_count, bin_hz, spec_flags(ctx));
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_spectrum launch");
}

}  // namespace

int pss_fft_tables(pss_ctx *ctx, int n, const double2 **tw, const double **win)
{
    auto it = ctx->tw.find(n);
    if (it == ctx->tw.end()) {
        std::vector<double2> h(n);
        std::vector<double> w(n);
        for (int k = 0; k < n; k++) {
            // exact octant symmetries keep the table accurate to < 1 ulp everywhere
            double ang = -2.0 * M_PI * (double)k / (double)n;
            h[k] = make_double2(std::cos(ang), std::sin(ang));
            // np.hamming(M): 0.54 - 0.46*cos(2 pi k / (M-1))  (signal_processing.py:246)
            w[k] = (n == 1) ? 1.0 : 0.54 - 0.46 * std::cos(2.0 * M_PI * (double)k / (double)(n - 1));
        }
        if (n >= 4) { h[n / 4] = make_double2(0.0, -1.0); h[n / 2] = make_double2(-1.0, 0.0); h[3 * n / 4] = make_double2(0.0, 1.0); }
        double2 *dtw = nullptr;
        double *dw = nullptr;
        PSS_HIP(ctx, hipMalloc(&dtw, sizeof(double2) * n));
        PSS_HIP(ctx, hipMalloc(&dw, sizeof(double) * n));
        PSS_HIP(ctx, hipMemcpy(dtw, h.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
        PSS_HIP(ctx, hipMemcpy(dw, w.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        ctx->tw[n] = dtw;
        ctx->win[n] = dw;
    }
    *tw = ctx->tw[n];
    *win = ctx->win[n];
    return PSS_OK;
}

// ---- frame lengths that are not a power of two: Bluestein's algorithm on the N = 256 * NS transform ------------------------
// compute_fft and the scanner's fft accept any length in the reference (NumPy's pocketfft); its reads are powers of two
// except the sweep driver's, int(0.1 * sample_rate) samples (pyspecsdr.py:1026, 240 000 at 2.4 MS/s).  With the chirp
// c[k] = exp(-i pi k^2 / n):  X[k] = c[k] * sum_j (x[j] c[j]) conj(c)[k - j], a circular convolution of length
// M = 2^m >= 2 n - 1 (M >= 2^17, the smallest size the two-pass transform handles; n <= 2^19):
//   A = FFT_M(x c, zero-padded);  B = FFT_M(conj(c) wrapped) (once per n);  conv = IFFT_M(A B) = conj(FFT_M(conj(A B))) / M.
// Four launches per batch chunk; the spectrum epilogue (chirp, |.|^2, dB, fftshift by n // 2) sits in the last store functor.
namespace {

struct BsLoadX {
    const float2 *iq; const double2 *chirp; const double *win; int n;
    __device__ double2 operator()(long f, size_t idx) const
    {
        if (idx >= (size_t)n) return make_double2(0.0, 0.0);
        const float2 s = iq[(size_t)f * n + idx];
        const double w = win ? win[idx] : 1.0;
        return pss_r16::cmul(make_double2((double)s.x * w, (double)s.y * w), chirp[idx]);
    }
};
struct BsLoadArr {
    const double2 *a;
    __device__ double2 operator()(long, size_t idx) const { return a[idx]; }
};
struct BsLoadConv {
    const double2 *A, *B; size_t M;
    __device__ double2 operator()(long f, size_t idx) const
    {
        const double2 p = pss_r16::cmul(A[(size_t)f * M + idx], B[idx]);
        return make_double2(p.x, -p.y);
    }
};
// the plain complex store; hilbert != 0: scipy.signal.hilbert's spectrum mask (1 at DC and Nyquist, 2 below, 0 above) and the conjugate that
// turns the next forward transform into the inverse one
struct BsStoreC {
    double2 *A; size_t M; int hilbert;
    __device__ void operator()(long f, size_t k, double2 X) const
    {
        if (hilbert) {
            const double h = (k == 0 || k == M / 2) ? 1.0 : (k < M / 2 ? 2.0 : 0.0);
            X = make_double2(X.x * h, -(X.y * h));
        }
        A[(size_t)f * M + k] = X;
    }
};
struct BsStoreDb {
    float *db; const double2 *chirp; int n, shift; double inv_m; bool np32; int flags;   // np32: the scanner's float32 chain (scan_db_np)
    __device__ void operator()(long f, size_t k, double2 X) const
    {
        if (k >= (size_t)n) return;
        const double2 z = pss_r16::cmul(make_double2(X.x * inv_m, -X.y * inv_m), chirp[k]);
        int o = (int)k + shift;                      // np.fft.fftshift: out[(k + n // 2) % n] = X[k]
        if (o >= n) o -= n;
        db[(size_t)f * n + o] = np32 ? pss::scan_db_np(z.x, z.y) : pss_r16::db_of(z.x * z.x + z.y * z.y + 1e-10, flags);
    }
};

template <class Load>
int bs_pass1(pss_ctx *ctx, Load load, const double2 *tw, double2 *Y, int NS, long n_frames)
{
    using C0 = pss_r16::Cfg<0>;
    const size_t lds1 = (size_t)16 * (C0::EX + 1) * sizeof(double2);
    auto kern = pss_r16::k_huge_p1_g<Load>;
    PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    const long total1 = n_frames * (NS / 16);
    hipLaunchKernelGGL(kern, dim3((unsigned)(total1 < 8192 ? total1 : 8192)), dim3(256), lds1, PSS_STREAM(ctx), load, tw, Y, NS, n_frames);
    return PSS_OK;
}
template <class Store>
int bs_pass2(pss_ctx *ctx, const double2 *Y, Store store, const double2 *tw, int NS, long n_frames)
{
    const long rows = n_frames * 256;
    auto go = [&](auto kern, size_t lds2, int fpw) -> int {
        if (lds2 > 64 * 1024)
            PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        const long groups = rows / fpw;
        hipLaunchKernelGGL(kern, dim3((unsigned)(groups < 8192 ? groups : 8192)), dim3(256), lds2, PSS_STREAM(ctx), Y, store, tw, rows);
        return PSS_OK;
    };
    switch (NS) {
    case 512: return go(pss_r16::k_huge_p2_g<1, Store>, pss_r16::Cfg<1>::LDS, pss_r16::Cfg<1>::FPW);
    case 1024: return go(pss_r16::k_huge_p2_g<2, Store>, pss_r16::Cfg<2>::LDS, pss_r16::Cfg<2>::FPW);
    case 2048: return go(pss_r16::k_huge_p2_g<3, Store>, pss_r16::Cfg<3>::LDS, pss_r16::Cfg<3>::FPW);
    default: return go(pss_r16::k_huge_p2_g<4, Store>, pss_r16::Cfg<4>::LDS, pss_r16::Cfg<4>::FPW);
    }
}

int bs_plan(pss_ctx *ctx, int n, pss_ctx::Bluestein **out)
{
    auto it = ctx->bs.find(n);
    if (it == ctx->bs.end()) {
        int M = 1 << 17;
        while (M < 2 * n - 1) M <<= 1;
        std::vector<double2> c(n), b((size_t)M, make_double2(0.0, 0.0));
        std::vector<double> w(n);
        for (int k = 0; k < n; k++) {
            const long long q = ((long long)k * k) % (2LL * n);          // k^2 mod 2n keeps the angle small: exp(-i pi q / n)
            const double ang = -M_PI * (double)q / (double)n;
            c[k] = make_double2(std::cos(ang), std::sin(ang));
            b[k] = make_double2(c[k].x, -c[k].y);
            if (k) b[(size_t)M - k] = b[k];
            w[k] = (n == 1) ? 1.0 : 0.54 - 0.46 * std::cos(2.0 * M_PI * (double)k / (double)(n - 1));   // np.hamming(n)
        }
        pss_ctx::Bluestein p{nullptr, nullptr, nullptr, M};
        double2 *d_b = nullptr;
        PSS_HIP(ctx, hipMalloc(&p.d_chirp, sizeof(double2) * n));
        PSS_HIP(ctx, hipMalloc(&p.d_B, sizeof(double2) * (size_t)M));
        PSS_HIP(ctx, hipMalloc(&p.d_win, sizeof(double) * n));
        PSS_HIP(ctx, hipMalloc(&d_b, sizeof(double2) * (size_t)M));
        PSS_HIP(ctx, hipMemcpy(p.d_chirp, c.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
        PSS_HIP(ctx, hipMemcpy(p.d_win, w.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        PSS_HIP(ctx, hipMemcpy(d_b, b.data(), sizeof(double2) * (size_t)M, hipMemcpyHostToDevice));
        const double2 *tw;
        const double *wm;
        int r = pss_fft_tables(ctx, M, &tw, &wm);
        if (!r) r = pss_ensure_buffer(ctx, &ctx->scratch_fft, &ctx->scratch_fft_bytes, (size_t)M * sizeof(double2), "spectrum scratch");
        if (!r) r = bs_pass1(ctx, BsLoadArr{d_b}, tw, reinterpret_cast<double2 *>(ctx->scratch_fft), M >> 8, 1);
        if (!r) r = bs_pass2(ctx, reinterpret_cast<const double2 *>(ctx->scratch_fft), BsStoreC{p.d_B, (size_t)M, 0}, tw, M >> 8, 1);
        hipStreamSynchronize(PSS_STREAM(ctx));
        hipFree(d_b);
        if (r) return r;
        ctx->bs[n] = p;
    }
    *out = &ctx->bs[n];
    return PSS_OK;
}

// dB rows (float32 [n_frames][n], DC-centred) of frames of any length 2 <= n <= 2^19; window: np.hamming (compute_fft) or none (scanner)
int bluestein_db(pss_ctx *ctx, const float *d_iq, long n_frames, int n, bool window, float *d_db)
{
    if (n < 2 || n > (1 << 19)) return pss_fail(ctx, PSS_E_ARG, "frame lengths that are not a power of two must lie in [2, 524288]");
    pss_ctx::Bluestein *p;
    int r = bs_plan(ctx, n, &p);
    if (r) return r;
    const size_t M = (size_t)p->M;
    const double2 *tw;
    const double *wm;
    r = pss_fft_tables(ctx, p->M, &tw, &wm);
    if (r) return r;
    long chunk = (long)(((size_t)1 << 30) / (M * sizeof(double2)));   // <= 1 GiB per scratch half
    if (chunk < 1) chunk = 1;
    if (chunk > n_frames) chunk = n_frames;
    r = pss_ensure_buffer(ctx, &ctx->scratch_fft, &ctx->scratch_fft_bytes, 2 * (size_t)chunk * M * sizeof(double2), "spectrum scratch");
    if (r) return r;
    double2 *Y = reinterpret_cast<double2 *>(ctx->scratch_fft), *A = Y + (size_t)chunk * M;
    const int NS = p->M >> 8;
    pss_time_begin(ctx);
    for (long f0 = 0; f0 < n_frames && !r; f0 += chunk) {
        const long nf = (n_frames - f0) < chunk ? (n_frames - f0) : chunk;
        const float2 *x = reinterpret_cast<const float2 *>(d_iq) + (size_t)f0 * n;
        pss_kernel_begin(ctx, "k_bluestein_p1");
        r = bs_pass1(ctx, BsLoadX{x, p->d_chirp, window ? p->d_win : nullptr, n}, tw, Y, NS, nf);
        pss_kernel_end(ctx);
        pss_kernel_begin(ctx, "k_bluestein_p2");
        if (!r) r = bs_pass2(ctx, Y, BsStoreC{A, M, 0}, tw, NS, nf);
        pss_kernel_end(ctx);
        pss_kernel_begin(ctx, "k_bluestein_p1");
        if (!r) r = bs_pass1(ctx, BsLoadConv{A, p->d_B, M}, tw, Y, NS, nf);
        pss_kernel_end(ctx);
        pss_kernel_begin(ctx, "k_bluestein_p2");
        if (!r) r = bs_pass2(ctx, Y, BsStoreDb{d_db + (size_t)f0 * n, p->d_chirp, n, n / 2, 1.0 / (double)M, !window && ctx->scan_exact, spec_flags(ctx)}, tw, NS, nf);
        pss_kernel_end(ctx);
    }
    pss_time_end(ctx);
    if (r) return r;
    return pss_hip_check(ctx, hipGetLastError(), "bluestein launch");
}

// per-row peak and number of bins above (mode 0) peak - 20 dB, pyspecsdr.py:2546-2552, or (mode 1) an absolute threshold,
// the sweep driver's pyspecsdr.py:1051-1057
__global__ __launch_bounds__(256) void k_scan_reduce(const float *__restrict__ db, int n, long n_rows, int mode, float threshold,
                                                     double bin_hz, float *__restrict__ peak, double *__restrict__ bw,
                                                     int32_t *__restrict__ count)
{
    __shared__ float redf[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x;
    for (long f = blockIdx.x; f < n_rows; f += gridDim.x) {
        const float *row = db + (size_t)f * n;
        float m = -INFINITY;
        for (int i = tid; i < n; i += 256) m = fmaxf(m, row[i]);
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((tid & 63) == 0) redf[tid >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
        const float thr = mode == 0 ? m - 20.0f : threshold;
        int c = 0;
        for (int i = tid; i < n; i += 256) c += row[i] > thr;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if ((tid & 63) == 0) redi[tid >> 6] = c;
        __syncthreads();
        if (tid == 0) {
            c = redi[0] + redi[1] + redi[2] + redi[3];
            if (peak) peak[f] = m;
            if (bw) bw[f] = (double)c * bin_hz;
            if (count) count[f] = c;
        }
        __syncthreads();
    }
}

int scan_reduce(pss_ctx *ctx, const float *d_db, int n, long n_rows, int mode, float threshold, double bin_hz, float *d_peak,
                double *d_bw, int32_t *d_count)
{
    pss_kernel_begin(ctx, "k_scan_reduce");
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)(n_rows < 16384 ? n_rows : 16384)), dim3(256), 0, PSS_STREAM(ctx), d_db, n, n_rows,
                       mode, threshold, bin_hz, d_peak, d_bw, d_count);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_scan_reduce launch");
}

bool is_pow2(int n) { return n > 0 && !(n & (n - 1)); }

}  // namespace

// scipy.signal.hilbert on n_rows float64 rows of n samples, n a power of two in [256, 16384] (the register transforms).
// out_mode 0: complex128 analytic signal to d_out; 1: real part to d_out (may be d_x itself) and, if d_maxbits is given,
// the row's max |real| as a double's bit pattern (atomicMax onto d_maxbits[row]; the caller clears it).
// ---- scipy.signal.hilbert for rows longer than one workgroup's registers (N = 2^15 .. 2^20: the reference's read buffers are
// (2 ** SAMPLES) * 256 samples, SAMPLES = 5 .. 12, default 7 = 32768: pyspecsdr.py:105, :2236) ----------------------------------
// Two transforms through a complex float64 spectrum Z in HBM: forward (real input) -> Z = conj(h X) (one-sided mask and the
// conjugate that turns the second forward transform into the inverse), forward again -> conj(.) / N.  N = 32768, 65536: the
// radix-R pre-pass + 4096-point kernel (k_big_g); N >= 2^17: the two-pass 256 x NS transform (bs_pass1 / bs_pass2).
namespace {

struct HilLoadReal {
    const double *x; size_t n;
    __device__ double2 operator()(long f, size_t idx) const { return make_double2(x[(size_t)f * n + idx], 0.0); }
};
struct HilStoreMasked {
    double2 *Z; size_t n;
    __device__ void operator()(long f, size_t k, double2 X) const
    {
        const double h = (k == 0 || k == n / 2) ? 1.0 : (k < n / 2 ? 2.0 : 0.0);
        Z[(size_t)f * n + k] = make_double2(X.x * h, -(X.y * h));
    }
};
struct HilLoadZ {
    const double2 *Z; size_t n;
    __device__ double2 operator()(long f, size_t idx) const { return Z[(size_t)f * n + idx]; }
};
// OUT 0: complex128 analytic signal; 1: real part + the row's peak |re| (bit pattern, atomicMax; one atomic per wavefront and call:
// every lane of the calling kernels is active here)
struct HilStoreOut {
    double *out; unsigned long long *mxbits; size_t n; double inv_n; int OUT;
    __device__ void operator()(long f, size_t k, double2 W) const
    {
        const double re = W.x * inv_n, im = -(W.y * inv_n);
        if (OUT == 0) reinterpret_cast<double2 *>(out)[(size_t)f * n + k] = make_double2(re, im);
        else {
            out[(size_t)f * n + k] = re;
            if (mxbits) {
                // lanes of one wavefront may belong to different rows in the two-pass kernels (16 frames' columns side by side is
                // not how they are called here: n_frames rows are walked one after the other) — still, reduce only among equal f
                double m = fabs(re);
                const int fi = (int)f, f0 = __shfl(fi, 0);
                if (__all(fi == f0)) {
                    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(m, off); m = (o != o || o > m) ? o : m; }
                    if ((threadIdx.x & 63) == 0) atomicMax(&mxbits[f], (unsigned long long)__double_as_longlong(m));
                } else {
                    atomicMax(&mxbits[f], (unsigned long long)__double_as_longlong(m));
                }
            }
        }
    }
};

template <int LOG_R, class Load, class Store>
int big_pass(pss_ctx *ctx, Load load, Store store, const double2 *tw, long n_rows, double2 *scr, int grid)
{
    using C = pss_r16::Cfg<4>;
    auto kern = pss_r16::k_big_g<LOG_R, Load, Store>;
    PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), C::LDS, PSS_STREAM(ctx), load, store, tw, n_rows, scr);
    return pss_hip_check(ctx, hipGetLastError(), "k_big_g launch");
}

int hilbert_long(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_out, int out_mode, unsigned long long *d_maxbits,
                 const double2 *tw)
{
    if (out_mode != 0 && out_mode != 1) return pss_fail(ctx, PSS_E_ARG, "pss_hilbert: rows longer than 16384 samples have no fused demodulator tail");
    const size_t N = (size_t)n;
    const int grid = (int)(n_rows < 512 ? n_rows : 512);
    // Z [n_rows][N] + the per-workgroup pre-pass scratch [grid][N] (N <= 65536) or the pass-1 output Y [n_rows][N] (N >= 2^17)
    const size_t szZ = (size_t)n_rows * N * sizeof(double2), szS = (n <= 65536 ? (size_t)grid : (size_t)n_rows) * N * sizeof(double2);
    // its own buffer, not scratch_fft: pss_frame_pipeline runs pss_spectrum_db (whose 2^15 .. 2^20-point kernels use scratch_fft) on the side
    // stream at the same time as the demodulator that calls this
    int r = pss_ensure_buffer(ctx, &ctx->scratch_hil, &ctx->scratch_hil_bytes, szZ + szS, "hilbert spectrum");
    if (r) return r;
    double2 *Z = reinterpret_cast<double2 *>(ctx->scratch_hil), *S = Z + (size_t)n_rows * N;
    const HilLoadReal lx{d_x, N};
    const HilStoreMasked sz{Z, N};
    const HilLoadZ lz{Z, N};
    const double inv_n = 1.0 / (double)n;
    pss_kernel_begin(ctx, "k_hilbert");
    auto second = [&](auto pass) -> int { return pass(HilStoreOut{d_out, out_mode == 0 ? nullptr : d_maxbits, N, inv_n, out_mode}); };
    if (n == 32768) {
        r = big_pass<3>(ctx, lx, sz, tw, n_rows, S, grid);
        if (!r) r = second([&](auto st) { return big_pass<3>(ctx, lz, st, tw, n_rows, S, grid); });
    } else if (n == 65536) {
        r = big_pass<4>(ctx, lx, sz, tw, n_rows, S, grid);
        if (!r) r = second([&](auto st) { return big_pass<4>(ctx, lz, st, tw, n_rows, S, grid); });
    } else {
        const int NS = n >> 8;
        r = bs_pass1(ctx, lx, tw, S, NS, n_rows);
        if (!r) r = bs_pass2(ctx, S, BsStoreC{Z, N, 1}, tw, NS, n_rows);
        if (!r) r = bs_pass1(ctx, lz, tw, S, NS, n_rows);
        if (!r) r = second([&](auto st) { return bs_pass2(ctx, S, st, tw, NS, n_rows); });
        if (!r) r = pss_hip_check(ctx, hipGetLastError(), "hilbert two-pass launch");
    }
    pss_kernel_end(ctx);
    return r;
}

}  // namespace

bool pss_hilbert_supported(int n) { return n >= 256 && n <= (1 << 20) && !(n & (n - 1)); }

// exp(2 pi i k / n), k < n, as pocketfft's plans tabulate it (sincos_2pibyn<double>: every value the product of two entries of two small
// tables — index & mask and index >> shift —, each entry cos / sin of a multiple of pi / (4 n) folded into the first octant; libm's cos
// and sin in double, as in SciPy's build).  One table per length serves the real forward and the complex inverse plan (pss_hilbert_pf.h).
static int pf_twiddles(pss_ctx *ctx, int n, const double2 **tw)
{
    auto it = ctx->tw_pf.find(n);
    if (it == ctx->tw_pf.end()) {
        const size_t N = (size_t)n;
        const long double pi = 3.141592653589793238462643383279502884197L;
        const double ang = (double)(0.25L * pi / (long double)n);
        auto calc = [&](size_t x) {
            x <<= 3;
            if (x < 4 * N) {
                if (x < 2 * N) {
                    if (x < N) return make_double2(std::cos((double)x * ang), std::sin((double)x * ang));
                    return make_double2(std::sin((double)(2 * N - x) * ang), std::cos((double)(2 * N - x) * ang));
                }
                x -= 2 * N;
                if (x < N) return make_double2(-std::sin((double)x * ang), std::cos((double)x * ang));
                return make_double2(-std::cos((double)(2 * N - x) * ang), std::sin((double)(2 * N - x) * ang));
            }
            x = 8 * N - x;
            if (x < 2 * N) {
                if (x < N) return make_double2(std::cos((double)x * ang), -std::sin((double)x * ang));
                return make_double2(std::sin((double)(2 * N - x) * ang), -std::cos((double)(2 * N - x) * ang));
            }
            x -= 6 * N;
            if (x < N) return make_double2(-std::sin((double)x * ang), -std::cos((double)x * ang));
            return make_double2(-std::cos((double)(2 * N - x) * ang), -std::sin((double)(2 * N - x) * ang));
        };
        const size_t nval = (N + 2) / 2;
        size_t shift = 1;
        while (((size_t)1 << shift) * ((size_t)1 << shift) < nval) ++shift;
        const size_t mask = ((size_t)1 << shift) - 1;
        std::vector<double2> v1(mask + 1), v2((nval + mask) / (mask + 1)), h(N);
        v1[0] = make_double2(1.0, 0.0);
        for (size_t i = 1; i < v1.size(); i++) v1[i] = calc(i);
        v2[0] = make_double2(1.0, 0.0);
        for (size_t i = 1; i < v2.size(); i++) v2[i] = calc(i * (mask + 1));
        for (size_t k = 0; k < N; k++) {
            const size_t idx = 2 * k <= N ? k : N - k;
            const double2 x1 = v1[idx & mask], x2 = v2[idx >> shift];
            const volatile double a = x1.x * x2.x, b = x1.y * x2.y, c = x1.x * x2.y, d = x1.y * x2.x;   // separate products: no contraction
            h[k] = make_double2(a - b, 2 * k <= N ? c + d : -(c + d));
        }
        double2 *d = nullptr;
        PSS_HIP(ctx, hipMalloc(&d, sizeof(double2) * N));
        PSS_HIP(ctx, hipMemcpy(d, h.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
        ctx->tw_pf[n] = d;
    }
    *tw = ctx->tw_pf[n];
    return PSS_OK;
}

// option "hilbert_exact" for rows of 2^15 .. 2^20 samples: the same passes through two scratch arrays per workgroup in global memory
static int hilbert_pf_long(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_out, int out_mode, unsigned long long *d_maxbits)
{
    if (out_mode != 0 && out_mode != 1) return pss_fail(ctx, PSS_E_ARG, "pss_hilbert: rows longer than 16384 samples have no fused demodulator tail");
    const double2 *tw;
    int r = pf_twiddles(ctx, n, &tw);
    if (r) return r;
    const size_t per_wg = (size_t)n * 2 * sizeof(double2);
    long grid = (long)(((size_t)1 << 30) / per_wg);       // at most 1 GiB of scratch
    grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
    if (grid > n_rows) grid = n_rows;
    r = pss_ensure_buffer(ctx, &ctx->scratch_hil, &ctx->scratch_hil_bytes, (size_t)grid * per_wg, "hilbert (exact) scratch");
    if (r) return r;
    pss_kernel_begin(ctx, "k_hilbert");
    hipLaunchKernelGGL(pss_pf::k_hilbert_pf_long, dim3((unsigned)grid), dim3(1024), 0, PSS_STREAM(ctx), d_x, d_out, tw, ilog2(n), n_rows, out_mode,
                       d_maxbits, reinterpret_cast<double *>(ctx->scratch_hil));
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_hilbert_pf_long launch");
}

// option "hilbert_exact": pocketfft's own butterfly order (pss_hilbert_pf.h), rows of 256 .. 16384 samples
static int hilbert_pf(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_out, int out_mode, unsigned long long *d_maxbits,
                      int16_t *d_pcm)
{
    const double2 *tw;
    int r = pf_twiddles(ctx, n, &tw);
    if (r) return r;
    const size_t lds = (size_t)n * sizeof(double);
    const int ept = n >= 1024 ? 16 : 8;
    auto kern = n >= 16384 ? pss_pf::k_hilbert_pf<16, 1024> : (ept == 16 ? pss_pf::k_hilbert_pf<16, 512> : pss_pf::k_hilbert_pf<8, 64>);
    if (lds > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / (lds + 512));
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    const long cap = 256L * per_cu;
    pss_kernel_begin(ctx, "k_hilbert");
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_rows < cap ? n_rows : cap)), dim3(n / ept), lds, PSS_STREAM(ctx), d_x, d_out, tw, ilog2(n), n_rows,
                       out_mode, d_maxbits, reinterpret_cast<unsigned *>(d_pcm));
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_hilbert_pf launch");
}

int pss_hilbert_rows(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_out, int out_mode, unsigned long long *d_maxbits,
                     int16_t *d_pcm)
{
    if (!pss_hilbert_supported(n)) return pss_fail(ctx, PSS_E_ARG, "pss_hilbert: the row length must be a power of two in [256, 1048576]");
    if (n_rows == 0) return PSS_OK;
    if (ctx->hilbert_exact && n <= 16384) return hilbert_pf(ctx, d_x, n_rows, n, d_out, out_mode, d_maxbits, d_pcm);
    if (ctx->hilbert_exact) return hilbert_pf_long(ctx, d_x, n_rows, n, d_out, out_mode, d_maxbits);
    const double2 *tw;
    const double *win;
    int r = pss_fft_tables(ctx, n, &tw, &win);
    if (r) return r;
    if (n > 16384) return hilbert_long(ctx, d_x, n_rows, n, d_out, out_mode, d_maxbits, tw);
    auto go = [&](auto kern, size_t lds, int threads, long groups, long cap) -> int {
        if (lds > 64 * 1024)
            PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        pss_kernel_begin(ctx, "k_hilbert");
        hipLaunchKernelGGL(kern, dim3((unsigned)(groups < cap ? groups : cap)), dim3(threads), lds, PSS_STREAM(ctx), d_x, d_out, tw, n_rows,
                           d_maxbits, reinterpret_cast<unsigned *>(d_pcm));
        pss_kernel_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "k_hilbert launch");
    };
    auto go_xl = [&](auto kern, size_t lds, int threads, long cap) -> int {
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        pss_kernel_begin(ctx, "k_hilbert");
        hipLaunchKernelGGL(kern, dim3((unsigned)(n_rows < cap ? n_rows : cap)), dim3(threads), lds, PSS_STREAM(ctx), d_x, d_out, tw, n_rows,
                           d_maxbits, reinterpret_cast<unsigned *>(d_pcm), out_mode);
        pss_kernel_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "k_hilbert_xl launch");
    };
#define HIL_R16(L)                                                                                                           \
    {                                                                                                                        \
        using C = pss_r16::Cfg<L>;                                                                                           \
        const long groups = (n_rows + C::FPW - 1) / C::FPW;                                                                  \
        int per_cu = (int)((160 * 1024) / (C::LDS + 256));                                                                   \
        if (per_cu > 2) per_cu = 2;                                                                                          \
        return out_mode == 0 ? go(pss_hil::k_hilbert_r16<L, 0>, C::LDS, 256, groups, 256L * per_cu * 2)                      \
             : out_mode == 1 ? go(pss_hil::k_hilbert_r16<L, 1>, C::LDS, 256, groups, 256L * per_cu * 2)                      \
                             : go(pss_hil::k_hilbert_r16<L, 2>, C::LDS, 256, groups, 256L * per_cu * 2);                     \
    }
    switch (n) {
    case 256: HIL_R16(0)
    case 512: HIL_R16(1)
    case 1024: HIL_R16(2)
    case 2048: HIL_R16(3)
    case 4096: HIL_R16(4)
    case 8192: return go_xl(pss_hil::k_hilbert_xl<1>, pss_xl::CfgX<1>::LDS, 512, 512);
    default: return go_xl(pss_hil::k_hilbert_xl<2>, pss_xl::CfgX<2>::LDS, 1024, 256);
    }
#undef HIL_R16
}

// demodulate_ssb for frames of 8192 / 16 384 samples in one kernel (pss_hilbert.h k_ssb_hilbert_xl): FIR (real part, zdot order) +
// hilbert() round trip + normalisation + PCM.  taps65: firwin's taps[0..64].
bool pss_ssb_fused_supported(int n) { return n == 8192 || n == 16384; }
int pss_ssb_hilbert_fused(pss_ctx *ctx, const float *d_iq, long n_rows, int n, const double *taps65, double *d_audio, int16_t *d_pcm)
{
    if (!pss_ssb_fused_supported(n)) return pss_fail(ctx, PSS_E_ARG, "fused SSB: frames of 8192 or 16384 samples");
    if (n_rows == 0) return PSS_OK;
    const double2 *tw;
    const double *win;
    int r = pss_fft_tables(ctx, n, &tw, &win);
    if (r) return r;
    pss_hil::SsbTaps tp;
    for (int j = 0; j < 72; j++) { tp.rev[j] = j < 65 ? taps65[64 - j] : 0.0; tp.fwd[j] = j < 65 ? taps65[j] : 0.0; }
    auto go = [&](auto kern, size_t lds, int threads, long cap) -> int {
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        pss_kernel_begin(ctx, "k_ssb_hilbert");
        hipLaunchKernelGGL(kern, dim3((unsigned)(n_rows < cap ? n_rows : cap)), dim3(threads), lds, PSS_STREAM(ctx),
                           reinterpret_cast<const float2 *>(d_iq), d_audio, tw, n_rows, reinterpret_cast<unsigned *>(d_pcm), tp);
        pss_kernel_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "k_ssb_hilbert launch");
    };
    if (ctx->ssb_rfft) {
        // hilbert() as a real transform pair (k_ssb_rfft): two half-length transforms, 256 / 512 threads per frame, 4 / 2 frames resident per CU
        const long ncu = ctx->n_cus > 0 ? ctx->n_cus : 256;
        return n == 8192 ? go(pss_hil::k_ssb_rfft<0>, pss_xl::CfgX<0>::LDS, 256, 4 * ncu) : go(pss_hil::k_ssb_rfft<1>, pss_xl::CfgX<1>::LDS, 512, 2 * ncu);
    }
    return n == 8192 ? go(pss_hil::k_ssb_hilbert_xl<1>, pss_hil::SsbLay<1>::LDS, 512, 512) : go(pss_hil::k_ssb_hilbert_xl<2>, pss_hil::SsbLay<2>::LDS, 1024, 256);
}

extern "C" int pss_hilbert(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_analytic)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_rows < 0 || (n_rows > 0 && (!d_x || !d_analytic))) return pss_fail(ctx, PSS_E_ARG, "pss_hilbert: bad argument");
    pss_time_begin(ctx);
    const int r = pss_hilbert_rows(ctx, d_x, n_rows, n, d_analytic, 0, nullptr, nullptr);
    pss_time_end(ctx);
    return r;
}

extern "C" int pss_spectrum_db(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, float *d_db)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_db && n_frames > 0) return pss_fail(ctx, PSS_E_ARG, "d_db is null");
    if (n_frames > 0 && d_iq && n_fft >= 2 && !(is_pow2(n_fft) && n_fft >= 16)) return bluestein_db(ctx, d_iq, n_frames, n_fft, true, d_db);
    return launch_spectrum<false>(ctx, d_iq, n_frames, n_fft, d_db, nullptr, nullptr, nullptr, 0.0);
}

// float64 dB rows, float64 post-processed rows: what the reference's display functions are fed (compute_fft returns float64).  For callers
// that want the display cells of the reference itself and not those of the float32 rows (the rows agree to 1e-7, a cell can differ where
// a value sits on a quantisation edge: <= 2e-3 of the cells); not a throughput path — one plain kernel each.
extern "C" int pss_spectrum_db_f64(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, double *d_db)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || (n_frames > 0 && (!d_iq || !d_db))) return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_db_f64: null buffer / negative n_frames");
    if (n_fft < 16 || n_fft > 65536 || (n_fft & (n_fft - 1)))
        return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_db_f64: n_fft must be a power of two in [16, 65536]");
    if (n_frames == 0) return PSS_OK;
    const double2 *tw;
    const double *win;
    int r = pss_fft_tables(ctx, n_fft, &tw, &win);
    if (r) return r;
    // 256 .. 4096 points: the register kernels (option "f64_plain" = 1: the generic LDS transform with hypot / log10, the round-3 path)
    if (!ctx->f64_plain) {
        switch (n_fft) {
        case 256: return launch_r16_f64<0>(ctx, d_iq, n_frames, d_db, tw, win);
        case 512: return launch_r16_f64<1>(ctx, d_iq, n_frames, d_db, tw, win);
        case 1024: return launch_r16_f64<2>(ctx, d_iq, n_frames, d_db, tw, win);
        case 2048: return launch_r16_f64<3>(ctx, d_iq, n_frames, d_db, tw, win);
        case 4096: return launch_r16_f64<4>(ctx, d_iq, n_frames, d_db, tw, win);
        default: break;
        }
    }
    const int logn = ilog2(n_fft), logNsub = logn < LOG_NSUB_MAX ? logn : LOG_NSUB_MAX;
    const size_t lds = ((size_t)1 << logNsub) * sizeof(double2);
    auto kern = k_spectrum<false, false, true>;
    if (lds > 64 * 1024) PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / (lds + 64));
    per_cu = per_cu > 8 ? 8 : per_cu;
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_spectrum_f64");
    hipLaunchKernelGGL(kern, dim3(grid_for(n_frames, per_cu)), dim3(TPB), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq),
                       reinterpret_cast<float *>(d_db), tw, win, n_fft, logNsub, n_fft >> logNsub, n_frames, 0, nullptr, nullptr, nullptr, 0.0, 0);
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_spectrum (float64 rows) launch");
}

// compute_fft of complex128 frames (signal_processing.py:243-264 handed a complex128 buffer: float64 from the window product on): d_iq interleaved
// float64 (re, im) [n_frames][n_fft], d_db float64 [n_frames][n_fft].  The plain LDS transform (a conformance path: the reference's SDR
// buffer is complex64); n_fft a power of two in [16, 65536].
extern "C" int pss_spectrum_db_c128(pss_ctx *ctx, const double *d_iq, long n_frames, int n_fft, double *d_db)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || (n_frames > 0 && (!d_iq || !d_db))) return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_db_c128: null buffer / negative n_frames");
    if (n_fft < 16 || n_fft > 65536 || (n_fft & (n_fft - 1)))
        return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_db_c128: n_fft must be a power of two in [16, 65536]");
    if (n_frames == 0) return PSS_OK;
    const double2 *tw;
    const double *win;
    int r = pss_fft_tables(ctx, n_fft, &tw, &win);
    if (r) return r;
    const int logn = ilog2(n_fft), logNsub = logn < LOG_NSUB_MAX ? logn : LOG_NSUB_MAX;
    const size_t lds = ((size_t)1 << logNsub) * sizeof(double2);
    auto kern = k_spectrum<false, false, true, true>;
    if (lds > 64 * 1024) PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = (int)((160 * 1024) / (lds + 64));
    per_cu = per_cu > 8 ? 8 : per_cu;
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_spectrum_c128");
    hipLaunchKernelGGL(kern, dim3(grid_for(n_frames, per_cu)), dim3(TPB), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq),
                       reinterpret_cast<float *>(d_db), tw, win, n_fft, logNsub, n_fft >> logNsub, n_frames, 0, nullptr, nullptr, nullptr, 0.0, 0);
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_spectrum (complex128 frames) launch");
}

// dB rows of the scanner's unwindowed fft into d_db or, when the caller wants only the per-slice numbers, into scratch
static int scan_rows(pss_ctx *ctx, const float *d_iq, long n_slices, int n, float *d_db, const float **rows)
{
    if (!d_db) {
        int r = pss_ensure_buffer(ctx, &ctx->scratch_scan, &ctx->scratch_scan_bytes, (size_t)n_slices * n * sizeof(float), "scan rows");
        if (r) return r;
        d_db = reinterpret_cast<float *>(ctx->scratch_scan);
    }
    *rows = d_db;
    if (is_pow2(n) && n >= 16 && n <= 16384) {
        // the one-kernel scanner path also produces its peak-relative numbers; they land in a throw-away buffer here
        int r = pss_ensure_buffer(ctx, &ctx->scratch_pk, &ctx->scratch_pk_bytes, (size_t)n_slices * sizeof(float), "scan peaks");
        if (r) return r;
        return launch_spectrum<true>(ctx, d_iq, n_slices, n, d_db, reinterpret_cast<float *>(ctx->scratch_pk), nullptr, nullptr, 0.0);
    }
    return bluestein_db(ctx, d_iq, n_slices, n, false, d_db);
}

extern "C" int pss_scan(pss_ctx *ctx, const float *d_iq, long n_slices, int n_fft, double fs, float *d_db, float *d_peak,
                        double *d_bw, int32_t *d_count)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_peak) return pss_fail(ctx, PSS_E_ARG, "d_peak is null");
    if (n_slices > 0 && d_iq && n_fft >= 2 && !(is_pow2(n_fft) && n_fft >= 16 && n_fft <= 16384)) {
        const float *rows;
        int r = scan_rows(ctx, d_iq, n_slices, n_fft, d_db, &rows);
        if (r) return r;
        return scan_reduce(ctx, rows, n_fft, n_slices, 0, 0.0f, fs / (double)n_fft, d_peak, d_bw, d_count);
    }
    return launch_spectrum<true>(ctx, d_iq, n_slices, n_fft, d_db, d_peak, d_bw, d_count, fs / (double)n_fft);
}

extern "C" int pss_scan_threshold(pss_ctx *ctx, const float *d_iq, long n_slices, int n, double fs, double threshold_db, float *d_db,
                                  float *d_peak, double *d_bw, int32_t *d_count)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_slices < 0 || n < 2 || (n_slices > 0 && !d_iq)) return pss_fail(ctx, PSS_E_ARG, "pss_scan_threshold: bad argument");
    if (n_slices == 0) return PSS_OK;
    const float *rows;
    int r = scan_rows(ctx, d_iq, n_slices, n, d_db, &rows);
    if (r) return r;
    return scan_reduce(ctx, rows, n, n_slices, 1, (float)threshold_db, fs / (double)n, d_peak, d_bw, d_count);
}

namespace {

template <int EPL, int W, class TR>
int launch_post_sel(pss_ctx *ctx, const TR *d_db, long n_frames, int n_fft, TR *d_post, TR *d_lo, TR *d_hi, TR *d_thr, double *d_vals = nullptr,
                    int disp_w = 0)
{
    constexpr int T = 64 * W, RPW = W == 1 ? 4 : 1;
    const size_t lds = (size_t)RPW * (T + 1) * pss_post::PostCfg<EPL, TR>::S * sizeof(TR);
    // the exact-fit specialisation (no padding tests) only for the read-buffer lengths of the display path: 1024, 2048, 4096 points
    constexpr bool FULL_BUILT = W == 1 ? EPL >= 16 : (W == 4 && EPL == 16);
    auto kern = FULL_BUILT && n_fft == T * EPL ? pss_post::k_post_sel<EPL, W, FULL_BUILT, TR> : pss_post::k_post_sel<EPL, W, false, TR>;
    if (lds > 64 * 1024)
        PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long groups = (n_frames + RPW - 1) / RPW;
    long per_cu = (long)((160 * 1024) / (lds + 512));
    const long want = W == 1 ? 6 : (W <= 4 ? 4 : 1);
    per_cu = per_cu < 1 ? 1 : (per_cu > want ? want : per_cu);
    const long cap = 256L * per_cu;
    pss_kernel_begin(ctx, "k_post");
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups < cap ? groups : cap)), dim3(W == 1 ? 256 : 64 * W), lds, PSS_STREAM(ctx), d_db,
                       d_post, n_fft, n_frames, d_lo, d_hi, d_thr, d_vals, disp_w);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_post_sel launch");
}

// the register select kernel (k_post_sel) serves rows of a multiple of 4 points up to 32 772 (float64 rows: up to 16 388 — the staging
// buffer of a longer row does not fit the LDS); only it can leave the rows unwritten
bool post_sel_serves(const pss_ctx *ctx, int n_fft, bool f64 = false) { return !ctx->post_legacy && n_fft - 4 <= (f64 ? 16384 : 32768) && (n_fft & 3) == 0; }

// the register select over rows of either type: rows written (d_post) or not, thresholds (d_thr), extremes, resampled rows (d_vals)
template <class TR>
int post_sel_any(pss_ctx *ctx, const TR *d_db, long n_frames, int n_fft, TR *d_post, TR *d_lo, TR *d_hi, TR *d_thr, double *d_vals, int disp_w)
{
    const int m = n_fft - 4;
    if (m <= 256) return launch_post_sel<4, 1, TR>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    if (m <= 512) return launch_post_sel<8, 1, TR>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    if (m <= 1024) return launch_post_sel<16, 1, TR>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    if (m <= 2048) return launch_post_sel<32, 1, TR>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    if (m <= 4096) return launch_post_sel<16, 4, TR>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    if (m <= 8192) return launch_post_sel<32, 4, TR>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    if (m <= 16384) return launch_post_sel<16, 16, TR>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    if constexpr (sizeof(TR) == 4) return launch_post_sel<32, 16, TR>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    return pss_fail(ctx, PSS_E_ARG, "float64 rows longer than 16388 points: no register select");
}

// smoothing + median clamp of n_frames rows; d_lo / d_hi (both or neither) receive the finite extremes of every clamped row.
// d_post == nullptr (with d_thr, d_lo, d_hi; rows k_post_sel serves): the rows are not written, only their clamp thresholds
// float32(median - 10) and extremes — 12 bytes per row instead of 4 (n_fft - 4).
int spectrum_post(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_post, float *d_lo, float *d_hi, float *d_thr = nullptr,
                  double *d_vals = nullptr, int disp_w = 0)
{
    if (d_vals && !post_sel_serves(ctx, n_fft)) return pss_fail(ctx, PSS_E_ARG, "resampled rows: only from the register select kernel");
    if (n_frames < 0 || (n_frames > 0 && (!d_db || (!d_post && !(d_thr && d_lo))))) return pss_fail(ctx, PSS_E_ARG, "null pointer");
    if (!d_post && n_fft >= 8 && !post_sel_serves(ctx, n_fft))
        return pss_fail(ctx, PSS_E_ARG, "post-process without materialised rows: n_fft must be a multiple of 4 and at most 32772");
    if ((d_lo == nullptr) != (d_hi == nullptr)) return pss_fail(ctx, PSS_E_ARG, "row extremes: pass both arrays or neither");
    if (n_fft < 8 || n_fft > (1 << 20)) return pss_fail(ctx, PSS_E_ARG, "post-process supports 8 <= n_fft <= 1048576");
    if (n_frames == 0) return PSS_OK;
    const int m = n_fft - 4;
    int r = PSS_OK;
    pss_time_begin(ctx);
    if (post_sel_serves(ctx, n_fft)) {
        // register-resident binary-search select: one wavefront per row up to 2048 points, 4 / 16 wavefronts above
        r = post_sel_any<float>(ctx, d_db, n_frames, n_fft, d_post, d_lo, d_hi, d_thr, d_vals, disp_w);
    } else if (n_fft > ctx->post_sort_max) {
        // rows too long for registers / the LDS sort: MSD radix select with an LDS histogram
        pss_kernel_begin(ctx, "k_post");
        const int thr = n_fft <= 2048 ? 256 : 1024;
        hipLaunchKernelGGL(k_post_select, dim3((unsigned)(n_frames < 8192 ? n_frames : 8192)), dim3(thr), 0, PSS_STREAM(ctx), d_db,
                           d_post, n_fft, n_frames);
        pss_kernel_end(ctx);
        r = pss_hip_check(ctx, hipGetLastError(), "k_post_select launch");
    } else {
        // option "post_legacy": bitonic sort of the smoothed row in LDS (kept as an A/B reference)
        int P = 1;
        while (P < n_fft - 4) P <<= 1;
        const size_t lds = (size_t)P * sizeof(float);
        if (lds > 64 * 1024)
            PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_post), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds));
        int per_cu = (int)((160 * 1024) / (lds + 64));
        if (per_cu > 8) per_cu = 8;
        pss_kernel_begin(ctx, "k_post");
        const int pthreads = n_frames < 1024 ? PT : TPB;
        hipLaunchKernelGGL(k_post, dim3(grid_for(n_frames, per_cu)), dim3(pthreads), lds, PSS_STREAM(ctx), d_db, d_post, n_fft, P,
                           n_frames);
        pss_kernel_end(ctx);
        r = pss_hip_check(ctx, hipGetLastError(), "k_post launch");
    }
    if (!r && d_lo && (ctx->post_legacy || m > 32768 || (n_fft & 3) != 0)) {
        pss_kernel_begin(ctx, "k_row_extremes");
        hipLaunchKernelGGL(pss_post::k_row_extremes<float>, dim3((unsigned)((n_frames + 3) / 4 < 8192 ? (n_frames + 3) / 4 : 8192)),
                           dim3(256), 0, PSS_STREAM(ctx), d_post, n_frames, m, d_lo, d_hi);
        pss_kernel_end(ctx);
        r = pss_hip_check(ctx, hipGetLastError(), "k_row_extremes launch");
    }
    pss_time_end(ctx);
    return r;
}

template <class T>
int row_extremes(pss_ctx *ctx, const T *d_rows, long n_rows, int len, T *d_lo, T *d_hi)
{
    if (n_rows < 0 || len < 1 || (n_rows > 0 && (!d_rows || !d_lo || !d_hi))) return pss_fail(ctx, PSS_E_ARG, "bad row-extremes arguments");
    if (n_rows == 0) return PSS_OK;
    pss_kernel_begin(ctx, "k_row_extremes");
    hipLaunchKernelGGL(pss_post::k_row_extremes<T>, dim3((unsigned)((n_rows + 3) / 4 < 8192 ? (n_rows + 3) / 4 : 8192)), dim3(256), 0,
                       PSS_STREAM(ctx), d_rows, n_rows, len, d_lo, d_hi);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_row_extremes launch");
}

// MODE 0: waterfall line (glyph, colour); MODE 1: persistence trace (y).  d_lo / d_hi: [n_halo + n_frames] row extremes.
// d_thr != nullptr: d_post holds the dB rows (len + 4 points each), the post-processed rows are rebuilt per cell
// d_vals != nullptr: the rows already resampled to disp_w columns (k_post_sel's `vals`); d_post / d_thr are not read
template <class T, int MODE>
int display_rows(pss_ctx *ctx, const T *d_post, long n_frames, int len, const T *d_lo, const T *d_hi, int n_halo, int window,
                 int disp_h, int disp_w, int8_t *d_a, int8_t *d_b, const T *d_thr = nullptr, const double *d_vals = nullptr)
{
    if (n_frames < 0 || len < 2 || disp_w < 1 || disp_h < 1 || disp_h > 127 || window < 1 || n_halo < 0 ||
        (n_frames > 0 && ((!d_post && !d_vals) || !d_lo || !d_hi || !d_a || (MODE == 0 && !d_b))))
        return pss_fail(ctx, PSS_E_ARG, "bad display-rows arguments");
    if (n_frames == 0) return PSS_OK;
    pss_time_begin(ctx);
    if (d_vals) {
        // the batched steps: window extremes and the lines from the resampled rows in ONE launch
        constexpr int RPB = 32;
        const long groups = (n_frames + RPB - 1) / RPB;
        pss_kernel_begin(ctx, "k_disp_rows");
        hipLaunchKernelGGL((pss_post::k_disp_vals_win<MODE, T, RPB>), dim3((unsigned)(groups < 8192 ? groups : 8192)), dim3(256), 0, PSS_STREAM(ctx),
                           d_vals, d_lo, d_hi, n_frames, n_halo, window, disp_w, disp_h, d_a, d_b);
        pss_kernel_end(ctx);
        pss_time_end(ctx);
        return pss_hip_check(ctx, hipGetLastError(), "k_disp_vals_win launch");
    }
    int r = pss_ensure_buffer(ctx, &ctx->scratch_win, &ctx->scratch_win_bytes, (size_t)n_frames * 2 * sizeof(double), "window extremes");
    if (r) { pss_time_end(ctx); return r; }
    double *wlo = reinterpret_cast<double *>(ctx->scratch_win), *whi = wlo + n_frames;
    pss_kernel_begin(ctx, "k_slide_extremes");
    hipLaunchKernelGGL(pss_post::k_slide_extremes<T>, dim3((unsigned)((n_frames + 255) / 256 < 4096 ? (n_frames + 255) / 256 : 4096)),
                       dim3(256), 0, PSS_STREAM(ctx), d_lo, d_hi, n_frames, n_halo, window, wlo, whi);
    pss_kernel_end(ctx);
    const long cells = n_frames * disp_w;
    pss_kernel_begin(ctx, "k_disp_rows");
    const dim3 dgrid((unsigned)((cells + 255) / 256 < 16384 ? (cells + 255) / 256 : 16384));
    if (d_thr)
        hipLaunchKernelGGL((pss_post::k_disp_rows<T, MODE, true>), dgrid, dim3(256), 0, PSS_STREAM(ctx), d_post, wlo, whi, n_frames, len,
                           disp_w, disp_h, d_a, d_b, d_thr);
    else
        hipLaunchKernelGGL((pss_post::k_disp_rows<T, MODE>), dgrid, dim3(256), 0, PSS_STREAM(ctx), d_post, wlo, whi, n_frames, len, disp_w,
                           disp_h, d_a, d_b, (const T *)nullptr);
    pss_kernel_end(ctx);
    pss_time_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_disp_rows launch");
}

}  // namespace

// compute_fft + post-process (+ row extremes) of the same frames, as two launches.  d_lo / d_hi may both be NULL.
// (A fused kernel for 1024-point frames — the dB row handed from the transform's registers through LDS to the post-process — was built
// in round 2 and measured no faster inside a pipeline step: 0.52 ms against 0.30 + 0.19, the post-process then runs at the
// transform's two wavefronts per SIMD; removed in round 3.)
extern "C" int pss_spectrum_db_post(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, float *d_db, float *d_post,
                                    float *d_row_lo, float *d_row_hi)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || (n_frames > 0 && (!d_iq || !d_db || !d_post))) return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_db_post: null buffer");
    if ((d_row_lo == nullptr) != (d_row_hi == nullptr)) return pss_fail(ctx, PSS_E_ARG, "row extremes: pass both arrays or neither");
    if (n_frames == 0) return PSS_OK;
    int r = pss_spectrum_db(ctx, d_iq, n_frames, n_fft, d_db);
    if (!r) r = spectrum_post(ctx, d_db, n_frames, n_fft, d_post, d_row_lo, d_row_hi);
    return r;
}

extern "C" int pss_spectrum_post(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_post)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return spectrum_post(ctx, d_db, n_frames, n_fft, d_post, nullptr, nullptr);
}

extern "C" int pss_spectrum_post_extremes(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_post, float *d_row_lo,
                                          float *d_row_hi)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames > 0 && (!d_row_lo || !d_row_hi)) return pss_fail(ctx, PSS_E_ARG, "null row-extremes array");
    return spectrum_post(ctx, d_db, n_frames, n_fft, d_post, d_row_lo, d_row_hi);
}

extern "C" int pss_row_extremes(pss_ctx *ctx, const float *d_rows, long n_rows, int len, float *d_row_lo, float *d_row_hi)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return row_extremes<float>(ctx, d_rows, n_rows, len, d_row_lo, d_row_hi);
}
extern "C" int pss_row_extremes_f64(pss_ctx *ctx, const double *d_rows, long n_rows, int len, double *d_row_lo, double *d_row_hi)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return row_extremes<double>(ctx, d_rows, n_rows, len, d_row_lo, d_row_hi);
}

extern "C" int pss_spectrum_post_f64(pss_ctx *ctx, const double *d_db, long n_frames, int n_fft, double *d_post, double *d_row_lo, double *d_row_hi)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames < 0 || n_fft < 5 || (n_frames > 0 && (!d_db || !d_post))) return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_post_f64: bad argument");
    if ((d_row_lo == nullptr) != (d_row_hi == nullptr)) return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_post_f64: row_lo and row_hi go together");
    if (n_frames == 0) return PSS_OK;
    if (!ctx->f64_plain && n_fft >= 8 && post_sel_serves(ctx, n_fft, true)) {   // the register select on 64-bit keys (option "f64_plain" = 1: the radix-select kernel below)
        pss_time_begin(ctx);
        const int rq = post_sel_any<double>(ctx, d_db, n_frames, n_fft, d_post, d_row_lo, d_row_hi, nullptr, nullptr, 0);
        pss_time_end(ctx);
        return rq;
    }
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_post_f64");
    hipLaunchKernelGGL(k_post_f64, dim3((unsigned)(n_frames < 2048 ? n_frames : 2048)), dim3(256), 0, PSS_STREAM(ctx), d_db, d_post, n_fft, n_frames);
    pss_kernel_end(ctx);
    int r = pss_hip_check(ctx, hipGetLastError(), "k_post_f64 launch");
    if (!r && d_row_lo) r = row_extremes<double>(ctx, d_post, n_frames, n_fft - 4, d_row_lo, d_row_hi);
    pss_time_end(ctx);
    return r;
}

// The post-process WITHOUT writing the post-processed rows: per row the clamp threshold float32(median - 10) and the finite extremes of
// the clamped row (12 bytes).  pss_waterfall_rows_db / pss_persistence_rows_db rebuild the elements a display line needs from the dB rows.
extern "C" int pss_spectrum_post_thresholds(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_row_thr, float *d_row_lo,
                                            float *d_row_hi)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames > 0 && (!d_row_thr || !d_row_lo || !d_row_hi)) return pss_fail(ctx, PSS_E_ARG, "pss_spectrum_post_thresholds: null buffer");
    return spectrum_post(ctx, d_db, n_frames, n_fft, nullptr, d_row_lo, d_row_hi, d_row_thr);
}
extern "C" int pss_waterfall_rows_db(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, const float *d_row_thr, const float *d_row_lo,
                                     const float *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph, int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames > 0 && !d_row_thr) return pss_fail(ctx, PSS_E_ARG, "pss_waterfall_rows_db: null thresholds");
    return display_rows<float, 0>(ctx, d_db, n_frames, n_fft - 4, d_row_lo, d_row_hi, n_halo, window, 1, disp_w, d_glyph, d_colour, d_row_thr);
}
extern "C" int pss_persistence_rows_db(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, const float *d_row_thr,
                                       const float *d_row_lo, const float *d_row_hi, int n_halo, int window, int disp_h, int disp_w, int8_t *d_y)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (n_frames > 0 && !d_row_thr) return pss_fail(ctx, PSS_E_ARG, "pss_persistence_rows_db: null thresholds");
    return display_rows<float, 1>(ctx, d_db, n_frames, n_fft - 4, d_row_lo, d_row_hi, n_halo, window, disp_h, disp_w, d_y, nullptr, d_row_thr);
}

// The display chain of pss_frame_pipeline behind the dB rows, for rows of either type (pss_ctx.h): post-process WITHOUT materialised rows —
// thresholds, extremes and the rows resampled to the display width in ONE pass over the dB rows (k_post_sel's `vals`) —, then the sliding
// extremes and the line of every frame from the resampled values.  display 0: waterfall (a = glyph, b = colour), 1: persistence (a = y).
// Serves the lengths the register select serves (pss_post_sel_serves); d_vals: n_frames x disp_w doubles of scratch.
bool pss_post_sel_serves(const pss_ctx *ctx, int n_fft, bool f64) { return n_fft >= 8 && post_sel_serves(ctx, n_fft, f64); }
template <class TR>
static int chain_vals(pss_ctx *ctx, const TR *d_db, long n_frames, int n_fft, TR *d_lo, TR *d_hi, int n_halo, int window, int display, int disp_h,
                      int disp_w, int8_t *d_a, int8_t *d_b, double *d_vals)
{
    if (n_frames == 0) return PSS_OK;
    pss_time_begin(ctx);
    int r = post_sel_any<TR>(ctx, d_db, n_frames, n_fft, (TR *)nullptr, d_lo + n_halo, d_hi + n_halo, (TR *)nullptr, d_vals, disp_w);
    if (!r) r = display ? display_rows<TR, 1>(ctx, (const TR *)nullptr, n_frames, n_fft - 4, d_lo, d_hi, n_halo, window, disp_h, disp_w, d_a, nullptr, (const TR *)nullptr, d_vals)
                        : display_rows<TR, 0>(ctx, (const TR *)nullptr, n_frames, n_fft - 4, d_lo, d_hi, n_halo, window, 1, disp_w, d_a, d_b, (const TR *)nullptr, d_vals);
    pss_time_end(ctx);
    return r;
}
int pss_chain_vals_f32(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_lo, float *d_hi, int n_halo, int window, int display,
                       int disp_h, int disp_w, int8_t *d_a, int8_t *d_b, double *d_vals)
{
    return chain_vals<float>(ctx, d_db, n_frames, n_fft, d_lo, d_hi, n_halo, window, display, disp_h, disp_w, d_a, d_b, d_vals);
}
int pss_chain_vals_f64(pss_ctx *ctx, const double *d_db, long n_frames, int n_fft, double *d_lo, double *d_hi, int n_halo, int window, int display,
                       int disp_h, int disp_w, int8_t *d_a, int8_t *d_b, double *d_vals)
{
    return chain_vals<double>(ctx, d_db, n_frames, n_fft, d_lo, d_hi, n_halo, window, display, disp_h, disp_w, d_a, d_b, d_vals);
}

// The fused transform + post-process (pss_spec_post.h) and the lines from its resampled rows: compute_fft -> cells for 1024-point frames without
// the float64 rows going through HBM.  d_db32 / d_db64: the dB row as float32 and / or float64 (either may be NULL, not both).
bool pss_spec_post_serves(const pss_ctx *ctx, int n_fft) { return n_fft == 1024 && !ctx->f64_plain && !ctx->post_legacy; }
int pss_spec_post_chain(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, float *d_db32, double *d_db64, double *d_lo, double *d_hi,
                        int n_halo, int window, int display, int disp_h, int disp_w, int8_t *d_a, int8_t *d_b, double *d_vals)
{
    if (!pss_spec_post_serves(ctx, n_fft) || (!d_db32 && !d_db64)) return pss_fail(ctx, PSS_E_ARG, "fused transform + post-process: 1024-point frames, a row buffer");
    if (n_frames == 0) return PSS_OK;
    const double2 *tw;
    const double *win;
    int r = pss_fft_tables(ctx, n_fft, &tw, &win);
    if (r) return r;
    using C = pss_r16::Cfg<2>;
    auto kern = d_db32 ? (d_db64 ? pss_sp::k_spectrum_post<true, true> : pss_sp::k_spectrum_post<true, false>) : pss_sp::k_spectrum_post<false, true>;
#ifdef PSS_EXP_FUSE_LDS      // timing experiment (with PSS_EXP_FUSE_NOFFT: the staged rows only)
    const size_t lds = PSS_EXP_FUSE_LDS;
    const long cap = 256L * PSS_EXP_FUSE_WAVES * 2;
#else
    const size_t lds = (size_t)C::FPW * C::EX * sizeof(double2) + (size_t)C::TW2 * sizeof(double2);
    const long cap = 256L * 2 * 2;       // two 256-thread workgroups per CU (LDS, registers), two rounds
#endif
    if (lds > 64 * 1024) PSS_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long groups = (n_frames + C::FPW - 1) / C::FPW;
    pss_time_begin(ctx);
    pss_kernel_begin(ctx, "k_spectrum_post");
    hipLaunchKernelGGL(kern, dim3((unsigned)(groups < cap ? groups : cap)), dim3(256), lds, PSS_STREAM(ctx), reinterpret_cast<const float2 *>(d_iq), d_db32,
                       d_db64, tw, win, n_frames, d_lo + n_halo, d_hi + n_halo, d_vals, disp_w);
    pss_kernel_end(ctx);
    r = pss_hip_check(ctx, hipGetLastError(), "k_spectrum_post launch");
    if (!r) r = display ? display_rows<double, 1>(ctx, (const double *)nullptr, n_frames, n_fft - 4, d_lo, d_hi, n_halo, window, disp_h, disp_w, d_a, nullptr, (const double *)nullptr, d_vals)
                        : display_rows<double, 0>(ctx, (const double *)nullptr, n_frames, n_fft - 4, d_lo, d_hi, n_halo, window, 1, disp_w, d_a, d_b, (const double *)nullptr, d_vals);
    pss_time_end(ctx);
    return r;
}

extern "C" int pss_waterfall_rows(pss_ctx *ctx, const float *d_post, long n_frames, int len, const float *d_row_lo,
                                  const float *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph, int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return display_rows<float, 0>(ctx, d_post, n_frames, len, d_row_lo, d_row_hi, n_halo, window, 1, disp_w, d_glyph, d_colour);
}
extern "C" int pss_waterfall_rows_f64(pss_ctx *ctx, const double *d_post, long n_frames, int len, const double *d_row_lo,
                                      const double *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph, int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return display_rows<double, 0>(ctx, d_post, n_frames, len, d_row_lo, d_row_hi, n_halo, window, 1, disp_w, d_glyph, d_colour);
}
extern "C" int pss_persistence_rows(pss_ctx *ctx, const float *d_post, long n_frames, int len, const float *d_row_lo,
                                    const float *d_row_hi, int n_halo, int window, int disp_h, int disp_w, int8_t *d_y)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return display_rows<float, 1>(ctx, d_post, n_frames, len, d_row_lo, d_row_hi, n_halo, window, disp_h, disp_w, d_y, nullptr);
}
extern "C" int pss_persistence_rows_f64(pss_ctx *ctx, const double *d_post, long n_frames, int len, const double *d_row_lo,
                                        const double *d_row_hi, int n_halo, int window, int disp_h, int disp_w, int8_t *d_y)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return display_rows<double, 1>(ctx, d_post, n_frames, len, d_row_lo, d_row_hi, n_halo, window, disp_h, disp_w, d_y, nullptr);
}

extern "C" int pss_waterfall_cells(pss_ctx *ctx, const float *d_rows, int n_rows, int len, int disp_h, int disp_w,
                                   int8_t *d_glyph, int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_rows || !d_glyph || !d_colour || n_rows < 1 || len < 2 || disp_h < 1 || disp_w < 1)
        return pss_fail(ctx, PSS_E_ARG, "bad waterfall arguments");
    pss_kernel_begin(ctx, "k_cells");
    hipLaunchKernelGGL((k_cells<float, 0>), dim3(1), dim3(1024), 0, PSS_STREAM(ctx), d_rows, n_rows, len, disp_h, disp_w,
                       d_glyph, d_colour, 0, n_rows);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_cells launch");
}

extern "C" int pss_persistence_cells(pss_ctx *ctx, const float *d_rows, int n_rows, int len, int disp_h, int disp_w,
                                     int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_rows || !d_colour || n_rows < 1 || len < 2 || disp_h < 1 || disp_w < 1)
        return pss_fail(ctx, PSS_E_ARG, "bad persistence arguments");
    pss_kernel_begin(ctx, "k_cells");
    hipLaunchKernelGGL((k_cells<float, 1>), dim3(1), dim3(1024), 0, PSS_STREAM(ctx), d_rows, n_rows, len, disp_h, disp_w,
                       (int8_t *)nullptr, d_colour, 0, n_rows);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_cells launch");
}

// float64-row variants: the quantisers are integer-valued functions of float64 data in the reference, so the
// parity tests drive them with the reference's own float64 rows.
template <class T>
static int launch_spectrogram(pss_ctx *ctx, const T *d_rows, long n_rows, int len, int disp_h, int disp_w, int8_t *d_glyph,
                              int8_t *d_colour, double *d_range)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_rows || !d_glyph || !d_colour || n_rows < 0 || len < 2 || disp_h < 1 || disp_w < 1)
        return pss_fail(ctx, PSS_E_ARG, "bad spectrogram arguments");
    if (n_rows == 0) return PSS_OK;
    pss_kernel_begin(ctx, "k_spectrogram");
    hipLaunchKernelGGL(k_spectrogram<T>, dim3((unsigned)(n_rows < 4096 ? n_rows : 4096)), dim3(len <= 4096 ? 256 : 1024), 0,
                       PSS_STREAM(ctx), d_rows, n_rows, len, disp_h, disp_w, d_glyph, d_colour, d_range);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_spectrogram launch");
}

extern "C" int pss_spectrogram_cells(pss_ctx *ctx, const float *d_rows, long n_rows, int len, int disp_h, int disp_w,
                                     int8_t *d_glyph, int8_t *d_colour, double *d_range)
{
    return launch_spectrogram<float>(ctx, d_rows, n_rows, len, disp_h, disp_w, d_glyph, d_colour, d_range);
}

extern "C" int pss_spectrogram_cells_f64(pss_ctx *ctx, const double *d_rows, long n_rows, int len, int disp_h, int disp_w,
                                         int8_t *d_glyph, int8_t *d_colour, double *d_range)
{
    return launch_spectrogram<double>(ctx, d_rows, n_rows, len, disp_h, disp_w, d_glyph, d_colour, d_range);
}

// draw_vector_display (pyspecsdr.py:1718-1752): every IQ sample drops a '.' at (int(cx + i*scale), int(cy - q*scale)),
// float32 arithmetic as NumPy evaluates it.  grid[max_h][max_w] = 1 where a dot lands (cleared by the launcher).
__global__ __launch_bounds__(256) void k_vector(const float2 *__restrict__ iq, int n, int max_h, int max_w, int8_t *__restrict__ grid)
{
    const int cx = max_w / 2, cy = max_h / 2, scale = (max_w < max_h ? max_w : max_h) / 4;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const float2 v = iq[k];
        const float fx = __fadd_rn((float)cx, __fmul_rn(v.x, (float)scale)), fy = __fsub_rn((float)cy, __fmul_rn(v.y, (float)scale));
        if (!isfinite(fx) || !isfinite(fy)) continue;
        const int x = (int)fx, y = (int)fy;
        if (x >= 0 && x < max_w && y >= 0 && y < max_h) grid[y * max_w + x] = 1;
    }
}

extern "C" int pss_vector_cells(pss_ctx *ctx, const float *d_iq, int n, int max_h, int max_w, int8_t *d_grid)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_iq || !d_grid || n < 0 || max_h < 1 || max_w < 1) return pss_fail(ctx, PSS_E_ARG, "bad vector-display arguments");
    PSS_HIP(ctx, hipMemsetAsync(d_grid, 0, (size_t)max_h * max_w, PSS_STREAM(ctx)));
    if (n == 0) return PSS_OK;
    pss_kernel_begin(ctx, "k_vector");
    hipLaunchKernelGGL(k_vector, dim3((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024)), dim3(256), 0, PSS_STREAM(ctx),
                       reinterpret_cast<const float2 *>(d_iq), n, max_h, max_w, d_grid);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_vector launch");
}

template <class T>
static int launch_gradient(pss_ctx *ctx, const T *d_rows, int n_rows, int len, int disp_h, int disp_w, int8_t *d_glyph,
                           int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_rows || !d_glyph || !d_colour || n_rows < 1 || len < 2 || disp_h < 1 || disp_w < 1)
        return pss_fail(ctx, PSS_E_ARG, "bad waterfall arguments");
    pss_kernel_begin(ctx, "k_cells");
    hipLaunchKernelGGL((k_cells<T, 2>), dim3(1), dim3(1024), 0, PSS_STREAM(ctx), d_rows, n_rows, len, disp_h, disp_w, d_glyph,
                       d_colour, 0, n_rows);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_cells launch");
}
extern "C" int pss_gradient_cells(pss_ctx *ctx, const float *d_rows, int n_rows, int len, int disp_h, int disp_w,
                                  int8_t *d_glyph, int8_t *d_colour)
{
    return launch_gradient<float>(ctx, d_rows, n_rows, len, disp_h, disp_w, d_glyph, d_colour);
}
extern "C" int pss_gradient_cells_f64(pss_ctx *ctx, const double *d_rows, int n_rows, int len, int disp_h, int disp_w,
                                      int8_t *d_glyph, int8_t *d_colour)
{
    return launch_gradient<double>(ctx, d_rows, n_rows, len, disp_h, disp_w, d_glyph, d_colour);
}

template <class T>
static int launch_surface(pss_ctx *ctx, const T *d_row, int len, int max_h, int max_w, int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_row || !d_colour || len < 2 || max_h < 4 || max_w < 10) return pss_fail(ctx, PSS_E_ARG, "bad surface arguments");
    // one int key per screen cell, parked in the FFT scratch (the display path does not run beside a big-N spectrum)
    int r = pss_ensure_buffer(ctx, &ctx->scratch_fft, &ctx->scratch_fft_bytes, (size_t)max_h * max_w * sizeof(int), "surface keys");
    if (r) return r;
    pss_kernel_begin(ctx, "k_cells");
    hipLaunchKernelGGL((k_cells<T, 3>), dim3(1), dim3(1024), 0, PSS_STREAM(ctx), d_row, 1, len, max_h, max_w,
                       reinterpret_cast<int8_t *>(ctx->scratch_fft), d_colour, 0, 1);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_cells launch");
}
extern "C" int pss_surface_cells(pss_ctx *ctx, const float *d_row, int len, int max_h, int max_w, int8_t *d_colour)
{
    return launch_surface<float>(ctx, d_row, len, max_h, max_w, d_colour);
}
extern "C" int pss_surface_cells_f64(pss_ctx *ctx, const double *d_row, int len, int max_h, int max_w, int8_t *d_colour)
{
    return launch_surface<double>(ctx, d_row, len, max_h, max_w, d_colour);
}

extern "C" int pss_waterfall_cells_f64(pss_ctx *ctx, const double *d_rows, int n_rows, int len, int disp_h, int disp_w,
                                       int8_t *d_glyph, int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_rows || !d_glyph || !d_colour || n_rows < 1 || len < 2 || disp_h < 1 || disp_w < 1)
        return pss_fail(ctx, PSS_E_ARG, "bad waterfall arguments");
    pss_kernel_begin(ctx, "k_cells");
    hipLaunchKernelGGL((k_cells<double, 0>), dim3(1), dim3(1024), 0, PSS_STREAM(ctx), d_rows, n_rows, len, disp_h, disp_w,
                       d_glyph, d_colour, 0, n_rows);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_cells launch");
}

extern "C" int pss_persistence_cells_f64(pss_ctx *ctx, const double *d_rows, int n_rows, int len, int disp_h, int disp_w,
                                         int8_t *d_colour)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!d_rows || !d_colour || n_rows < 1 || len < 2 || disp_h < 1 || disp_w < 1)
        return pss_fail(ctx, PSS_E_ARG, "bad persistence arguments");
    pss_kernel_begin(ctx, "k_cells");
    hipLaunchKernelGGL((k_cells<double, 1>), dim3(1), dim3(1024), 0, PSS_STREAM(ctx), d_rows, n_rows, len, disp_h, disp_w,
                       (int8_t *)nullptr, d_colour, 0, n_rows);
    pss_kernel_end(ctx);
    return pss_hip_check(ctx, hipGetLastError(), "k_cells launch");
}

// ---- stateful display accumulators: the reference keeps WATERFALL_HISTORY (last 30 rows, pyspecsdr.py:130-131,
// :1351-1353) and PERSISTENCE_HISTORY (last 10 rows, :151-152, :1521-1523) as Python lists; here a device ring.
struct pss_ring {
    pss_ctx *ctx;
    float *d_rows;
    int cap, len, count, head;  // head = physical index of the oldest row
};

extern "C" int pss_ring_create(pss_ctx *ctx, int max_rows, int len, pss_ring **out)
{
    if (!ctx || !out) return PSS_E_ARG;
    PSS_GUARD(ctx);
    *out = nullptr;
    if (max_rows < 1 || len < 2) return pss_fail(ctx, PSS_E_ARG, "bad ring arguments");
    pss_ring *r = new pss_ring{ctx, nullptr, max_rows, len, 0, 0};
    hipError_t e = hipMalloc(&r->d_rows, sizeof(float) * (size_t)max_rows * len);
    if (e != hipSuccess) { delete r; return pss_fail(ctx, PSS_E_NOMEM, "ring hipMalloc failed"); }
    *out = r;
    return PSS_OK;
}

extern "C" void pss_ring_destroy(pss_ring *r)
{
    if (!r) return;
    PSS_GUARD(r->ctx);
    hipStreamSynchronize(PSS_STREAM(r->ctx));
    hipFree(r->d_rows);
    delete r;
}

extern "C" int pss_ring_count(pss_ring *r) { return r ? r->count : PSS_E_ARG; }

extern "C" int pss_ring_push(pss_ring *r, const float *d_row)
{
    if (!r || !d_row) return PSS_E_ARG;
    PSS_GUARD(r->ctx);
    int slot;
    if (r->count < r->cap) slot = (r->head + r->count++) % r->cap;
    else { slot = r->head; r->head = (r->head + 1) % r->cap; }  // history.pop(0)
    return pss_hip_check(r->ctx, hipMemcpyAsync(r->d_rows + (size_t)slot * r->len, d_row, sizeof(float) * r->len,
                                                hipMemcpyDeviceToDevice, PSS_STREAM(r->ctx)), "ring push");
}

extern "C" int pss_ring_waterfall(pss_ring *r, int disp_h, int disp_w, int8_t *d_glyph, int8_t *d_colour)
{
    if (!r || !d_glyph || !d_colour || disp_h < 1 || disp_w < 1 || r->count < 1) return PSS_E_ARG;
    PSS_GUARD(r->ctx);
    hipLaunchKernelGGL((k_cells<float, 0>), dim3(1), dim3(1024), 0, PSS_STREAM(r->ctx), r->d_rows, r->count, r->len, disp_h,
                       disp_w, d_glyph, d_colour, r->head, r->cap);
    return pss_hip_check(r->ctx, hipGetLastError(), "k_cells launch");
}

extern "C" int pss_ring_persistence(pss_ring *r, int disp_h, int disp_w, int8_t *d_colour)
{
    if (!r || !d_colour || disp_h < 1 || disp_w < 1 || r->count < 1) return PSS_E_ARG;
    PSS_GUARD(r->ctx);
    hipLaunchKernelGGL((k_cells<float, 1>), dim3(1), dim3(1024), 0, PSS_STREAM(r->ctx), r->d_rows, r->count, r->len, disp_h,
                       disp_w, (int8_t *)nullptr, d_colour, r->head, r->cap);
    return pss_hip_check(r->ctx, hipGetLastError(), "k_cells launch");
}
