// pss_hilbert_pf.h — scipy.signal.hilbert along rows, BIT FOR BIT (option "hilbert_exact"): the reference's SSB demodulator calls
// hilbert() (signal_processing.py:205, :210), SciPy runs it on pocketfft (scipy.fft: third-party, vendored in SciPy as
// scipy/_lib/pocketfft — not part of the reference tree), and a transform with any other butterfly order lands 1e-16 away from it.
// These kernels replay pocketfft's published algorithm for rows of N = 2^k samples, 256 <= N <= 16384 in LDS / registers (longer rows: below) (oracle/pss_pocketfft.c is the
// CPU restatement the tests compare with, itself equal to SciPy on every bit):
//   forward  : the REAL transform (rfftp) — radix-4 passes radf4 from ido = 1 upwards, a last radix-2 pass radf2 when k is odd — on the
//              row as N float64 in LDS, FFTPACK half-complex result r0, r1, i1, ..., r_{N/2};
//   mask     : h = 1 at bins 0 and N / 2, 2 below N / 2, 0 above, read straight out of the half-complex row;
//   inverse  : the complex transform (cfftp, backward) with factors 8, 8, ..., then 4, then 2 (the 2 swapped to the front), passes from
//              l1 = 1 upwards, then the scale 1 / N;
//   twiddles : exp(2 pi i k / N) as pocketfft tabulates it (sincos_2pibyn: the product of two table entries in double) — uploaded by the
//              host (pss_fft.hip, pf_twiddles), one table of N values serves every pass of both plans;
//   no fused multiply-add anywhere (SciPy's wheels are baseline x86-64).
// One workgroup of T = N / EPT threads per row (EPT values per thread: 8 below 1024 samples, 16 up to 8192, 32 at 16 384 — 512 threads
// with a 256-register budget each; 1024 threads of 128 registers spill the inverse's 2 EPT values + butterfly temporaries).  A pass is N / 8 units of
// eight float64 (forward) or N / ip butterflies of ip complex values (inverse); every thread gathers the inputs of its units into
// registers, the workgroup meets, the outputs are scattered — pocketfft's ping-pong between two arrays becomes one array and a barrier.
// The inverse holds the row in registers (2 EPT float64 per thread) and exchanges real and imaginary parts one after the other through
// the same N float64 of LDS, so that 16 384 samples (256 KB as complex values) fit the CU's 160 KB.
#pragma once
#include <hip/hip_runtime.h>

#include "pss_hilbert.h"

namespace pss_pf {

constexpr double HSQT2 = 0.707106781186547524400844362104849;

struct c2 { double r, i; };

template <int EPT, int MAXT>
__global__ __launch_bounds__(MAXT) void k_hilbert_pf(const double *x, double *out, const double2 *__restrict__ tw, int logn, long n_rows,
                                                      int out_mode, unsigned long long *__restrict__ mxbits, unsigned *__restrict__ pcm)
{
#pragma clang fp contract(off)   // this unit is compiled with contraction on; every product and sum below is pocketfft's own
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ double red[2][16];
    double *A = reinterpret_cast<double *>(smem);
    const int t = threadIdx.x, N = 1 << logn, T = N / EPT;
    constexpr int UP = EPT / 8;                      // forward units per thread
    const double fct = 1.0 / (double)N;
    // the inverse plan: log2 of the factors in order, log2(l1) before each pass
    const int n8 = logn / 3, rem = logn % 3, npass = n8 + (rem ? 1 : 0);
    for (long f = blockIdx.x; f < n_rows; f += gridDim.x) {
        const double *row = x + (size_t)f * N;
        __syncthreads();                             // the previous row's last gathers are done
#pragma unroll
        for (int q = 0; q < EPT; q++) A[t + T * q] = row[t + T * q];
        __syncthreads();
        // ---------------- forward: radf4 passes, ido = 1, 4, 16, ... ----------------
        for (int lido = 0; lido + 2 <= logn; lido += 2) {
            const int ido = 1 << lido, l1 = N >> (lido + 2);
            int tt = t;
            asm volatile("" : "+v"(tt));             // the thread index behind an opaque statement per pass: with the plain index the compiler
                                                     // computes every pass's per-lane addresses ahead of the loops and spills them
            double v[UP][8];
            int oa[UP][8];
            double2 w[UP][3];
            bool z[UP];
#pragma unroll
            for (int r = 0; r < UP; r++) {
                const int u = tt + T * r;
                int ia[8];
                if (lido == 0) {
                    // two butterflies without twiddles: k = u and k = u + N / 8
                    const int k1 = u, k2 = u + (N >> 3);
#pragma unroll
                    for (int j = 0; j < 4; j++) { ia[j] = k1 + l1 * j; ia[4 + j] = k2 + l1 * j; oa[r][j] = j + 4 * k1; oa[r][4 + j] = j + 4 * k2; }
                    z[r] = true;
                } else {
                    const int k = u >> (lido - 1), s = u & ((1 << (lido - 1)) - 1), i = 2 * s, ic = ido - i;
                    const int cb = ido * k, hb = 4 * ido * k;                          // CC(a, k, c) = a + cb + ido l1 c; CH(a, b, k) = a + ido b + hb
                    const int il = ido * l1;
                    z[r] = s == 0;
                    if (z[r]) {
#pragma unroll
                        for (int j = 0; j < 4; j++) { ia[j] = cb + il * j; ia[4 + j] = ido - 1 + cb + il * j; }
                        oa[r][0] = hb; oa[r][1] = ido - 1 + 3 * ido + hb; oa[r][2] = 2 * ido + hb; oa[r][3] = ido - 1 + ido + hb;
                        oa[r][4] = ido - 1 + hb; oa[r][5] = ido - 1 + 2 * ido + hb; oa[r][6] = 3 * ido + hb; oa[r][7] = ido + hb;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; j++) { ia[2 * j] = i - 1 + cb + il * j; ia[2 * j + 1] = i + cb + il * j; }
                        oa[r][0] = i - 1 + hb; oa[r][1] = ic - 1 + 3 * ido + hb; oa[r][2] = i + hb; oa[r][3] = ic + 3 * ido + hb;
                        oa[r][4] = i - 1 + 2 * ido + hb; oa[r][5] = ic - 1 + ido + hb; oa[r][6] = i + 2 * ido + hb; oa[r][7] = ic + ido + hb;
                    }
#pragma unroll
                    for (int j = 0; j < 3; j++) w[r][j] = tw[(size_t)(j + 1) * l1 * s];
                }
#pragma unroll
                for (int e = 0; e < 8; e++) v[r][e] = A[ia[e]];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < UP; r++) {
                double o[8];
                if (lido == 0) {
#pragma unroll
                    for (int h = 0; h < 8; h += 4) {
                        const double c0 = v[r][h], c1 = v[r][h + 1], c2_ = v[r][h + 2], c3 = v[r][h + 3];
                        const double tr1 = c3 + c1, tr2 = c0 + c2_;
                        o[h] = tr2 + tr1; o[h + 1] = c0 - c2_; o[h + 2] = c3 - c1; o[h + 3] = tr2 - tr1;
                    }
                } else if (z[r]) {
                    const double c0 = v[r][0], c1 = v[r][1], c2_ = v[r][2], c3 = v[r][3], d0 = v[r][4], d1 = v[r][5], d2 = v[r][6], d3 = v[r][7];
                    const double tr1 = c3 + c1, tr2 = c0 + c2_;
                    const double ti1 = -HSQT2 * (d1 + d3), tb = HSQT2 * (d1 - d3);
                    o[0] = tr2 + tr1; o[1] = tr2 - tr1; o[2] = c3 - c1; o[3] = c0 - c2_;
                    o[4] = d0 + tb; o[5] = d0 - tb; o[6] = ti1 + d2; o[7] = ti1 - d2;
                } else {
                    const double a0 = v[r][0], b0 = v[r][1], a1 = v[r][2], b1 = v[r][3], a2 = v[r][4], b2 = v[r][5], a3 = v[r][6], b3 = v[r][7];
                    const double cr2 = w[r][0].x * a1 + w[r][0].y * b1, ci2 = w[r][0].x * b1 - w[r][0].y * a1;
                    const double cr3 = w[r][1].x * a2 + w[r][1].y * b2, ci3 = w[r][1].x * b2 - w[r][1].y * a2;
                    const double cr4 = w[r][2].x * a3 + w[r][2].y * b3, ci4 = w[r][2].x * b3 - w[r][2].y * a3;
                    const double tr1 = cr4 + cr2, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
                    const double tr2 = a0 + cr3, tr3 = a0 - cr3, ti2 = b0 + ci3, ti3 = b0 - ci3;
                    o[0] = tr2 + tr1; o[1] = tr2 - tr1; o[2] = ti1 + ti2; o[3] = ti1 - ti2;
                    o[4] = tr3 + ti4; o[5] = tr3 - ti4; o[6] = tr4 + ti3; o[7] = tr4 - ti3;
                }
#pragma unroll
                for (int e = 0; e < 8; e++) A[oa[r][e]] = o[e];
            }
            __syncthreads();
        }
        if (logn & 1) {
            // radf2, ido = N / 2, l1 = 1: slot s = i / 2 handles 4 float64; slots u and u + N / 8 per unit
            const int ido = N >> 1;
            int tt = t;
            asm volatile("" : "+v"(tt));
            double v[UP][2][4];
            int oa[UP][2][4];
            double2 w[UP][2];
#pragma unroll
            for (int r = 0; r < UP; r++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int s = tt + T * r + h * (N >> 3), i = 2 * s, ic = ido - i;
                    int ia[4];
                    if (s == 0) {
                        ia[0] = 0; ia[1] = ido; ia[2] = ido - 1; ia[3] = 2 * ido - 1;
                        oa[r][h][0] = 0; oa[r][h][1] = 2 * ido - 1; oa[r][h][2] = ido; oa[r][h][3] = ido - 1;
                    } else {
                        ia[0] = i - 1; ia[1] = i; ia[2] = i - 1 + ido; ia[3] = i + ido;
                        oa[r][h][0] = i - 1; oa[r][h][1] = ic - 1 + ido; oa[r][h][2] = i; oa[r][h][3] = ic + ido;
                    }
                    w[r][h] = tw[s];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[r][h][e] = A[ia[e]];
                }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < UP; r++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int s = tt + T * r + h * (N >> 3);
                    double o[4];
                    if (s == 0) {
                        o[0] = v[r][h][0] + v[r][h][1]; o[1] = v[r][h][0] - v[r][h][1]; o[2] = -v[r][h][3]; o[3] = v[r][h][2];
                    } else {
                        const double a0 = v[r][h][0], b0 = v[r][h][1], a1 = v[r][h][2], b1 = v[r][h][3];
                        const double tr2 = w[r][h].x * a1 + w[r][h].y * b1, ti2 = w[r][h].x * b1 - w[r][h].y * a1;
                        o[0] = a0 + tr2; o[1] = a0 - tr2; o[2] = ti2 + b0; o[3] = ti2 - b0;
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) A[oa[r][h][e]] = o[e];
                }
            __syncthreads();
        }
        // ---------------- inverse ----------------
        double re[EPT], im[EPT];
        int ll1 = 0;                                 // log2(l1)
        for (int pi = 0; pi < npass; pi++) {
            const int lip = rem == 1 ? (pi == 0 ? 1 : 3) : (pi < n8 ? 3 : 2);
            const int lido = logn - ll1 - lip, idom = (1 << lido) - 1;
            int tt = t;
            asm volatile("" : "+v"(tt));
            // slot e = r ip + j: butterfly u = t + T r = (k, i), input CC(i, j, k) = i + ido (j + ip k), output CH(i, k, j) = i + ido (k + l1 j)
            auto cc_addr = [&](int e) {
                const int r = e >> lip, j = e & ((1 << lip) - 1), u = tt + T * r, k = u >> lido, i = u & idom;
                return i + ((j + (k << lip)) << lido);
            };
            auto ch_addr = [&](int e) {
                const int r = e >> lip, j = e & ((1 << lip) - 1), u = tt + T * r, k = u >> lido, i = u & idom;
                return i + ((k + (j << ll1)) << lido);
            };
            if (pi == 0) {
                // the masked spectrum out of the half-complex row: X[0] = r0, X[e] = 2 (r_e, i_e) for 0 < e < N / 2, X[N / 2] = r_{N/2}, 0 above
#pragma unroll
                for (int e = 0; e < EPT; e++) {
                    const int b = cc_addr(e);
                    const bool low = b > 0 && b < (N >> 1);
                    // (scale factors instead of selects on the products: branch-free; x * 1.0 and x * 0.0 are what SciPy's X * h does)
                    const double sr = low ? 2.0 : ((b == 0 || b == (N >> 1)) ? 1.0 : 0.0), sq = low ? 2.0 : 0.0;
                    re[e] = A[low ? 2 * b - 1 : (b == 0 ? 0 : N - 1)] * sr;
                    im[e] = A[low ? 2 * b : 0] * sq;
                }
            }                                        // later passes: the exchange behind the previous pass has filled re[] / im[]
            auto bmul = [&](c2 v, int j, int i) {
#pragma clang fp contract(off)
                const double2 wj = tw[(size_t)(j * i) << ll1];
                c2 o;
                o.r = v.r * wj.x - v.i * wj.y;
                o.i = v.r * wj.y + v.i * wj.x;
                if (i == 0) o = v;
                return o;
            };
            auto add = [](c2 a, c2 b) {
#pragma clang fp contract(off)
                c2 o = {a.r + b.r, a.i + b.i}; return o; };
            auto sub = [](c2 a, c2 b) {
#pragma clang fp contract(off)
                c2 o = {a.r - b.r, a.i - b.i}; return o; };
            auto rot90 = [](c2 a) { c2 o = {-a.i, a.r}; return o; };
            auto rot45 = [](c2 a) {
#pragma clang fp contract(off)
                c2 o = {HSQT2 * (a.r - a.i), HSQT2 * (a.i + a.r)}; return o; };
            auto rot135 = [](c2 a) {
#pragma clang fp contract(off)
                c2 o = {HSQT2 * (-a.r - a.i), HSQT2 * (a.r - a.i)}; return o; };
            if (lip == 3) {
#pragma unroll
                for (int r = 0; r < EPT / 8; r++) {
                    const int i = (tt + T * r) & idom;
                    c2 c[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { c[j].r = re[8 * r + j]; c[j].i = im[8 * r + j]; }
                    c2 a1 = add(c[1], c[5]), a5 = sub(c[1], c[5]), a3 = add(c[3], c[7]), a7 = sub(c[3], c[7]);
                    { const c2 q = a1; a1 = add(a1, a3); a3 = sub(q, a3); }
                    a3 = rot90(a3);
                    a7 = rot90(a7);
                    { const c2 q = a5; a5 = add(a5, a7); a7 = sub(q, a7); }
                    a5 = rot45(a5);
                    a7 = rot135(a7);
                    c2 a0 = add(c[0], c[4]), a4 = sub(c[0], c[4]), a2 = add(c[2], c[6]), a6 = sub(c[2], c[6]);
                    { const c2 q = a0; a0 = add(a0, a2); a2 = sub(q, a2); }
                    a6 = rot90(a6);
                    { const c2 q = a4; a4 = add(a4, a6); a6 = sub(q, a6); }
                    c2 o[8];
                    o[0] = add(a0, a1);
                    o[4] = bmul(sub(a0, a1), 4, i);
                    o[2] = bmul(add(a2, a3), 2, i);
                    o[6] = bmul(sub(a2, a3), 6, i);
                    o[1] = bmul(add(a4, a5), 1, i);
                    o[5] = bmul(sub(a4, a5), 5, i);
                    o[3] = bmul(add(a6, a7), 3, i);
                    o[7] = bmul(sub(a6, a7), 7, i);
#pragma unroll
                    for (int j = 0; j < 8; j++) { re[8 * r + j] = o[j].r; im[8 * r + j] = o[j].i; }
                    __builtin_amdgcn_sched_barrier(0);   // one butterfly at a time: interleaved, their temporaries spill
                }
            } else if (lip == 2) {
#pragma unroll
                for (int r = 0; r < EPT / 4; r++) {
                    const int i = (tt + T * r) & idom;
                    c2 c[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) { c[j].r = re[4 * r + j]; c[j].i = im[4 * r + j]; }
                    const c2 t2 = add(c[0], c[2]), t1 = sub(c[0], c[2]), t3 = add(c[1], c[3]);
                    const c2 t4 = rot90(sub(c[1], c[3]));
                    c2 o[4];
                    o[0] = add(t2, t3);
                    o[1] = bmul(add(t1, t4), 1, i);
                    o[2] = bmul(sub(t2, t3), 2, i);
                    o[3] = bmul(sub(t1, t4), 3, i);
#pragma unroll
                    for (int j = 0; j < 4; j++) { re[4 * r + j] = o[j].r; im[4 * r + j] = o[j].i; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < EPT / 2; r++) {
                    const int i = (tt + T * r) & idom;
                    c2 c0 = {re[2 * r], im[2 * r]}, c1 = {re[2 * r + 1], im[2 * r + 1]};
                    const c2 o0 = add(c0, c1), o1 = bmul(sub(c0, c1), 1, i);
                    re[2 * r] = o0.r; im[2 * r] = o0.i; re[2 * r + 1] = o1.r; im[2 * r + 1] = o1.i;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (pi + 1 < npass) {
                // hand the pass's outputs to the next pass's layout: real parts, then imaginary parts, through A
                const int nlip = rem == 1 ? 3 : (pi + 1 < n8 ? 3 : 2), nll1 = ll1 + lip, nlido = logn - nll1 - nlip, nidom = (1 << nlido) - 1;
                auto ncc_addr = [&](int e) {
                    const int r = e >> nlip, j = e & ((1 << nlip) - 1), u = tt + T * r, k = u >> nlido, i = u & nidom;
                    return i + ((j + (k << nlip)) << nlido);
                };
                __syncthreads();                     // every gather out of A (the spectrum's, or the previous exchange's) is done
#pragma unroll
                for (int e = 0; e < EPT; e++) A[ch_addr(e)] = im[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < EPT; e++) im[e] = A[ncc_addr(e)];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < EPT; e++) A[ch_addr(e)] = re[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < EPT; e++) re[e] = A[ncc_addr(e)];
            }
            ll1 += lip;
        }
        // ---------------- output: slot e = r ip + j of the last pass holds sample (t + T r) + (N / ip) j ----------------
        const int lipl = rem == 1 ? 3 : (rem == 2 ? 2 : 3);
        double m = 0.0;
#pragma unroll
        for (int e = 0; e < EPT; e++) {
            const int r = e >> lipl, j = e & ((1 << lipl) - 1), idx = t + T * r + ((N >> lipl) * j);
            const double a = re[e] * fct, b = im[e] * fct;
            re[e] = a;
            if (out_mode == 0) reinterpret_cast<double2 *>(out)[(size_t)f * N + idx] = make_double2(a, b);
            else if (out_mode == 1) out[(size_t)f * N + idx] = a;
            m = pss_hil::nanmax(m, fabs(a));
        }
        if (out_mode != 0) {
            for (int off = 32; off > 0; off >>= 1)
                if (off < T) m = pss_hil::nanmax(m, __shfl_xor(m, off));
        }
        if (out_mode == 1 && mxbits && (t & 63) == 0) atomicMax(&mxbits[f], (unsigned long long)__double_as_longlong(m));
        if (out_mode == 2) {
            const int par = (int)(((f - blockIdx.x) / gridDim.x) & 1);
            if ((t & 63) == 0) red[par][t >> 6] = m;
            __syncthreads();
            m = red[par][0];
            for (int w2 = 1; w2 < (T + 63) / 64; w2++) m = pss_hil::nanmax(m, red[par][w2]);
#pragma unroll
            for (int e = 0; e < EPT; e++) {
                const int r = e >> lipl, j = e & ((1 << lipl) - 1), idx = t + T * r + ((N >> lipl) * j);
                const double a = pss_hil::normalise95(re[e], m);
                if (out) out[(size_t)f * N + idx] = a;
                if (pcm) pcm[(size_t)f * N + idx] = pss_hil::pcm_pair(a);
            }
        }
    }
}

}  // namespace pss_pf

// ---- rows longer than 16 384 samples (the reference's read buffers go up to 2^20): the same passes through global memory -------------
// One workgroup of 1024 threads per row, grid-strided; pocketfft's two ping-pong arrays are two scratch arrays per workgroup (real
// float64 for the forward transform, complex128 for the inverse — the real ones alias the first complex one), a workgroup barrier
// between passes.  No register residency, no LDS: an exactness path (L2-resident up to ~2^17 samples), not a throughput one.
namespace pss_pf {

__global__ __launch_bounds__(1024) void k_hilbert_pf_long(const double *x, double *out, const double2 *__restrict__ tw, int logn, long n_rows,
                                                          int out_mode, unsigned long long *__restrict__ mxbits, double *scratch)
{
#pragma clang fp contract(off)
    const int t = threadIdx.x, T = blockDim.x;
    const size_t N = (size_t)1 << logn;
    // per workgroup: two arrays of N complex128; the forward transform's two real arrays live in the first
    double2 *C0 = reinterpret_cast<double2 *>(scratch) + (size_t)blockIdx.x * 2 * N, *C1 = C0 + N;
    double *R0 = reinterpret_cast<double *>(C1), *R1 = R0 + N;       // (in C1: C0 receives the masked spectrum)
    const double fct = 1.0 / (double)N;
    for (long f = blockIdx.x; f < n_rows; f += gridDim.x) {
        const double *row = x + (size_t)f * N;
        __syncthreads();
        for (size_t e = t; e < N; e += T) R0[e] = row[e];
        __syncthreads();
        double *cc = R0, *ch = R1;
        // forward: radf4 with ido = 1, 4, 16, ...
        for (int lido = 0; lido + 2 <= logn; lido += 2) {
            const size_t ido = (size_t)1 << lido, l1 = N >> (lido + 2);
            auto CC = [&](size_t a, size_t b, size_t c) -> double { return cc[a + ido * (b + l1 * c)]; };
            auto CH = [&](size_t a, size_t b, size_t c) -> double & { return ch[a + ido * (b + 4 * c)]; };
            const size_t half = ido >> 1;            // slots per k: s = 0 (the two twiddle-free butterflies), s = 1 .. ido/2 - 1
            const size_t units = l1 * (ido == 1 ? 1 : half);
            for (size_t u = t; u < units; u += T) {
                const size_t k = ido == 1 ? u : u / half, s = ido == 1 ? 0 : u % half;
                if (s == 0) {
                    const double c0 = CC(0, k, 0), c1 = CC(0, k, 1), c2_ = CC(0, k, 2), c3 = CC(0, k, 3);
                    const double tr1 = c3 + c1, tr2 = c0 + c2_;
                    CH(0, 0, k) = tr2 + tr1; CH(ido - 1, 3, k) = tr2 - tr1; CH(0, 2, k) = c3 - c1; CH(ido - 1, 1, k) = c0 - c2_;
                    if (ido > 1) {
                        const double d0 = CC(ido - 1, k, 0), d1 = CC(ido - 1, k, 1), d2 = CC(ido - 1, k, 2), d3 = CC(ido - 1, k, 3);
                        const double ti1 = -HSQT2 * (d1 + d3), tb = HSQT2 * (d1 - d3);
                        CH(ido - 1, 0, k) = d0 + tb; CH(ido - 1, 2, k) = d0 - tb; CH(0, 3, k) = ti1 + d2; CH(0, 1, k) = ti1 - d2;
                    }
                } else {
                    const size_t i = 2 * s, ic = ido - i;
                    const double2 w1 = tw[1 * l1 * s], w2 = tw[2 * l1 * s], w3 = tw[3 * l1 * s];
                    const double a0 = CC(i - 1, k, 0), b0 = CC(i, k, 0), a1 = CC(i - 1, k, 1), b1 = CC(i, k, 1);
                    const double a2 = CC(i - 1, k, 2), b2 = CC(i, k, 2), a3 = CC(i - 1, k, 3), b3 = CC(i, k, 3);
                    const double cr2 = w1.x * a1 + w1.y * b1, ci2 = w1.x * b1 - w1.y * a1;
                    const double cr3 = w2.x * a2 + w2.y * b2, ci3 = w2.x * b2 - w2.y * a2;
                    const double cr4 = w3.x * a3 + w3.y * b3, ci4 = w3.x * b3 - w3.y * a3;
                    const double tr1 = cr4 + cr2, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
                    const double tr2 = a0 + cr3, tr3 = a0 - cr3, ti2 = b0 + ci3, ti3 = b0 - ci3;
                    CH(i - 1, 0, k) = tr2 + tr1; CH(ic - 1, 3, k) = tr2 - tr1; CH(i, 0, k) = ti1 + ti2; CH(ic, 3, k) = ti1 - ti2;
                    CH(i - 1, 2, k) = tr3 + ti4; CH(ic - 1, 1, k) = tr3 - ti4; CH(i, 2, k) = tr4 + ti3; CH(ic, 1, k) = tr4 - ti3;
                }
            }
            __syncthreads();
            double *q = cc; cc = ch; ch = q;
        }
        if (logn & 1) {
            const size_t ido = N >> 1;               // radf2, l1 = 1
            for (size_t s = t; s < (ido >> 1); s += T) {
                if (s == 0) {
                    const double c0 = cc[0], c1 = cc[ido], e0 = cc[ido - 1], e1 = cc[2 * ido - 1];
                    ch[0] = c0 + c1; ch[2 * ido - 1] = c0 - c1; ch[ido] = -e1; ch[ido - 1] = e0;
                } else {
                    const size_t i = 2 * s, ic = ido - i;
                    const double2 w = tw[s];
                    const double a0 = cc[i - 1], b0 = cc[i], a1 = cc[i - 1 + ido], b1 = cc[i + ido];
                    const double tr2 = w.x * a1 + w.y * b1, ti2 = w.x * b1 - w.y * a1;
                    ch[i - 1] = a0 + tr2; ch[ic - 1 + ido] = a0 - tr2; ch[i] = ti2 + b0; ch[ic + ido] = ti2 - b0;
                }
            }
            __syncthreads();
            double *q = cc; cc = ch; ch = q;
        }
        // the masked spectrum out of the half-complex row (cc) into C0
        for (size_t b = t; b < N; b += T) {
            const bool low = b > 0 && b < (N >> 1);
            const double sr = low ? 2.0 : ((b == 0 || b == (N >> 1)) ? 1.0 : 0.0), sq = low ? 2.0 : 0.0;
            C0[b] = make_double2(cc[low ? 2 * b - 1 : (b == 0 ? 0 : N - 1)] * sr, cc[low ? 2 * b : 0] * sq);
        }
        __syncthreads();
        // inverse: factors 8 .. 8 (4) with the 2 in front, l1 = 1 upwards
        const int n8 = logn / 3, rem = logn % 3, npass = n8 + (rem ? 1 : 0);
        double2 *pc = C0, *ph = C1;
        int ll1 = 0;
        auto add = [](c2 a, c2 b) {
#pragma clang fp contract(off)
            c2 o = {a.r + b.r, a.i + b.i}; return o; };
        auto sub = [](c2 a, c2 b) {
#pragma clang fp contract(off)
            c2 o = {a.r - b.r, a.i - b.i}; return o; };
        auto rot90 = [](c2 a) { c2 o = {-a.i, a.r}; return o; };
        auto rot45 = [](c2 a) {
#pragma clang fp contract(off)
            c2 o = {HSQT2 * (a.r - a.i), HSQT2 * (a.i + a.r)}; return o; };
        auto rot135 = [](c2 a) {
#pragma clang fp contract(off)
            c2 o = {HSQT2 * (-a.r - a.i), HSQT2 * (a.r - a.i)}; return o; };
        for (int pi = 0; pi < npass; pi++) {
            const int lip = rem == 1 ? (pi == 0 ? 1 : 3) : (pi < n8 ? 3 : 2);
            const int lido = logn - ll1 - lip;
            const size_t ido = (size_t)1 << lido, l1 = (size_t)1 << ll1, ip = (size_t)1 << lip;
            auto bmul = [&](c2 v, size_t j, size_t i) {
#pragma clang fp contract(off)
                const double2 wj = tw[j * l1 * i];
                c2 o = {v.r * wj.x - v.i * wj.y, v.r * wj.y + v.i * wj.x};
                if (i == 0) o = v;
                return o;
            };
            for (size_t u = t; u < (N >> lip); u += T) {
                const size_t k = u >> lido, i = u & (ido - 1);
                c2 c[8], o[8];
                for (size_t j = 0; j < ip; j++) { const double2 v = pc[i + ido * (j + ip * k)]; c[j].r = v.x; c[j].i = v.y; }
                if (lip == 3) {
                    c2 a1 = add(c[1], c[5]), a5 = sub(c[1], c[5]), a3 = add(c[3], c[7]), a7 = sub(c[3], c[7]);
                    { const c2 q = a1; a1 = add(a1, a3); a3 = sub(q, a3); }
                    a3 = rot90(a3);
                    a7 = rot90(a7);
                    { const c2 q = a5; a5 = add(a5, a7); a7 = sub(q, a7); }
                    a5 = rot45(a5);
                    a7 = rot135(a7);
                    c2 a0 = add(c[0], c[4]), a4 = sub(c[0], c[4]), a2 = add(c[2], c[6]), a6 = sub(c[2], c[6]);
                    { const c2 q = a0; a0 = add(a0, a2); a2 = sub(q, a2); }
                    a6 = rot90(a6);
                    { const c2 q = a4; a4 = add(a4, a6); a6 = sub(q, a6); }
                    o[0] = add(a0, a1); o[4] = bmul(sub(a0, a1), 4, i); o[2] = bmul(add(a2, a3), 2, i); o[6] = bmul(sub(a2, a3), 6, i);
                    o[1] = bmul(add(a4, a5), 1, i); o[5] = bmul(sub(a4, a5), 5, i); o[3] = bmul(add(a6, a7), 3, i); o[7] = bmul(sub(a6, a7), 7, i);
                } else if (lip == 2) {
                    const c2 t2 = add(c[0], c[2]), t1 = sub(c[0], c[2]), t3 = add(c[1], c[3]);
                    const c2 t4 = rot90(sub(c[1], c[3]));
                    o[0] = add(t2, t3); o[1] = bmul(add(t1, t4), 1, i); o[2] = bmul(sub(t2, t3), 2, i); o[3] = bmul(sub(t1, t4), 3, i);
                } else {
                    o[0] = add(c[0], c[1]); o[1] = bmul(sub(c[0], c[1]), 1, i);
                }
                for (size_t j = 0; j < ip; j++) ph[i + ido * (k + l1 * j)] = make_double2(o[j].r, o[j].i);
            }
            __syncthreads();
            double2 *q = pc; pc = ph; ph = q;
            ll1 += lip;
        }
        double m = 0.0;
        for (size_t e = t; e < N; e += T) {
            const double2 v = pc[e];
            const double a = v.x * fct, b = v.y * fct;
            if (out_mode == 0) reinterpret_cast<double2 *>(out)[(size_t)f * N + e] = make_double2(a, b);
            else out[(size_t)f * N + e] = a;
            m = pss_hil::nanmax(m, fabs(a));
        }
        if (out_mode == 1 && mxbits) {
            for (int off = 32; off > 0; off >>= 1) m = pss_hil::nanmax(m, __shfl_xor(m, off));
            if ((t & 63) == 0) atomicMax(&mxbits[f], (unsigned long long)__double_as_longlong(m));
        }
    }
}

}  // namespace pss_pf
