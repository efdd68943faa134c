/*
 * pss_pocketfft.c — CPU restatement of the transform behind scipy.signal.hilbert for power-of-two lengths.  TEST INFRASTRUCTURE ONLY
 * (part of the oracle, see pss_oracle.h).
 *
 * The reference (signal_processing.py:205 / :210) calls scipy.signal.hilbert(x) on a real float64 row:
 *     Xf = scipy.fft.fft(x);  h = [1, 2, ..., 2, 1, 0, ..., 0];  analytic = scipy.fft.ifft(Xf * h)
 * scipy.fft is pypocketfft (third-party, vendored in SciPy 1.15.3 as scipy/_lib/pocketfft, the header-only C++ pocketfft; not part of
 * /root/reference).  Its published algorithm, restated here for n = 2^k:
 *   - real input: c2c on a real array runs r2c (rfftp: radix-4 passes radf4, one radix-2 pass radf2 for odd k, executed from the LAST
 *     factor to the first, FFTPACK half-complex layout) and fills the upper half with conjugates;
 *   - inverse: cfftp backward with factors 8, 8, ..., then 4, then 2 (the 2 swapped to the front), passes pass8 / pass4 / pass2 from the
 *     first factor on, ping-pong between two buffers, then the scale 1 / n;
 *   - twiddles: sincos_2pibyn — exp(2 pi i k / n) as the product of two table entries (k & mask, k >> shift), each entry cos / sin of
 *     a multiple of pi / (4 n) folded into the first octant, all in double (libm cos / sin);
 *   - plain multiplies and adds, no fused multiply-add (the wheels are built for baseline x86-64).
 * Pinned by tests/test_oracle_golden.py::test_pocketfft_model_is_scipy_bit_for_bit (against SciPy itself where the test runs) and by the
 * float64 SSB audio of tests/golden/am_ssb.npz (array_equal).
 */
#include "pss_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double r, i; } cpx;

/* ---- sincos_2pibyn<double> -------------------------------------------------------------------------------------------- */
typedef struct { size_t N, mask, shift; cpx *v1, *v2; } sincos_t;

static cpx sc_calc(size_t x, size_t n, double ang)
{
    cpx c;
    x <<= 3;
    if (x < 4 * n) {
        if (x < 2 * n) {
            if (x < n) { c.r = cos((double)x * ang); c.i = sin((double)x * ang); return c; }
            c.r = sin((double)(2 * n - x) * ang); c.i = cos((double)(2 * n - x) * ang); return c;
        }
        x -= 2 * n;
        if (x < n) { c.r = -sin((double)x * ang); c.i = cos((double)x * ang); return c; }
        c.r = -cos((double)(2 * n - x) * ang); c.i = sin((double)(2 * n - x) * ang); return c;
    }
    x = 8 * n - x;
    if (x < 2 * n) {
        if (x < n) { c.r = cos((double)x * ang); c.i = -sin((double)x * ang); return c; }
        c.r = sin((double)(2 * n - x) * ang); c.i = -cos((double)(2 * n - x) * ang); return c;
    }
    x -= 6 * n;
    if (x < n) { c.r = -sin((double)x * ang); c.i = -cos((double)x * ang); return c; }
    c.r = -cos((double)(2 * n - x) * ang); c.i = -sin((double)(2 * n - x) * ang); return c;
}

static void sc_init(sincos_t *s, size_t n)
{
    const long double pi = 3.141592653589793238462643383279502884197L;
    const double ang = (double)(0.25L * pi / (long double)n);
    const size_t nval = (n + 2) / 2;
    s->N = n;
    s->shift = 1;
    while (((size_t)1 << s->shift) * ((size_t)1 << s->shift) < nval) ++s->shift;
    s->mask = ((size_t)1 << s->shift) - 1;
    const size_t n1 = s->mask + 1, n2 = (nval + s->mask) / (s->mask + 1);
    s->v1 = (cpx *)malloc(sizeof(cpx) * n1);
    s->v2 = (cpx *)malloc(sizeof(cpx) * n2);
    s->v1[0].r = 1.0; s->v1[0].i = 0.0;
    for (size_t i = 1; i < n1; i++) s->v1[i] = sc_calc(i, n, ang);
    s->v2[0].r = 1.0; s->v2[0].i = 0.0;
    for (size_t i = 1; i < n2; i++) s->v2[i] = sc_calc(i * (s->mask + 1), n, ang);
}
static void sc_free(sincos_t *s) { free(s->v1); free(s->v2); }
static cpx sc_at(const sincos_t *s, size_t idx)
{
    cpx c;
    if (2 * idx <= s->N) {
        const cpx x1 = s->v1[idx & s->mask], x2 = s->v2[idx >> s->shift];
        c.r = x1.r * x2.r - x1.i * x2.i; c.i = x1.r * x2.i + x1.i * x2.r;
        return c;
    }
    idx = s->N - idx;
    const cpx x1 = s->v1[idx & s->mask], x2 = s->v2[idx >> s->shift];
    c.r = x1.r * x2.r - x1.i * x2.i; c.i = -(x1.r * x2.i + x1.i * x2.r);
    return c;
}

/* exp(2 pi i k / n), k = 0..n-1, exactly as pocketfft's plans tabulate it (the GPU side uploads this table) */
void pss_o_pocketfft_twiddles(int n, double *tw_re_im)
{
    sincos_t s;
    sc_init(&s, (size_t)n);
    for (int k = 0; k < n; k++) { const cpx c = sc_at(&s, (size_t)k); tw_re_im[2 * k] = c.r; tw_re_im[2 * k + 1] = c.i; }
    sc_free(&s);
}

/* ---- rfftp, forward (r2hc) -------------------------------------------------------------------------------------------- */
#define PM(a, b, c, d) { a = (c) + (d); b = (c) - (d); }
#define MULPM(a, b, c, d, e, f) { a = (c) * (e) + (d) * (f); b = (c) * (f) - (d) * (e); }

static void radf4(size_t ido, size_t l1, const double *cc, double *ch, const double *wa)
{
    const double hsqt2 = 0.707106781186547524400844362104849;
#define WA(x, i) wa[(i) + (x) * (ido - 1)]
#define CC(a, b, c) cc[(a) + ido * ((b) + l1 * (c))]
#define CH(a, b, c) ch[(a) + ido * ((b) + 4 * (c))]
    for (size_t k = 0; k < l1; k++) {
        double tr1, tr2;
        PM(tr1, CH(0, 2, k), CC(0, k, 3), CC(0, k, 1))
        PM(tr2, CH(ido - 1, 1, k), CC(0, k, 0), CC(0, k, 2))
        PM(CH(0, 0, k), CH(ido - 1, 3, k), tr2, tr1)
    }
    if ((ido & 1) == 0)
        for (size_t k = 0; k < l1; k++) {
            const double ti1 = -hsqt2 * (CC(ido - 1, k, 1) + CC(ido - 1, k, 3));
            const double tr1 = hsqt2 * (CC(ido - 1, k, 1) - CC(ido - 1, k, 3));
            PM(CH(ido - 1, 0, k), CH(ido - 1, 2, k), CC(ido - 1, k, 0), tr1)
            PM(CH(0, 3, k), CH(0, 1, k), ti1, CC(ido - 1, k, 2))
        }
    if (ido <= 2) return;
    for (size_t k = 0; k < l1; k++)
        for (size_t i = 2; i < ido; i += 2) {
            const size_t ic = ido - i;
            double ci2, ci3, ci4, cr2, cr3, cr4, ti1, ti2, ti3, ti4, tr1, tr2, tr3, tr4;
            MULPM(cr2, ci2, WA(0, i - 2), WA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
            MULPM(cr3, ci3, WA(1, i - 2), WA(1, i - 1), CC(i - 1, k, 2), CC(i, k, 2))
            MULPM(cr4, ci4, WA(2, i - 2), WA(2, i - 1), CC(i - 1, k, 3), CC(i, k, 3))
            PM(tr1, tr4, cr4, cr2)
            PM(ti1, ti4, ci2, ci4)
            PM(tr2, tr3, CC(i - 1, k, 0), cr3)
            PM(ti2, ti3, CC(i, k, 0), ci3)
            PM(CH(i - 1, 0, k), CH(ic - 1, 3, k), tr2, tr1)
            PM(CH(i, 0, k), CH(ic, 3, k), ti1, ti2)
            PM(CH(i - 1, 2, k), CH(ic - 1, 1, k), tr3, ti4)
            PM(CH(i, 2, k), CH(ic, 1, k), tr4, ti3)
        }
#undef CH
}

static void radf2(size_t ido, size_t l1, const double *cc, double *ch, const double *wa)
{
#define CH(a, b, c) ch[(a) + ido * ((b) + 2 * (c))]
    for (size_t k = 0; k < l1; k++) PM(CH(0, 0, k), CH(ido - 1, 1, k), CC(0, k, 0), CC(0, k, 1))
    if ((ido & 1) == 0)
        for (size_t k = 0; k < l1; k++) {
            CH(0, 1, k) = -CC(ido - 1, k, 1);
            CH(ido - 1, 0, k) = CC(ido - 1, k, 0);
        }
    if (ido <= 2) return;
    for (size_t k = 0; k < l1; k++)
        for (size_t i = 2; i < ido; i += 2) {
            const size_t ic = ido - i;
            double tr2, ti2;
            MULPM(tr2, ti2, WA(0, i - 2), WA(0, i - 1), CC(i - 1, k, 1), CC(i, k, 1))
            PM(CH(i - 1, 0, k), CH(ic - 1, 1, k), CC(i - 1, k, 0), tr2)
            PM(CH(i, 0, k), CH(ic, 1, k), ti2, CC(i, k, 0))
        }
#undef CH
#undef CC
#undef WA
}

/* half-complex transform of c[0..n) in place: r0, r1, i1, r2, i2, ..., r_{n/2} */
static void rfft_forward(double *c, size_t n)
{
    if (n == 1) return;
    size_t fact[32], nf = 0, len = n;
    while ((len & 3) == 0) { fact[nf++] = 4; len >>= 2; }
    if ((len & 1) == 0) { len >>= 1; fact[nf++] = 2; const size_t t = fact[0]; fact[0] = fact[nf - 1]; fact[nf - 1] = t; }
    sincos_t sc;
    sc_init(&sc, n);
    /* twiddles per factor, in plan order (l1 grows from 1); the last factor needs none */
    double *tw[32];
    size_t l1 = 1;
    for (size_t k = 0; k < nf; k++) {
        const size_t ip = fact[k], ido = n / (l1 * ip);
        tw[k] = NULL;
        if (k < nf - 1) {
            tw[k] = (double *)malloc(sizeof(double) * ((ip - 1) * (ido - 1) + 1));
            for (size_t j = 1; j < ip; j++)
                for (size_t i = 1; i <= (ido - 1) / 2; i++) {
                    const cpx w = sc_at(&sc, j * l1 * i);
                    tw[k][(j - 1) * (ido - 1) + 2 * i - 2] = w.r;
                    tw[k][(j - 1) * (ido - 1) + 2 * i - 1] = w.i;
                }
        }
        l1 *= ip;
    }
    double *ch = (double *)malloc(sizeof(double) * n), *p1 = c, *p2 = ch;
    l1 = n;
    for (size_t k1 = 0; k1 < nf; k1++) {
        const size_t k = nf - k1 - 1, ip = fact[k], ido = n / l1;
        l1 /= ip;
        if (ip == 4) radf4(ido, l1, p1, p2, tw[k]);
        else radf2(ido, l1, p1, p2, tw[k]);
        double *t = p1; p1 = p2; p2 = t;
    }
    if (p1 != c) memcpy(c, p1, sizeof(double) * n);
    for (size_t k = 0; k < nf; k++) free(tw[k]);
    free(ch);
    sc_free(&sc);
}

/* ---- cfftp, backward --------------------------------------------------------------------------------------------------- */
static inline cpx cadd(cpx a, cpx b) { cpx c = {a.r + b.r, a.i + b.i}; return c; }
static inline cpx csub(cpx a, cpx b) { cpx c = {a.r - b.r, a.i - b.i}; return c; }
static inline cpx bmul(cpx v1, cpx v2) { cpx c = {v1.r * v2.r - v1.i * v2.i, v1.r * v2.i + v1.i * v2.r}; return c; }   /* special_mul<false> */
static inline cpx rot90(cpx a) { cpx c = {-a.i, a.r}; return c; }                                                     /* ROTX90<false>  */
static inline cpx rot45(cpx a)
{
    const double hsqt2 = 0.707106781186547524400844362104849;
    cpx c = {hsqt2 * (a.r - a.i), hsqt2 * (a.i + a.r)};
    return c;
}
static inline cpx rot135(cpx a)
{
    const double hsqt2 = 0.707106781186547524400844362104849;
    cpx c = {hsqt2 * (-a.r - a.i), hsqt2 * (a.r - a.i)};
    return c;
}
#define CPM(a, b, c, d) { a = cadd(c, d); b = csub(c, d); }
#define CPMIN(a, b) { const cpx t_ = a; a = cadd(a, b); b = csub(t_, b); }

#define WAc(x, i) wa[(i) - 1 + (x) * (ido - 1)]
#define CHc(a, b, c) ch[(a) + ido * ((b) + l1 * (c))]

static void pass2b(size_t ido, size_t l1, const cpx *cc, cpx *ch, const cpx *wa)
{
#define CCc(a, b, c) cc[(a) + ido * ((b) + 2 * (c))]
    for (size_t k = 0; k < l1; k++) {
        CHc(0, k, 0) = cadd(CCc(0, 0, k), CCc(0, 1, k));
        CHc(0, k, 1) = csub(CCc(0, 0, k), CCc(0, 1, k));
        for (size_t i = 1; i < ido; i++) {
            CHc(i, k, 0) = cadd(CCc(i, 0, k), CCc(i, 1, k));
            CHc(i, k, 1) = bmul(csub(CCc(i, 0, k), CCc(i, 1, k)), WAc(0, i));
        }
    }
#undef CCc
}

static void pass4b(size_t ido, size_t l1, const cpx *cc, cpx *ch, const cpx *wa)
{
#define CCc(a, b, c) cc[(a) + ido * ((b) + 4 * (c))]
    for (size_t k = 0; k < l1; k++)
        for (size_t i = 0; i < ido; i++) {
            cpx t1, t2, t3, t4;
            CPM(t2, t1, CCc(i, 0, k), CCc(i, 2, k))
            CPM(t3, t4, CCc(i, 1, k), CCc(i, 3, k))
            t4 = rot90(t4);
            if (i == 0) {
                CPM(CHc(0, k, 0), CHc(0, k, 2), t2, t3)
                CPM(CHc(0, k, 1), CHc(0, k, 3), t1, t4)
            } else {
                CHc(i, k, 0) = cadd(t2, t3);
                CHc(i, k, 1) = bmul(cadd(t1, t4), WAc(0, i));
                CHc(i, k, 2) = bmul(csub(t2, t3), WAc(1, i));
                CHc(i, k, 3) = bmul(csub(t1, t4), WAc(2, i));
            }
        }
#undef CCc
}

static void pass8b(size_t ido, size_t l1, const cpx *cc, cpx *ch, const cpx *wa)
{
#define CCc(a, b, c) cc[(a) + ido * ((b) + 8 * (c))]
    for (size_t k = 0; k < l1; k++)
        for (size_t i = 0; i < ido; i++) {
            cpx a0, a1, a2, a3, a4, a5, a6, a7;
            CPM(a1, a5, CCc(i, 1, k), CCc(i, 5, k))
            CPM(a3, a7, CCc(i, 3, k), CCc(i, 7, k))
            CPMIN(a1, a3)
            a3 = rot90(a3);
            a7 = rot90(a7);
            CPMIN(a5, a7)
            a5 = rot45(a5);
            a7 = rot135(a7);
            CPM(a0, a4, CCc(i, 0, k), CCc(i, 4, k))
            CPM(a2, a6, CCc(i, 2, k), CCc(i, 6, k))
            if (i == 0) {
                CPM(CHc(0, k, 0), CHc(0, k, 4), cadd(a0, a2), a1)
                CPM(CHc(0, k, 2), CHc(0, k, 6), csub(a0, a2), a3)
                a6 = rot90(a6);
                CPM(CHc(0, k, 1), CHc(0, k, 5), cadd(a4, a6), a5)
                CPM(CHc(0, k, 3), CHc(0, k, 7), csub(a4, a6), a7)
            } else {
                CPMIN(a0, a2)
                CHc(i, k, 0) = cadd(a0, a1);
                CHc(i, k, 4) = bmul(csub(a0, a1), WAc(3, i));
                CHc(i, k, 2) = bmul(cadd(a2, a3), WAc(1, i));
                CHc(i, k, 6) = bmul(csub(a2, a3), WAc(5, i));
                a6 = rot90(a6);
                CPMIN(a4, a6)
                CHc(i, k, 1) = bmul(cadd(a4, a5), WAc(0, i));
                CHc(i, k, 5) = bmul(csub(a4, a5), WAc(4, i));
                CHc(i, k, 3) = bmul(cadd(a6, a7), WAc(2, i));
                CHc(i, k, 7) = bmul(csub(a6, a7), WAc(6, i));
            }
        }
#undef CCc
}

/* inverse transform of c[0..n) in place, scaled by 1 / n (scipy.fft.ifft, norm="backward") */
static void cfft_backward(cpx *c, size_t n)
{
    if (n == 1) return;
    size_t fact[32], nf = 0, len = n;
    while ((len & 7) == 0) { fact[nf++] = 8; len >>= 3; }
    while ((len & 3) == 0) { fact[nf++] = 4; len >>= 2; }
    if ((len & 1) == 0) { len >>= 1; fact[nf++] = 2; const size_t t = fact[0]; fact[0] = fact[nf - 1]; fact[nf - 1] = t; }
    sincos_t sc;
    sc_init(&sc, n);
    cpx *ch = (cpx *)malloc(sizeof(cpx) * n), *p1 = c, *p2 = ch;
    size_t l1 = 1;
    for (size_t k = 0; k < nf; k++) {
        const size_t ip = fact[k], l2 = ip * l1, ido = n / l2;
        cpx *wa = (cpx *)malloc(sizeof(cpx) * ((ip - 1) * (ido - 1) + 1));
        for (size_t j = 1; j < ip; j++)
            for (size_t i = 1; i < ido; i++) wa[(j - 1) * (ido - 1) + i - 1] = sc_at(&sc, j * l1 * i);
        if (ip == 8) pass8b(ido, l1, p1, p2, wa);
        else if (ip == 4) pass4b(ido, l1, p1, p2, wa);
        else pass2b(ido, l1, p1, p2, wa);
        free(wa);
        cpx *t = p1; p1 = p2; p2 = t;
        l1 = l2;
    }
    const double fct = (double)(1.0L / (long double)n);
    for (size_t i = 0; i < n; i++) { c[i].r = p1[i].r * fct; c[i].i = p1[i].i * fct; }
    free(ch);
    sc_free(&sc);
}

/* scipy.signal.hilbert(x) for a real float64 row of n = 2^k >= 2 samples: analytic[i] = out[2 i] + i out[2 i + 1]. */
void pss_o_hilbert(const double *x, int n, double *out)
{
    const size_t N = (size_t)n;
    double *t = (double *)malloc(sizeof(double) * N);
    memcpy(t, x, sizeof(double) * N);
    rfft_forward(t, N);
    cpx *X = (cpx *)out;
    X[0].r = t[0]; X[0].i = 0.0;
    size_t i = 1, ii = 1;
    for (; i < N - 1; i += 2, ++ii) { X[ii].r = t[i]; X[ii].i = t[i + 1]; }
    if (i < N) { X[ii].r = t[i]; X[ii].i = 0.0; }
    /* Xf * h with h = 1 at 0 and n / 2, 2 below n / 2, 0 above (complex128 h: the products are exact) */
    for (size_t k = 1; k < N / 2; k++) { X[k].r *= 2.0; X[k].i *= 2.0; }
    for (size_t k = N / 2 + 1; k < N; k++) { X[k].r = 0.0; X[k].i = 0.0; }
    cfft_backward(X, N);
    free(t);
}

/* the forward half alone (scipy.fft.fft of a real row), for the unit test of the model: out = n complex values */
void pss_o_rfft_full(const double *x, int n, double *out)
{
    const size_t N = (size_t)n;
    double *t = (double *)malloc(sizeof(double) * N);
    memcpy(t, x, sizeof(double) * N);
    rfft_forward(t, N);
    cpx *X = (cpx *)out;
    X[0].r = t[0]; X[0].i = 0.0;
    size_t i = 1, ii = 1;
    for (; i < N - 1; i += 2, ++ii) { X[ii].r = t[i]; X[ii].i = t[i + 1]; }
    if (i < N) { X[ii].r = t[i]; X[ii].i = 0.0; }
    for (size_t k = N / 2 + 1; k < N; k++) { X[k].r = X[N - k].r; X[k].i = -X[N - k].i; }
    free(t);
}

/* scipy.fft.ifft of a complex row, for the unit test of the model */
void pss_o_cifft(const double *in, int n, double *out)
{
    memcpy(out, in, sizeof(double) * 2 * (size_t)n);
    cfft_backward((cpx *)out, (size_t)n);
}
