/*
 * pss_oracle.c — CPU restatement of the PySpecSDR hot path.  TEST INFRASTRUCTURE ONLY (see pss_oracle.h).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -mfma; contraction OFF because the reference's
 * float64 IIR arithmetic is un-fused x86-64 baseline code, and every fused multiply-add that the
 * reference does execute is written out explicitly with fmaf()).
 *
 * Parity status of each piece against tests/golden (pinned by tests/test_oracle_golden.py):
 *   atan2f / cabsf / pairwise sums / discriminator ........ bit-exact float32
 *   sosfilt / sosfiltfilt / AM chain ...................... bit-exact float64
 *   65-tap FIR (OpenBLAS ddot / zdot accumulation order) ... bit-exact float64 -> NFM audio bit-exact float64
 *   SSB ................................................... bit-exact float64 for frames of 2^k samples (real FIR + SciPy's
 *                                                           hilbert() round trip, pss_pocketfft.c); other lengths skip the
 *                                                           round trip (~1e-16 noise, atol 1e-14), int16 exact
 *   spectrum dB, power dB, scanner dB ..................... tolerance (rtol 1e-9 f64 / few ulp f32)
 */
#include "pss_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------------------------------------
 * VRCP14PS model.  The instruction's result depends only on the top 16 mantissa bits of the input
 * (exhaustively verified over all 2^23 mantissas on an AVX-512 Xeon, tools/derive_rcp14.c):
 *   v17 = (A[m>>17] - B[m>>17] * ((m>>7) & 1023)) >> 9      (17-bit significand, value v17 / 2^17)
 *   result = v17 * 2^-17 * 2^-(e)   for x = (1 + m/2^23) * 2^e,   and exactly 2^-e when m == 0.
 * ---------------------------------------------------------------------------------------------- */
static const uint32_t RCP14_A[64] = {
    67107072u, 66074112u, 65073664u, 64102400u, 63159040u, 62244608u, 61354752u, 60491264u,
    59650560u, 58833920u, 58038272u, 57264640u, 56511488u, 55778048u, 55062784u, 54365184u,
    53686016u, 53022976u, 52377088u, 51745536u, 51129600u, 50528000u, 49940992u, 49366272u,
    48805376u, 48257024u, 47721728u, 47196672u, 46683904u, 46181632u, 45690368u, 45209344u,
    44739072u, 44277504u, 43826176u, 43382784u, 42949120u, 42523904u, 42106880u, 41698048u,
    41297920u, 40903936u, 40517888u, 40139520u, 39768320u, 39402752u, 39044608u, 38692864u,
    38347520u, 38008064u, 37674496u, 37347840u, 37025280u, 36708608u, 36398080u, 36091648u,
    35791360u, 35495680u, 35204352u, 34919168u, 34638080u, 34361088u, 34088192u, 33819392u};
static const uint16_t RCP14_B[64] = {
    1009, 977, 949, 921, 893, 869, 843, 821, 797, 777, 755, 735, 717, 699, 681, 663,
    647, 631, 617, 601, 587, 573, 561, 547, 535, 523, 513, 501, 491, 479, 469, 459,
    451, 441, 433, 423, 415, 407, 399, 391, 385, 377, 369, 363, 357, 349, 343, 337,
    331, 325, 319, 315, 309, 303, 299, 293, 289, 285, 279, 275, 271, 267, 263, 259};

float pss_o_rcp14f(float x)
{
    /* valid for normal x with a normal result (the only use: SVML atan2f main path, 2^-125 <= x < 2^123) */
    uint32_t u = f2u(x), sign = u & 0x80000000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    if (m == 0) return u2f(sign | ((254u - e) << 23));
    uint32_t idx = m >> 17, low = (m >> 7) & 1023u;
    uint32_t v = (RCP14_A[idx] - (uint32_t)RCP14_B[idx] * low) >> 9; /* in [2^16, 2^17) */
    return u2f(sign | ((253u - e) << 23) | ((v & 0xffffu) << 7));
}

/* ------------------------------------------------------------------------------------------------
 * numpy.arctan2 for float32 under the AVX512_SKX dispatch = Intel SVML __svml_atan2f16 ("la" variant,
 * shipped in NumPy's bundled numpy/SVML sources).  Called by np.angle at signal_processing.py:94.
 * Main path (both |x| and |y| in [2^-125, 2^123)): reciprocal by VRCP14 + one Newton step, quotient
 * with one correction, degree-8-in-q^2 polynomial split into two interleaved Horner chains, all in
 * float32 FMA.  Constants are the routine's __svml_satan2_data_internal table.
 * ---------------------------------------------------------------------------------------------- */
float pss_o_atan2f(float y, float x)
{
    const float PIO2 = 0x1.921fb6p+0f, PI = 0x1.921fb6p+1f;
    uint32_t xb = f2u(x), yb = f2u(y);
    uint32_t axb = xb & 0x7fffffffu, ayb = yb & 0x7fffffffu;
    uint32_t sx = xb & 0x80000000u, sy = yb & 0x80000000u;
    float ax = u2f(axb), ay = u2f(ayb);
    int inx = (axb >= 0x01000000u) && (axb < 0x7d000000u);
    int iny = (ayb >= 0x01000000u) && (ayb < 0x7d000000u);
    if (!(inx && iny)) {
        if (x != x || y != y) return x + y;
        if (axb == 0 || ayb == 0) { /* the routine's vector fix-up for zero operands */
            float v = (!(ay < ax) && !(axb == 0 && ayb == 0)) ? PIO2 : 0.0f;
            v = u2f(f2u(v) | sx);
            if (sx) v = v + PI;
            return u2f(f2u(v) | sy);
        }
        /* inf / denormal / huge operands go to the routine's scalar "rare" helper, which works in
         * double precision; restated as correctly rounded double atan2 (not bit-pinned). */
        return (float)atan2((double)y, (double)x);
    }
    int k1 = ay < ax;
    float a = k1 ? ay : -ax;
    float b = k1 ? ax : ay;
    float base = k1 ? 0.0f : PIO2;
    float r0 = pss_o_rcp14f(b);
    float e = fmaf(-b, r0, 1.0f);
    float r1 = fmaf(r0, e, r0);
    float q0 = a * r1;
    float rem = fmaf(-b, q0, a);
    float q = fmaf(rem, r1, q0);
    float s = q * q;
    float s2 = s * s;
    float pa = fmaf(s2, 0x1.64598p-9f, 0x1.578708p-5f);
    float pb = fmaf(s2, -0x1.fe4c62p-7f, -0x1.30ec52p-4f);
    pa = fmaf(pa, s2, 0x1.b2c8e8p-4f);
    pb = fmaf(pb, s2, -0x1.22c3fp-3f);
    pa = fmaf(pa, s2, 0x1.996f3ep-3f);
    pb = fmaf(pb, s2, -0x1.555492p-2f);
    pa = fmaf(pa, s2, 1.0f);
    float p = fmaf(pb, s, pa);
    float r = fmaf(p, q, base);
    r = u2f(f2u(r) | sx);
    if (x <= 0.0f) r = r + PI;
    return u2f(f2u(r) | sy);
}

/* numpy.abs(complex64), AVX512F loop: scaled hypot mx*sqrt(fma(r,r,1)), r = mn/mx.
 * Used by signal_processing.py:182 (AM envelope) and :327 (power). */
float pss_o_cabsf(float re, float im)
{
    float a = fabsf(re), b = fabsf(im);
    float mx = a > b ? a : b, mn = a > b ? b : a;
    if (mx == 0.0f) return 0.0f;
    float r = mn / mx;
    return mx * sqrtf(fmaf(r, r, 1.0f));
}

/* numpy add.reduce over float32 (np.mean at signal_processing.py:185 and :327):
 *  - the ufunc machinery hands the inner loop at most 8192 elements at a time (its buffer size) and adds the chunk
 *    results up sequentially: sum = ((S(c0) + S(c1)) + S(c2)) + ...   (probed: tools/probe notes in DESIGN.md §2);
 *  - inside a chunk: pairwise summation, block 128, 8 accumulators (numpy loops_utils.h.src @TYPE@_pairwise_sum). */
static float pairwise_chunk_f32(const float *a, long n)
{
    if (n < 8) {
        float res = 0.0f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_chunk_f32(a, n2) + pairwise_chunk_f32(a + n2, n - n2);
    }
}

float pss_o_pairwise_sum_f32(const float *a, long n)
{
    const long B = 8192;
    if (n <= B) return pairwise_chunk_f32(a, n);
    float acc = pairwise_chunk_f32(a, B);
    for (long st = B; st < n; st += B) acc += pairwise_chunk_f32(a + st, (n - st) < B ? (n - st) : B);
    return acc;
}

/* np.log10 on float32 under NumPy's AVX512_SKX dispatch = Intel SVML __svml_log10f16 (inside NumPy's _multiarray_umath):
 * m = mantissa scaled into [0.75, 1.5), k the matching exponent, r = m - 1, degree-4 polynomial in r with coefficients
 * selected by the top four mantissa bits of m, result fma(p, r, k * log10(2)), all float32.  tools/check_log10f_model.py
 * compares this function with np.log10 on EVERY positive finite float32: no differing bit.  Specials as the library
 * returns them: +-0 -> -inf, negative -> NaN, +inf -> +inf, NaN -> NaN. */
static const uint32_t L10_C0[16] = {0xbdc9ae9bu, 0xbda6fcf4u, 0xbd8bac76u, 0xbd6bca30u, 0xbd48a99bu, 0xbd2c0a9fu, 0xbd1480dbu, 0xbd00faf2u,
                                    0xbe823aa9u, 0xbe656348u, 0xbe4afbb9u, 0xbe346895u, 0xbe20ffffu, 0xbe103a0bu, 0xbe01a91cu, 0xbde9e84eu};
static const uint32_t L10_C1[16] = {0x3e13d888u, 0x3e10a87cu, 0x3e0b95c3u, 0x3e057f0bu, 0x3dfde038u, 0x3df080d9u, 0x3de34c1eu, 0x3dd68333u,
                                    0x3dac6e8eu, 0x3dd54a51u, 0x3df30f40u, 0x3e04235du, 0x3e0b7033u, 0x3e102c90u, 0x3e12ebadu, 0x3e141ff8u};
static const uint32_t L10_C2[16] = {0xbe5e5a9bu, 0xbe5e2677u, 0xbe5d83f5u, 0xbe5c6016u, 0xbe5abd0bu, 0xbe58a6fdu, 0xbe562e02u, 0xbe5362f8u,
                                    0xbe68e27cu, 0xbe646747u, 0xbe619a73u, 0xbe5ff05au, 0xbe5f0570u, 0xbe5e92d0u, 0xbe5e662bu, 0xbe5e5c08u};
static const uint32_t L10_C3[16] = {0x3ede5bd8u, 0x3ede5b45u, 0x3ede57d8u, 0x3ede4eb1u, 0x3ede3d37u, 0x3ede2166u, 0x3eddf9d9u, 0x3eddc5bbu,
                                    0x3ede08edu, 0x3ede32e7u, 0x3ede4967u, 0x3ede5490u, 0x3ede597fu, 0x3ede5b50u, 0x3ede5bcau, 0x3ede5bd9u};
float pss_o_log10f_np(float x)
{
    uint32_t b;
    memcpy(&b, &x, 4);
    if (x != x) return x;
    if ((b & 0x7fffffffu) == 0u) return -INFINITY;
    if (b & 0x80000000u) return NAN;
    if (b == 0x7f800000u) return x;
    int e = (int)(b >> 23);
    uint32_t man = b & 0x7fffffu;
    if (e == 0) { int sh = 0; while (!(man & 0x800000u)) { man <<= 1; sh++; } man &= 0x7fffffu; e = 1 - sh; }
    int k = e - 127;
    uint32_t mb;
    if (man >= 0x400000u) { mb = (126u << 23) | man; k += 1; } else mb = (127u << 23) | man;
    const int idx = (int)(mb >> 19) & 15;
    const float r = u2f(mb) - 1.0f;
    float p = fmaf(u2f(L10_C0[idx]), r, u2f(L10_C1[idx]));
    p = fmaf(p, r, u2f(L10_C2[idx]));
    p = fmaf(p, r, u2f(L10_C3[idx]));
    return fmaf(p, r, (float)k * u2f(0x3e9a209bu));
}
float pss_o_log10f_ref(float x) { return pss_o_log10f_np(x); }
void pss_o_log10f_np_many(const float *x, float *y, long n) { for (long i = 0; i < n; i++) y[i] = pss_o_log10f_np(x[i]); }
void pss_o_atan2f_many(const float *y, const float *x, float *out, long n) { for (long i = 0; i < n; i++) out[i] = pss_o_atan2f(y[i], x[i]); }
void pss_o_cabsf_many(const float *re, const float *im, float *out, long n) { for (long i = 0; i < n; i++) out[i] = pss_o_cabsf(re[i], im[i]); }

/* numpy add.reduce on complex64 (CFLOAT_pairwise_sum, loops_utils.h.src): the interleaved float array is summed
 * with 8 accumulators (4 complex lanes), block 128 FLOATS, fold (r0+r2)+(r4+r6) / (r1+r3)+(r5+r7); the ufunc
 * buffer hands over 8192 COMPLEX elements at a time and chunk sums are added sequentially. a = interleaved, n floats. */
static void cpairwise_chunk_f32(const float *a, long n, float *rr, float *ri)
{
    if (n < 8) {
        *rr = 0.0f; *ri = 0.0f;
        for (long i = 0; i < n; i += 2) { *rr += a[i]; *ri += a[i + 1]; }
    } else if (n <= 128) {
        float r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        *rr = (r[0] + r[2]) + (r[4] + r[6]);
        *ri = (r[1] + r[3]) + (r[5] + r[7]);
        for (; i < n; i += 2) { *rr += a[i]; *ri += a[i + 1]; }
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        float ar, ai, br, bi;
        cpairwise_chunk_f32(a, n2, &ar, &ai);
        cpairwise_chunk_f32(a + n2, n - n2, &br, &bi);
        *rr = ar + br; *ri = ai + bi;
    }
}
static void csum_f32(const float *x, long n, float *rr, float *ri) /* n complex elements */
{
    const long B = 8192;
    cpairwise_chunk_f32(x, 2 * (n < B ? n : B), rr, ri);
    for (long st = B; st < n; st += B) {
        float cr, ci;
        cpairwise_chunk_f32(x + 2 * st, 2 * ((n - st) < B ? (n - st) : B), &cr, &ci);
        *rr += cr; *ri += ci;
    }
}
/* np.var(complex64 array) -> float32 (numpy _methods._var): mean = add.reduce / n (true_divide), x = arr - mean, then the
 * fast path for built-in complex types: view as float pairs, square every float, add the two squares (three separately
 * rounded float32 operations, no fused multiply-add), ret = add.reduce(.) / n.  (Found by fuzzing against the reference:
 * fma(xr, xr, xi*xi) — the complex-multiply form — agrees on ~92 % of frames only.) */
static float var_c64(const float *z, long n, float *tmp)
{
    float sr, si;
    csum_f32(z, n, &sr, &si);
    const float mr = sr / (float)n, mi = si / (float)n;
    for (long i = 0; i < n; i++) {
        const float dr = z[2 * i] - mr, di = z[2 * i + 1] - mi;
        tmp[i] = (dr * dr) + (di * di);
    }
    return pss_o_pairwise_sum_f32(tmp, n) / (float)n;
}

/* iq_correction — signal_processing.py:46-80, complex64 in, complex64 out, every step in float32 as NumPy 2.2
 * evaluates it (NEP 50: the Python scalars 2, 1, 1j are weak; complex64 / float32-scalar multiplies by the
 * reciprocal; complex64 * float32-scalar is a plain per-component multiply). */
void pss_o_iq_correction(const float *iq, int n, float *out)
{
    float *c = (float *)malloc(sizeof(float) * 2 * (size_t)n), *t = (float *)malloc(sizeof(float) * (size_t)n);
    float sr, si;
    csum_f32(iq, n, &sr, &si);                                   /* :48 np.mean(samples) */
    const float mr = sr / (float)n, mi = si / (float)n;
    for (int i = 0; i < n; i++) { c[2 * i] = iq[2 * i] - mr; c[2 * i + 1] = iq[2 * i + 1] - mi; }
    const float input_power = var_c64(c, n, t);                  /* :49 */
    for (int i = 0; i < n; i++) t[i] = iq[2 * i + 1] * iq[2 * i + 1];
    const float qa = sqrtf(2.0f * (pss_o_pairwise_sum_f32(t, n) / (float)n));   /* :52 */
    const float scl = 1.0f / qa;                                 /* :55 samples / q_amplitude */
    for (int i = 0; i < n; i++) { const float is = iq[2 * i] * scl; t[i] = is * is; }
    const float alpha = sqrtf(2.0f * (pss_o_pairwise_sum_f32(t, n) / (float)n)); /* :60 */
    for (int i = 0; i < n; i++) t[i] = (iq[2 * i] * scl) * (iq[2 * i + 1] * scl);
    const float sinphi = (2.0f / alpha) * (pss_o_pairwise_sum_f32(t, n) / (float)n); /* :61 */
    const float cosphi = sqrtf(1.0f - sinphi * sinphi);          /* :64 */
    const float ia = 1.0f / alpha, qa2 = -sinphi / alpha, sc = 1.0f / cosphi;
    for (int i = 0; i < n; i++) {                                /* :67-71 */
        const float is = iq[2 * i] * scl, qs = iq[2 * i + 1] * scl;
        const float i_new = ia * is, q_new = qa2 * is + qs;
        /* i_new + 1j*q_new: (0+1j)*(q+0j) = (fma(0,q,-(1*0)), fma(0,0,1*q)); adding (i_new + 0j) normalises -0 */
        const float jr = fmaf(0.0f, q_new, -0.0f), ji = fmaf(0.0f, 0.0f, q_new);
        c[2 * i] = (i_new + jr) * sc;
        c[2 * i + 1] = (0.0f + ji) * sc;
    }
    const float v2 = var_c64(c, n, t);                           /* :80 */
    const float g = sqrtf(input_power / v2);
    for (int i = 0; i < 2 * n; i++) out[i] = c[i] * g;
    free(c); free(t);
}

/* ------------------------------------------------------------------------------------------------
 * float64 FFT (iterative radix-2, table twiddles).  np.fft.fft is pocketfft; results agree to
 * ~1e-15 relative, the spectrum is tolerance-checked (north star: 1e-4 relative on dB).
 * ---------------------------------------------------------------------------------------------- */
static void fft_f64(double *re, double *im, int n)
{
    int lg = 0;
    while ((1 << lg) < n) lg++;
    for (int i = 0; i < n; i++) { /* bit reversal */
        int j = 0;
        for (int b = 0; b < lg; b++) j |= ((i >> b) & 1) << (lg - 1 - b);
        if (j > i) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    /* the twiddles of the most recent length, kept per thread (the batched drivers transform thousands of frames of one length: the
     * cos / sin of every frame were a third of compute_fft's time); same values, same arithmetic */
    static __thread double *wr = NULL, *wi = NULL;
    static __thread int w_n = 0;
    if (w_n != n) {
        free(wr); free(wi);
        wr = (double *)malloc(sizeof(double) * (n / 2 + 1));
        wi = (double *)malloc(sizeof(double) * (n / 2 + 1));
        for (int k = 0; k < n / 2; k++) {
            double ang = -2.0 * M_PI * (double)k / (double)n;
            wr[k] = cos(ang);
            wi[k] = sin(ang);
        }
        w_n = n;
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1, step = n / len;
        for (int i = 0; i < n; i += len)
            for (int k = 0; k < half; k++) {
                double cr = wr[k * step], ci = wi[k * step];
                double xr = re[i + k + half], xi = im[i + k + half];
                double tr = xr * cr - xi * ci, ti = xr * ci + xi * cr;
                re[i + k + half] = re[i + k] - tr;
                im[i + k + half] = im[i + k] - ti;
                re[i + k] += tr;
                im[i + k] += ti;
            }
    }
}

/* Any length: powers of two go to fft_f64; other lengths (np.fft.fft takes any n; the sweep driver reads int(0.1 fs)
 * samples, pyspecsdr.py:1026) use Bluestein's chirp-z identity over a power-of-two circular convolution. */
static void fft_any_f64(double *re, double *im, int n)
{
    if (n <= 1) return;
    if (!(n & (n - 1))) { fft_f64(re, im, n); return; }
    int M = 1;
    while (M < 2 * n - 1) M <<= 1;
    double *ar = (double *)calloc(M, sizeof(double)), *ai = (double *)calloc(M, sizeof(double));
    double *br = (double *)calloc(M, sizeof(double)), *bi = (double *)calloc(M, sizeof(double));
    double *cr = (double *)malloc(sizeof(double) * n), *ci = (double *)malloc(sizeof(double) * n);
    for (int k = 0; k < n; k++) {
        long long q = ((long long)k * k) % (2LL * n);
        double ang = -M_PI * (double)q / (double)n;
        cr[k] = cos(ang); ci[k] = sin(ang);
        ar[k] = re[k] * cr[k] - im[k] * ci[k];
        ai[k] = re[k] * ci[k] + im[k] * cr[k];
        br[k] = cr[k]; bi[k] = -ci[k];
        if (k) { br[M - k] = br[k]; bi[M - k] = bi[k]; }
    }
    fft_f64(ar, ai, M);
    fft_f64(br, bi, M);
    for (int k = 0; k < M; k++) { /* conj(A B), so that a forward transform inverts */
        double pr = ar[k] * br[k] - ai[k] * bi[k], pi = ar[k] * bi[k] + ai[k] * br[k];
        ar[k] = pr; ai[k] = -pi;
    }
    fft_f64(ar, ai, M);
    for (int k = 0; k < n; k++) {
        double zr = ar[k] / (double)M, zi = -ai[k] / (double)M;
        re[k] = zr * cr[k] - zi * ci[k];
        im[k] = zr * ci[k] + zi * cr[k];
    }
    free(ar); free(ai); free(br); free(bi); free(cr); free(ci);
}

/* compute_fft — signal_processing.py:243-264:
 *   window = np.hamming(N) (:246); samples*window -> complex128 (:247); fftshift(fft()) (:250);
 *   10*log10(abs(fft)**2 + 1e-10) (:262). */
void pss_o_compute_fft(const float *iq, int n, double *db)
{
    double *re = (double *)malloc(sizeof(double) * n), *im = (double *)malloc(sizeof(double) * n);
    static __thread double *win = NULL;     /* np.hamming(n) of the most recent length, per thread */
    static __thread int win_n = 0;
    if (win_n != n) {
        free(win);
        win = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
        for (int i = 0; i < n; i++) win[i] = (n == 1) ? 1.0 : 0.54 - 0.46 * cos(2.0 * M_PI * (double)i / (double)(n - 1));
        win_n = n;
    }
    for (int i = 0; i < n; i++) {
        const double w = win[i];
        re[i] = (double)iq[2 * i] * w;
        im[i] = (double)iq[2 * i + 1] * w;
    }
    fft_any_f64(re, im, n);
    for (int k = 0; k < n; k++) {
        int src = (k + (n + 1) / 2) % n; /* np.fft.fftshift: out[(k + n // 2) % n] = X[k] */
        double a = hypot(re[src], im[src]);
        db[k] = 10.0 * log10(a * a + 1e-10);
    }
    free(re);
    free(im);
}

/* compute_fft on a complex128 buffer (`samples * window` is then a float64 product of float64 samples, :247): the same statements, no narrowing. */
void pss_o_compute_fft_c128(const double *iq, int n, double *db)
{
    double *re = (double *)malloc(sizeof(double) * n), *im = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) {
        const double w = (n == 1) ? 1.0 : 0.54 - 0.46 * cos(2.0 * M_PI * (double)i / (double)(n - 1));
        re[i] = iq[2 * i] * w;
        im[i] = iq[2 * i + 1] * w;
    }
    fft_any_f64(re, im, n);
    for (int k = 0; k < n; k++) {
        int src = (k + (n + 1) / 2) % n;
        double a = hypot(re[src], im[src]);
        db[k] = 10.0 * log10(a * a + 1e-10);
    }
    free(re);
    free(im);
}

/* ------------------------------------------------------------------------------------------------
 * classify_signal — signal_processing.py:296-322 (helpers :267-293), SURVEY §8(f) #3.
 * The reference calls `welch` without importing it (NameError on every call, App. C2); this is the function as it
 * runs with `from scipy.signal import welch` in place (the fixture script binds it the same way).
 *   freqs, psd = welch(samples, fs=sample_rate, nperseg=1024)                                  :299
 *     SciPy 1.15.3 scipy/signal/_spectral_py.py: welch :454 -> csd :603 -> _spectral_helper :1863 -> _fft_helper :2158.
 *     complex64 input => everything in complex64: periodic Hann window cast to complex64 (:2083), segments of 1024
 *     every 512 samples (noverlap = nperseg // 2, no padding, no boundary extension), per-segment mean removed
 *     (detrend 'constant'), two-sided, scale = 1 / (fs * sum(win * win)) (:2087), mean over segments (:603 ff.).
 *     The segment FFT is scipy.fft (pocketfft) in SINGLE precision; it is restated in float64 here — not bit-pinned:
 *     the difference is the reference's own float32 FFT noise (~1e-7 of the peak bin), far below the +1e-10 floor
 *     of the two features that read the PSD.  PSD-derived outputs are tolerance-checked, see tests.
 *   estimate_bandwidth :267-280 — note freqs is in FFT order (np.fft.fftfreq), so the "bandwidth" is
 *     freqs[last bin above max-20 dB] - freqs[first such bin] in THAT order (often one negative bin width).
 *   estimate_modulation_index :283-293 — float32 throughout, restated operation by operation (np.abs, np.angle ->
 *     SVML atan2f, np.unwrap's float32 arithmetic incl. the sequential float32 cumsum, np.diff, np.var) => bit-exact.
 *   spectral flatness :304: exp(mean(log(psd + 1e-10))) / mean(psd), float32 (libm logf/expf here; NumPy uses its
 *     own SIMD log/exp: <= 1 ulp apart).
 * Reads shorter than 1024 samples: SciPy shrinks nperseg to n — one segment, Hann window and FFT of length n.
 * Returns the label (PSS_O_CLS_*), or -1 when n < 1.  psd_out (may be NULL): float32[min(n, 1024)] in FFT order.
 * ---------------------------------------------------------------------------------------------- */
static float var_f32(const float *a, long n, float *tmp) /* np.var of a float32 vector (_methods._var) */
{
    float mean = pss_o_pairwise_sum_f32(a, n) / (float)n;
    for (long i = 0; i < n; i++) { float d = a[i] - mean; tmp[i] = d * d; }
    return pss_o_pairwise_sum_f32(tmp, n) / (float)n;
}

static float np_modf32(float a, float b) /* npy_remainderf */
{
    float m = fmodf(a, b);
    if (m != 0.0f) { if ((b < 0) != (m < 0)) m += b; }
    else m = copysignf(0.0f, b);
    return m;
}

float pss_o_modulation_index(const float *iq, long n)
{
    float *a = (float *)malloc(sizeof(float) * n), *p = (float *)malloc(sizeof(float) * n), *t = (float *)malloc(sizeof(float) * n);
    for (long i = 0; i < n; i++) {
        a[i] = pss_o_cabsf(iq[2 * i], iq[2 * i + 1]);          /* :286 np.abs(complex64) */
        p[i] = pss_o_atan2f(iq[2 * i + 1], iq[2 * i]);         /* :287 np.angle -> arctan2(imag, real) */
    }
    float amp_var = var_f32(a, n, t);                            /* :290 */
    /* np.unwrap (numpy/lib/_function_base_impl.py), float32: period and +-period/2 are Python floats = weak scalars */
    const float PI32 = (float)M_PI, TWOPI32 = (float)(2.0 * M_PI), NPI32 = (float)(-M_PI);
    float cs = 0.0f, prev_up = n > 0 ? p[0] : 0.0f;
    float *d = a; /* reuse: d has n-1 entries */
    for (long i = 0; i + 1 < n; i++) {
        float dd = p[i + 1] - p[i];
        float ddmod = np_modf32(dd - NPI32, TWOPI32) + NPI32;
        if (ddmod == NPI32 && dd > 0.0f) ddmod = PI32;
        float corr = ddmod - dd;
        if (fabsf(dd) < PI32) corr = 0.0f;
        cs = cs + corr;                                          /* ph_correct.cumsum(): sequential float32 */
        float up = p[i + 1] + cs;
        d[i] = up - prev_up;                                     /* :291 np.diff(phase_env) */
        prev_up = up;
    }
    float phase_var = n > 1 ? var_f32(d, n - 1, t) : NAN;        /* np.var of an empty array is nan */
    free(a); free(p); free(t);
    return phase_var / (amp_var + (float)1e-10);                 /* :293 */
}

void pss_o_hann_f32(float *w, int n) /* scipy.signal.get_window('hann', n) -> general_cosine(n + 1, [0.5, 0.5])[:-1], cast to float32 */
{
    if (n <= 1) { if (n == 1) w[0] = 1.0f; return; }             /* windows/_windows.py _len_guards: M <= 1 -> ones(M) */
    const double start = -M_PI, step = (M_PI - (-M_PI)) / (double)n;  /* np.linspace(-pi, pi, n + 1) */
    for (int i = 0; i < n; i++) {
        double fac = (double)i * step + start;
        w[i] = (float)(0.5 + 0.5 * cos(fac));
    }
}

void pss_o_hann1024_f32(float *w) { pss_o_hann_f32(w, 1024); }

int pss_o_classify(const float *iq, long n, double fs, double *bw_out, float *mi_out, float *flat_out, float *psd_out)
{
    if (n < 1) return -1;
    /* :299 nperseg=1024; a shorter read makes SciPy take nperseg = n (_spectral_py.py _triage_segments: "nperseg = 1024 is
     * greater than input length ... using nperseg = n"): ONE segment, Hann window of length n, nfft = n (any length) */
    const int NP = n < 1024 ? (int)n : 1024, STEP = NP - NP / 2;
    const long nseg = (n - NP / 2) / STEP;                       /* = (n - 1024) / 512 + 1 for n >= 1024, 1 below */
    float w[1024], w2[2048];
    pss_o_hann_f32(w, NP);
    for (int i = 0; i < NP; i++) { w2[2 * i] = w[i] * w[i]; w2[2 * i + 1] = 0.0f; }
    float sr, si;
    csum_f32(w2, NP, &sr, &si);                                  /* (win*win).sum(), complex64 */
    const float scale = 1.0f / ((float)fs * sr);                 /* 1.0 / (fs * sum): complex64 scalars with zero imaginary parts */
    double acc[1024], re[1024], im[1024];
    for (int k = 0; k < NP; k++) acc[k] = 0.0;
    for (long s = 0; s < nseg; s++) {
        const float *x = iq + 2 * s * STEP;
        float mr, mi;
        csum_f32(x, NP, &mr, &mi);                               /* detrend 'constant': data - mean(data) */
        mr /= (float)NP; mi /= (float)NP;
        for (int i = 0; i < NP; i++) {
            float dr = x[2 * i] - mr, di = x[2 * i + 1] - mi;
            re[i] = (double)(w[i] * dr);                         /* win * segment, complex64 with win.imag == 0 */
            im[i] = (double)(w[i] * di);
        }
        if (NP == 1024) fft_f64(re, im, NP);
        else {                                                   /* short read: plain DFT of length n in float64 */
            double xr[1024], xi[1024];
            memcpy(xr, re, sizeof(double) * NP); memcpy(xi, im, sizeof(double) * NP);
            for (int k = 0; k < NP; k++) {
                double ar = 0.0, ai = 0.0;
                for (int i = 0; i < NP; i++) {
                    const double ang = -2.0 * M_PI * (double)(((long)i * k) % NP) / (double)NP, c = cos(ang), sn = sin(ang);
                    ar += xr[i] * c - xi[i] * sn;
                    ai += xr[i] * sn + xi[i] * c;
                }
                re[k] = ar; im[k] = ai;
            }
        }
        for (int k = 0; k < NP; k++) acc[k] += re[k] * re[k] + im[k] * im[k];
    }
    float psd[1024], tmp[1024];
    for (int k = 0; k < NP; k++) psd[k] = (float)(acc[k] / (double)nseg * (double)scale);
    if (psd_out) memcpy(psd_out, psd, sizeof(float) * NP);
    /* estimate_bandwidth(psd, freqs, -20) :267-280 */
    float mx = -INFINITY;
    for (int k = 0; k < NP; k++) { tmp[k] = 10.0f * pss_o_log10f_np(psd[k] + (float)1e-10); if (tmp[k] > mx || tmp[k] != tmp[k]) mx = tmp[k]; }
    const float thr = mx + -20.0f;
    int first = -1, last = -1;
    for (int k = 0; k < NP; k++) if (tmp[k] > thr) { if (first < 0) first = k; last = k; }
    double bw = 0.0;
    if (first >= 0) {
        const double val = 1.0 / ((double)NP * (1.0 / fs));      /* np.fft.fftfreq(n, d): integers * (1 / (n d)) */
        const int npos = (NP - 1) / 2 + 1;                       /* bins 0 .. npos-1 are >= 0, the rest k - n */
        const double f0 = (double)(first < npos ? first : first - NP) * val, f1 = (double)(last < npos ? last : last - NP) * val;
        bw = f1 - f0;
    }
    const float mi_v = pss_o_modulation_index(iq, n);
    for (int k = 0; k < NP; k++) tmp[k] = logf(psd[k] + (float)1e-10);
    const float gm = expf(pss_o_pairwise_sum_f32(tmp, NP) / (float)NP);
    const float flat = gm / (pss_o_pairwise_sum_f32(psd, NP) / (float)NP);
    if (bw_out) *bw_out = bw;
    if (mi_out) *mi_out = mi_v;
    if (flat_out) *flat_out = flat;
    /* :307-322; np.float32 against a Python float compares in float32 (NEP 50) */
    if (bw > 150e3) return mi_v > 0.8f ? PSS_O_CLS_FM_BROADCAST : PSS_O_CLS_UNKNOWN;
    if (8e3 <= bw && bw <= 16e3) return mi_v < 0.3f ? PSS_O_CLS_NARROW_FM : PSS_O_CLS_UNKNOWN;
    /* :314 (8e3 <= bw <= 10e3 -> AM_BROADCAST) can never be reached: the branch above already took that range */
    if (2e3 <= bw && bw <= 3e3) return flat < 0.2f ? PSS_O_CLS_SSB : PSS_O_CLS_UNKNOWN;
    if (flat > 0.7f) return PSS_O_CLS_DIGITAL;
    return PSS_O_CLS_UNKNOWN;
}

/* ------------------------------------------------------------------------------------------------
 * decode_morse, front half — decoders.py:149-165 (called with the default threshold = -20 dB, pyspecsdr.py:573):
 *   envelope = np.abs(samples); envelope /= np.max(envelope); envelope_db = 20*np.log10(envelope + 1e-10)   (float32)
 *   signals = envelope_db > threshold; transitions = np.diff(signals.astype(int)); rise/fall = where(== +1 / -1)
 * NumPy's float32 log10 under AVX512_SKX dispatch is SVML's __svml_log10f16.  Only the comparison matters: probing
 * the reference environment around 0.1 (+-2000 ulp, monotone there) shows 20*log10(v) > -20 exactly for
 * v >= 0x3dcccccf (SVML returns -1.0 for 0x3dccccce, where a correctly rounded log10f gives -0.99999994).
 * Returns the number of transitions written is min(count, cap); *n_rise / *n_fall are the true counts.
 * ---------------------------------------------------------------------------------------------- */
void pss_o_morse_edges(const float *iq, long n, int32_t *rise, int32_t *fall, long cap, long *n_rise, long *n_fall)
{
    const float CUT = u2f(0x3dcccccfu);
    float mx = -INFINITY;
    int has_nan = 0;
    for (long i = 0; i < n; i++) {
        float e = pss_o_cabsf(iq[2 * i], iq[2 * i + 1]);
        if (e != e) has_nan = 1;
        if (e > mx) mx = e;
    }
    if (has_nan) mx = NAN;
    long nr = 0, nf = 0;
    int prev = 0;
    for (long i = 0; i < n; i++) {
        float v = pss_o_cabsf(iq[2 * i], iq[2 * i + 1]) / mx + (float)1e-10;
        int sgn = v >= CUT; /* false for NaN */
        if (i > 0) {
            if (!prev && sgn) { if (nr < cap) rise[nr] = (int32_t)(i - 1); nr++; }
            if (prev && !sgn) { if (nf < cap) fall[nf] = (int32_t)(i - 1); nf++; }
        }
        prev = sgn;
    }
    *n_rise = nr; *n_fall = nf;
}

/* decode_morse's mask at ANY threshold (decoders.py:149-156): envelope_db = 20 * np.log10(envelope + 1e-10) in float32 (NumPy's SVML log10
 * model above), signals = envelope_db > threshold with the Python scalar cast to float32 (NEP 50: a weak scalar beside a float32 array). */
void pss_o_morse_edges_thr(const float *iq, long n, double threshold, int32_t *rise, int32_t *fall, long cap, long *n_rise, long *n_fall)
{
    const float thr = (float)threshold;
    float mx = -INFINITY;
    int has_nan = 0;
    for (long i = 0; i < n; i++) {
        float e = pss_o_cabsf(iq[2 * i], iq[2 * i + 1]);
        if (e != e) has_nan = 1;
        if (e > mx) mx = e;
    }
    if (has_nan) mx = NAN;
    long nr = 0, nf = 0;
    int prev = 0;
    for (long i = 0; i < n; i++) {
        float v = pss_o_cabsf(iq[2 * i], iq[2 * i + 1]) / mx + (float)1e-10;
        int sgn = 20.0f * pss_o_log10f_np(v) > thr; /* false for NaN */
        if (i > 0) {
            if (!prev && sgn) { if (nr < cap) rise[nr] = (int32_t)(i - 1); nr++; }
            if (prev && !sgn) { if (nf < cap) fall[nf] = (int32_t)(i - 1); nf++; }
        }
        prev = sgn;
    }
    *n_rise = nr; *n_fall = nf;
}

static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* pyspecsdr.py:2278-2283: np.convolve(freq_data, ones(5)/5, 'valid'); thr = median - 10; clamp below. */
void pss_o_postprocess(const double *db, int n, double *out)
{
    int m = n - 4;
    for (int i = 0; i < m; i++) {
        double acc = 0.0;
        for (int k = 0; k < 5; k++) acc += db[i + k] * 0.2;
        out[i] = acc;
    }
    double *tmp = (double *)malloc(sizeof(double) * m);
    memcpy(tmp, out, sizeof(double) * m);
    qsort(tmp, m, sizeof(double), cmp_double);
    double med = (m & 1) ? tmp[m / 2] : 0.5 * (tmp[m / 2 - 1] + tmp[m / 2]);
    free(tmp);
    double thr = med - 10.0;
    for (int i = 0; i < m; i++)
        if (out[i] < thr) out[i] = thr;
}

/* measure_signal_power — signal_processing.py:325-328, float32: mean(abs(x)**2) then 10*log10(p + 1e-10). */
float pss_o_power_db(const float *iq, int n)
{
    float *s = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        float m = pss_o_cabsf(iq[2 * i], iq[2 * i + 1]);
        s[i] = m * m;
    }
    float p = pss_o_pairwise_sum_f32(s, n) / (float)n;
    free(s);
    float t = p + 1e-10f;
    return 10.0f * pss_o_log10f_ref(t);
}

/* inline scanner — pyspecsdr.py:2542-2552.  np.fft.fft on complex64 returns complex64, but NumPy 2.2 computes it in DOUBLE
 * and rounds the result (checked: identical bits to fft(x.astype(complex128)).astype(complex64) for n = 8 .. 240000), so the
 * spectrum is round_f32(double transform) and everything behind it is float32: np.abs(complex64) (pss_o_cabsf), ** 2,
 * + 1e-10 (weak scalar -> float32), np.log10 float32 (SVML, pss_o_log10f_np), * 10.  This restatement's own float64
 * transform differs from pocketfft's by ~1e-16 relative, which the rounding to float32 hides except when a component sits
 * within that distance of a rounding boundary (~1e-8 of the values). */
int pss_o_scan_slice(const float *iq, int n, double fs, float *db, float *peak, double *bw)
{
    double *re = (double *)malloc(sizeof(double) * n), *im = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) { re[i] = iq[2 * i]; im[i] = iq[2 * i + 1]; }
    fft_any_f64(re, im, n);
    float pk = -INFINITY;
    for (int k = 0; k < n; k++) {
        int src = (k + (n + 1) / 2) % n;
        float a = pss_o_cabsf((float)re[src], (float)im[src]);
        float p = a * a + 1e-10f;
        db[k] = 10.0f * pss_o_log10f_np(p);
        if (db[k] > pk) pk = db[k];
    }
    int count = 0;
    float thr = pk - 20.0f;
    for (int k = 0; k < n; k++) count += db[k] > thr;
    *peak = pk;
    *bw = (double)count * (fs / (double)n);
    free(re);
    free(im);
    return count;
}

/* scan_frequencies' per-read arithmetic — pyspecsdr.py:1049-1057: the same unwindowed spectrum, max_power, and the bins above an
 * ABSOLUTE threshold: bandwidth = np.sum(power_db > threshold) * (sample_rate / len(power_db)). */
int pss_o_scan_threshold(const float *iq, int n, double fs, double threshold_db, float *db, float *peak, double *bw)
{
    pss_o_scan_slice(iq, n, fs, db, peak, bw);
    int count = 0;
    for (int k = 0; k < n; k++) count += db[k] > (float)threshold_db;
    *bw = (double)count * (fs / (double)n);
    return count;
}

/* scipy _sosfilt (Cython, _sosfilt.pyx _sosfilt_float): DF2T cascade, un-fused float64.
 * state z[nsec][2] is updated in place; x filtered in place. */
static void sosfilt_inplace(const double *sos, int nsec, double *x, long n, double *z)
{
    for (long i = 0; i < n; i++) {
        double xc = x[i];
        for (int s = 0; s < nsec; s++) {
            const double *c = sos + 6 * s;
            double xn = (c[0] * xc) + z[2 * s];
            z[2 * s] = ((c[1] * xc) - (c[4] * xn)) + z[2 * s + 1];
            z[2 * s + 1] = (c[2] * xc) - (c[5] * xn);
            xc = xn;
        }
        x[i] = xc;
    }
}

/* bandpass_filter — signal_processing.py:34-42 given its SOS table: sosfilt(sos, data), zero initial state. */
void pss_o_sosfilt(const double *sos, int nsec, const double *x, long n, double *y)
{
    double z[32] = {0};
    memcpy(y, x, sizeof(double) * n);
    sosfilt_inplace(sos, nsec, y, n, z);
}

/* numpy add.reduce float64 over n <= 8192 elements of a[i]^2 (DOUBLE_pairwise_sum, same tree as the float32 one). */
static double pairwise_sq_f64(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i] * a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j] * a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j] * a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i] * a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sq_f64(a, n2) + pairwise_sq_f64(a + n2, n - n2);
    }
}

/* decode_afsk — decoders.py:94-112: two butter(5) band-passes (1100-1300 Hz, 2100-2300 Hz; bandpass_filter,
 * signal_processing.py:34-42), energy of each band over every bit period of int(fs/1200) samples (np.sum of the squares,
 * float64 pairwise), bit = e2200 > e1200.  Returns the number of bits = len(range(0, n - window, window)). */
int pss_o_afsk_bits(const double *x, int n, double fs, const double *sos1200, const double *sos2200, int nsec, uint8_t *bits)
{
    const int w = (int)(fs / 1200.0);
    if (w < 1 || n - w <= 0) return 0;
    double *f1 = (double *)malloc(sizeof(double) * n), *f2 = (double *)malloc(sizeof(double) * n);
    pss_o_sosfilt(sos1200, nsec, x, n, f1);
    pss_o_sosfilt(sos2200, nsec, x, n, f2);
    int nb = 0;
    for (int i = 0; i < n - w; i += w) {
        double e1 = 0.0, e2 = 0.0;
        for (int st = 0; st < w; st += 8192) {  /* windows longer than the ufunc buffer: chunk sums added in order */
            const int len = (w - st) < 8192 ? (w - st) : 8192;
            const double c1 = pairwise_sq_f64(f1 + i + st, len), c2 = pairwise_sq_f64(f2 + i + st, len);
            e1 = st ? e1 + c1 : c1;
            e2 = st ? e2 + c2 : c2;
        }
        bits[nb++] = e2 > e1;
    }
    free(f1); free(f2);
    return nb;
}

/* 65-tap FIR with zero initial state: scipy.signal.lfilter(taps, 1.0, x) FIR branch
 * = np.convolve(taps, x)[:len(x)] = multiarray.correlate(x, taps[::-1], 'full') whose inner product is
 * DOUBLE_dot -> cblas_ddot, i.e. OpenBLAS kernel/x86_64/ddot.c + ddot_microk_skylakex-2.c on the
 * AVX-512 host that produced the goldens.  Its accumulation order, restated (all multiply-adds fused):
 *   n1 = n & -16 elements go through 4 accumulators of 8 lanes (32 per step), folded 8->4 lanes, then
 *   4 accumulators of 4 lanes (16 per step); s = ((a0+a1)+a2)+a3 per lane; dot = (s0+s2)+(s1+s3);
 *   the n - n1 tail elements are added one by one with fma.
 * xw[j]*yw[j], j = 0..n-1.  Verified bit-exact on every output (edges included) of tests/golden/nfm.npz. */
static double ddot_skx(const double *xw, const double *yw, int n)
{
    int n1 = n & -16, i = 0;
    double dot = 0.0;
    if (n1) {
        double a5[4][8] = {{0}}, a[4][4];
        int n32 = n1 & ~31;
        for (; i < n32; i += 32)
            for (int k = 0; k < 4; k++)
                for (int l = 0; l < 8; l++) a5[k][l] = fma(xw[i + 8 * k + l], yw[i + 8 * k + l], a5[k][l]);
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < 4; l++) a[k][l] = a5[k][l] + a5[k][l + 4];
        for (; i < n1; i += 16)
            for (int k = 0; k < 4; k++)
                for (int l = 0; l < 4; l++) a[k][l] = fma(xw[i + 4 * k + l], yw[i + 4 * k + l], a[k][l]);
        double s[4];
        for (int l = 0; l < 4; l++) s[l] = ((a[0][l] + a[1][l]) + a[2][l]) + a[3][l];
        dot = (s[0] + s[2]) + (s[1] + s[3]);
    }
    for (; i < n; i++) dot = fma(yw[i], xw[i], dot);
    return dot;
}

/* Real part of the complex dot product behind lfilter(taps, 1.0, complex_samples) at
 * signal_processing.py:204/:209: CDOUBLE_dot -> cblas_zdotu = OpenBLAS kernel/x86_64/zdot.c +
 * zdot_microk_haswell-2.c (also used on SKYLAKEX).  The taps are real, so the imag*imag lane only ever
 * accumulates +-0 and real = sum xr*tr in this order: n & -8 elements over 4 ymm accumulators holding
 * 2 complex each (8 complex per step, fused), c_p = (a0+a1)+(a2+a3) per slot p, dot = c0 + c1, then the
 * tail elements one by one with fma.  Verified bit-exact against scipy on tests/golden/am_ssb.npz inputs. */
static double zdot_re_skx(const double *xw, const double *yw, int n)
{
    int n1 = n & -8, i = 0;
    double dot = 0.0;
    if (n1) {
        double acc[4][2] = {{0}};
        for (; i < n1; i += 8)
            for (int a = 0; a < 4; a++)
                for (int p = 0; p < 2; p++) acc[a][p] = fma(xw[i + 2 * a + p], yw[i + 2 * a + p], acc[a][p]);
        double c0 = (acc[0][0] + acc[1][0]) + (acc[2][0] + acc[3][0]);
        double c1 = (acc[0][1] + acc[1][1]) + (acc[2][1] + acc[3][1]);
        dot = c0 + c1;
    }
    for (; i < n; i++) dot = fma(xw[i], yw[i], dot);
    return dot;
}

static inline double fir65z_at(const double *taps, const double *taps_rev, const double *x, long i, long m)
{
    if (m <= 65) {
        double xr[65];
        for (long j = 0; j <= i; j++) xr[j] = x[i - j];
        return zdot_re_skx(taps, xr, (int)i + 1);
    }
    if (i >= 64) return zdot_re_skx(x + i - 64, taps_rev, 65);
    return zdot_re_skx(x, taps_rev + 64 - i, (int)i + 1);
}

/* taps_rev[j] = taps[64 - j].  np.convolve(taps, x) swaps its operands only when len(x) > 65, so for
 * frames of <= 65 samples the dot runs over ascending TAP index (taps[j]*x[i-j]) instead of ascending
 * sample index; m = len(x). */
static inline double fir65_at(const double *taps, const double *taps_rev, const double *x, long i, long m)
{
    if (m <= 65) {
        double xr[65];
        for (long j = 0; j <= i; j++) xr[j] = x[i - j];
        return ddot_skx(taps, xr, (int)i + 1);
    }
    if (i >= 64) return ddot_skx(x + i - 64, taps_rev, 65);
    return ddot_skx(x, taps_rev + 64 - i, (int)i + 1);
}

int pss_o_demod_nfm(const float *iq, int n, double fs, int q, const double *taps, const double *sos,
                    const double *zi, double *audio, float *disc_out, double *fir_out)
{
    const int NSEC = 4, EDGE = 27; /* sosfiltfilt: ntaps = 2*4+1 = 9, edge = 3*9 = 27 */
    long M = (long)n - 1;
    if (M <= EDGE) return -1; /* ValueError: length of the input vector must be greater than padlen */
    /* :94  np.angle(samples[1:] * np.conj(samples[:-1]))  — complex64 multiply in FMA form, operand
     *      order first = samples[1:], second = conj(samples[:-1]); NumPy temporary elision swaps the
     *      operands once the conj temporary reaches 262144 bytes (N-1 >= 32768) — SURVEY App. A2.1. */
    /* :97  demod * (fs/(2*pi)) — python float is weak -> float32 scalar */
    float kscale = (float)(fs / (2.0 * M_PI));
    double *u = (double *)malloc(sizeof(double) * M);
    double *d64 = (double *)malloc(sizeof(double) * M);
    int swapped = (M * 8 >= 262144);
    for (long i = 0; i < M; i++) {
        float aI = iq[2 * (i + 1)], aQ = iq[2 * (i + 1) + 1];
        float c = iq[2 * i], d = -iq[2 * i + 1];
        float re = fmaf(aI, c, -(aQ * d));
        float im = swapped ? fmaf(aQ, c, aI * d) : fmaf(aI, d, aQ * c);
        float th = pss_o_atan2f(im, re);
        float dv = th * kscale;
        if (disc_out) disc_out[i] = dv;
        d64[i] = (double)dv;
    }
    /* :105-108 firwin + lfilter */
    double tr[65];
    for (int j = 0; j < 65; j++) tr[j] = taps[64 - j];
    for (long i = 0; i < M; i++) u[i] = fir65_at(taps, tr, d64, i, M);
    if (fir_out) memcpy(fir_out, u, sizeof(double) * M);
    /* :111-112 decimate(filtered, q) -> sosfiltfilt(cheby1 sos) then [::q]  (scipy _signaltools.py:4831, :4718) */
    long L = M + 2 * EDGE;
    double *ext = (double *)malloc(sizeof(double) * L);
    for (int i = 0; i < EDGE; i++) ext[i] = 2.0 * u[0] - u[EDGE - i];               /* odd_ext, _arraytools.py:57 */
    memcpy(ext + EDGE, u, sizeof(double) * M);
    for (int i = 0; i < EDGE; i++) ext[EDGE + M + i] = 2.0 * u[M - 1] - u[M - 2 - i];
    double z[8];
    for (int i = 0; i < 2 * NSEC; i++) z[i] = zi[i] * ext[0];
    sosfilt_inplace(sos, NSEC, ext, L, z);
    for (long i = 0; i < L / 2; i++) { double t = ext[i]; ext[i] = ext[L - 1 - i]; ext[L - 1 - i] = t; }
    for (int i = 0; i < 2 * NSEC; i++) z[i] = zi[i] * ext[0]; /* y_0 = last forward output */
    sosfilt_inplace(sos, NSEC, ext, L, z);
    for (long i = 0; i < L / 2; i++) { double t = ext[i]; ext[i] = ext[L - 1 - i]; ext[L - 1 - i] = t; }
    int n_out = (int)((M + q - 1) / q);
    double mx = 0.0;
    int has_nan = 0;
    for (int j = 0; j < n_out; j++) {
        double v = ext[EDGE + (long)j * q];
        audio[j] = v;
        double av = fabs(v);
        if (av != av) has_nan = 1;
        if (av > mx) mx = av;
    }
    if (has_nan) mx = NAN; /* np.max propagates NaN */
    /* :115 audio / np.max(np.abs(audio)) * 0.95 */
    for (int j = 0; j < n_out; j++) audio[j] = (audio[j] / mx) * 0.95;
    free(u);
    free(d64);
    free(ext);
    return n_out;
}

/* demodulate_am — signal_processing.py:179-195 */
void pss_o_demod_am(const float *iq, int n, const double *sos, int nsec, double *audio)
{
    float *e = (float *)malloc(sizeof(float) * n);
    for (int i = 0; i < n; i++) e[i] = pss_o_cabsf(iq[2 * i], iq[2 * i + 1]);      /* :182 */
    float mu = pss_o_pairwise_sum_f32(e, n) / (float)n;                             /* :185 np.mean (float32) */
    for (int i = 0; i < n; i++) audio[i] = (double)(e[i] - mu);                     /* :185 float32 subtract */
    double z[16] = {0};
    sosfilt_inplace(sos, nsec, audio, n, z);                                        /* :191 -> :42 sosfilt */
    double mx = 0.0;
    int has_nan = 0;
    for (int i = 0; i < n; i++) { double a = fabs(audio[i]); if (a != a) has_nan = 1; if (a > mx) mx = a; }
    if (has_nan) mx = NAN;
    for (int i = 0; i < n; i++) audio[i] = (audio[i] / mx) * 0.95;                  /* :194 */
    free(e);
}

/* demodulate_am on a complex128 buffer (signal_processing.py:179-195 with float64 samples): numpy.abs(complex128) is the same scaled hypot as
 * the complex64 loop, in float64 — mx * sqrt(fma(r, r, 1)), r = mn / mx (probed: 200 000 random values, every bit) —, np.mean the same pairwise
 * tree over float64 (8192-element buffer chunks added up in order; blocks of 128 with 8 accumulators), the subtraction float64. */
double pss_o_cabs(double re, double im)
{
    double a = fabs(re), b = fabs(im);
    double mx = a > b ? a : b, mn = a > b ? b : a;
    if (mx == 0.0) return 0.0;
    double r = mn / mx;
    return mx * sqrt(fma(r, r, 1.0));
}
static double pairwise_chunk_f64(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_chunk_f64(a, n2) + pairwise_chunk_f64(a + n2, n - n2);
    }
}
double pss_o_pairwise_sum_f64(const double *a, long n)
{
    const long B = 8192;
    if (n <= B) return pairwise_chunk_f64(a, n);
    double acc = pairwise_chunk_f64(a, B);
    for (long st = B; st < n; st += B) acc += pairwise_chunk_f64(a + st, (n - st) < B ? (n - st) : B);
    return acc;
}
/* measure_signal_power (signal_processing.py:325-328) of a complex128 buffer, the array part: np.mean(np.abs(samples) ** 2) in float64 — np.abs as
 * above, x ** 2 = x * x, np.mean = the pairwise sum / n.  The caller finishes with the scalar 10 * log10(power + 1e-10) (NumPy's own float64 log10:
 * tests/test_oracle_golden.py applies it with NumPy, as the drop-in shim does).  scratch: n doubles. */
double pss_o_mean_power_c128(const double *iq, int n, double *scratch)
{
    for (int i = 0; i < n; i++) { const double a = pss_o_cabs(iq[2 * i], iq[2 * i + 1]); scratch[i] = a * a; }
    return pss_o_pairwise_sum_f64(scratch, n) / (double)n;
}
void pss_o_demod_am_c128(const double *iq, int n, const double *sos, int nsec, double *audio)
{
    for (int i = 0; i < n; i++) audio[i] = pss_o_cabs(iq[2 * i], iq[2 * i + 1]);   /* :182 */
    const double mu = pss_o_pairwise_sum_f64(audio, n) / (double)n;                /* :185 np.mean (float64) */
    for (int i = 0; i < n; i++) audio[i] = audio[i] - mu;
    double z[16] = {0};
    sosfilt_inplace(sos, nsec, audio, n, z);                                       /* :191 -> :42 sosfilt */
    double mx = 0.0;
    int has_nan = 0;
    for (int i = 0; i < n; i++) { double a = fabs(audio[i]); if (a != a) has_nan = 1; if (a > mx) mx = a; }
    if (has_nan) mx = NAN;
    for (int i = 0; i < n; i++) audio[i] = (audio[i] / mx) * 0.95;                 /* :194 */
}

/* scipy.signal.lfilter(b=[b0], a=[1, a1], x) — _sigtools._linear_filter (lfilter.c.in, double loop), b zero-padded
 * to [b0, 0], zero initial state:  y = z + b0*x;  z = x*0 - y*a1  (un-fused).  In place. */
static void lfilter_1pole(double b0, double a1, double *x, long n)
{
    double z = 0.0;
    for (long i = 0; i < n; i++) {
        const double xi = x[i], y = z + b0 * xi;
        z = (xi * 0.0) - (y * a1);
        x[i] = y;
    }
}

/* scipy.signal.decimate(u, q, zero_phase=True) (_signaltools.py: cheby1(8, 0.05, 0.8/q) -> sosfiltfilt -> [::q]).
 * Same restatement as inside pss_o_demod_nfm.  out[ceil(M/q)]. */
static int decimate_zero_phase(const double *u, long M, int q, const double *sos, const double *zi, double *out)
{
    const int NSEC = 4, EDGE = 27;
    long L = M + 2 * EDGE;
    double *ext = (double *)malloc(sizeof(double) * L);
    for (int i = 0; i < EDGE; i++) ext[i] = 2.0 * u[0] - u[EDGE - i];
    memcpy(ext + EDGE, u, sizeof(double) * M);
    for (int i = 0; i < EDGE; i++) ext[EDGE + M + i] = 2.0 * u[M - 1] - u[M - 2 - i];
    double z[8];
    for (int i = 0; i < 2 * NSEC; i++) z[i] = zi[i] * ext[0];
    sosfilt_inplace(sos, NSEC, ext, L, z);
    for (long i = 0; i < L / 2; i++) { double t = ext[i]; ext[i] = ext[L - 1 - i]; ext[L - 1 - i] = t; }
    for (int i = 0; i < 2 * NSEC; i++) z[i] = zi[i] * ext[0];
    sosfilt_inplace(sos, NSEC, ext, L, z);
    for (long i = 0; i < L / 2; i++) { double t = ext[i]; ext[i] = ext[L - 1 - i]; ext[L - 1 - i] = t; }
    int n_out = (int)((M + q - 1) / q);
    for (int j = 0; j < n_out; j++) out[j] = ext[EDGE + (long)j * q];
    free(ext);
    return n_out;
}

static double max_abs_np(const double *a, long n) /* np.max(np.abs(a)): NaN propagates */
{
    double mx = 0.0;
    int has_nan = 0;
    for (long i = 0; i < n; i++) { double v = fabs(a[i]); if (v != v) has_nan = 1; if (v > mx) mx = v; }
    return has_nan ? NAN : mx;
}

/* demodulate_wfm — signal_processing.py:119-176 (the RDS hooks at :165-174 call undefined names and are swallowed by
 * the try/except).  The "pilot" the reference extracts is sin(unwrap(angle(REAL signal))): the angle
 * of a real float64 is 0 or pi, unwrap leaves that sequence untouched (every jump is exactly +-pi, ph_correct = 0),
 * so pilot[i] is 0.0 or sin(pi) = 0x1.1a62633145c07p-53 and the L-R branch only perturbs the last bits of L and R.
 * It is restated in full all the same, so that those last bits match. */
int pss_o_demod_wfm(const float *iq, int n, int q, const double *lp_sos, const double *pilot_sos,
                    const double *lmr_sos, double alpha, const double *dec_sos, const double *dec_zi, double *left,
                    double *right)
{
    const double SIN_PI = 0x1.1a62633145c07p-53; /* np.sin(np.pi) */
    long M = (long)n - 1;
    if (M < 1) return -1;                  /* np.max of an empty array: ValueError */
    if (q > 1 && M <= 27) return -1;       /* sosfiltfilt padlen ValueError */
    double *d = (double *)malloc(sizeof(double) * M), *lpr = (double *)malloc(sizeof(double) * M);
    double *pil = (double *)malloc(sizeof(double) * M), *lmr = (double *)malloc(sizeof(double) * M);
    int swapped = (M * 8 >= 262144);       /* as in pss_o_demod_nfm (:122 is the same expression as :94) */
    for (long i = 0; i < M; i++) {
        float aI = iq[2 * (i + 1)], aQ = iq[2 * (i + 1) + 1];
        float c = iq[2 * i], dd = -iq[2 * i + 1];
        float re = fmaf(aI, c, -(aQ * dd));
        float im = swapped ? fmaf(aQ, c, aI * dd) : fmaf(aI, dd, aQ * c);
        d[i] = (double)pss_o_atan2f(im, re);                     /* :122, float32 -> float64 inside sosfilt */
    }
    double z[16];
    memcpy(lpr, d, sizeof(double) * M); memset(z, 0, sizeof z);
    sosfilt_inplace(lp_sos, 3, lpr, M, z);                       /* :126 */
    memcpy(pil, d, sizeof(double) * M); memset(z, 0, sizeof z);
    sosfilt_inplace(pilot_sos, 5, pil, M, z);                    /* :129 */
    lfilter_1pole(1.0, -0.99, pil, M);                           /* :130 lfilter([1], [1, -0.99], pilot) */
    memcpy(lmr, d, sizeof(double) * M); memset(z, 0, sizeof z);
    sosfilt_inplace(lmr_sos, 5, lmr, M, z);                      /* :133 */
    for (long i = 0; i < M; i++) {
        const double y = pil[i];
        const double p = (y != y) ? NAN : ((y < 0.0 || (y == 0.0 && signbit(y))) ? SIN_PI : 0.0);
        lmr[i] = lmr[i] * (2.0 * p);                             /* :134 */
    }
    memset(z, 0, sizeof z);
    sosfilt_inplace(lp_sos, 3, lmr, M, z);                       /* :137 */
    for (long i = 0; i < M; i++) {                               /* :140-141 */
        const double a = lpr[i], b = lmr[i];
        lpr[i] = (a + b) / 2.0;
        lmr[i] = (a - b) / 2.0;
    }
    lfilter_1pole(1.0 - alpha, -alpha, lpr, M);                  /* :144-149 de-emphasis */
    lfilter_1pole(1.0 - alpha, -alpha, lmr, M);
    int n_out;
    if (q > 1) {                                                 /* :152-155 */
        n_out = decimate_zero_phase(lpr, M, q, dec_sos, dec_zi, left);
        decimate_zero_phase(lmr, M, q, dec_sos, dec_zi, right);
    } else {
        n_out = (int)M;
        memcpy(left, lpr, sizeof(double) * M);
        memcpy(right, lmr, sizeof(double) * M);
    }
    const double ml = max_abs_np(left, n_out), mr = max_abs_np(right, n_out);
    const double mx = (mr > ml) ? mr : ml;                       /* :158 python max(a, b): b only if b > a */
    for (int j = 0; j < n_out; j++) { left[j] /= mx; right[j] /= mx; }
    free(d); free(lpr); free(pil); free(lmr);
    return n_out;
}

/* demodulate_ssb — signal_processing.py:198-217 (USB and LSB branches are the same code) */
void pss_o_demod_ssb(const float *iq, int n, const double *taps, double *audio) { pss_o_demod_ssb_ex(iq, n, taps, audio, 1); }

/* with_hilbert = 0: the chain without :205 / :210 (what the library's option "ssb_hilbert" = 0 computes) */
/* the chain on the samples' real parts r[n] (float64): the complex FIR's real part only ever reads those (real taps) */
static void demod_ssb_real(const double *r, int n, const double *taps, double *audio, int with_hilbert)
{
    double tr[65];
    for (int j = 0; j < 65; j++) tr[j] = taps[64 - j];
    for (int i = 0; i < n; i++) audio[i] = fir65z_at(taps, tr, r, i, n);                    /* :204/:209 real part */
    if (with_hilbert && n >= 2 && (n & (n - 1)) == 0) {                                     /* :205/:210 hilbert(), :213 its real part */
        double *an = (double *)malloc(sizeof(double) * 2 * n);
        pss_o_hilbert(audio, n, an);                                                        /* pss_pocketfft.c: SciPy's transform bit for bit */
        for (int i = 0; i < n; i++) audio[i] = an[2 * i];
        free(an);
    }                                                                                       /* other lengths: the round trip is skipped (~1e-16) */
    double mx = 0.0;
    int has_nan = 0;
    for (int i = 0; i < n; i++) { double a = fabs(audio[i]); if (a != a) has_nan = 1; if (a > mx) mx = a; }
    if (has_nan) mx = NAN;
    for (int i = 0; i < n; i++) audio[i] = (audio[i] / mx) * 0.95;                  /* :216 */
}
void pss_o_demod_ssb_ex(const float *iq, int n, const double *taps, double *audio, int with_hilbert)
{
    double *r = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) r[i] = (double)iq[2 * i];
    demod_ssb_real(r, n, taps, audio, with_hilbert);
    free(r);
}
/* a complex128 buffer: lfilter(taps, 1.0, samples) is the same complex128 convolution the reference runs for complex64 input (which it widens
 * first), on samples that need no widening */
void pss_o_demod_ssb_c128(const double *iq, int n, const double *taps, double *audio, int with_hilbert)
{
    double *r = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < n; i++) r[i] = iq[2 * i];
    demod_ssb_real(r, n, taps, audio, with_hilbert);
    free(r);
}

/* np.int16(data * 32767): C truncation toward zero; NaN -> 0 (x86 cvttsd2si 0x8000...0 then low 16 bits) */
void pss_o_pcm16_stereo(const double *audio, int n, int16_t *pcm)
{
    for (int i = 0; i < n; i++) {
        double v = audio[i] * 32767.0;
        int16_t s = (v != v) ? 0 : (int16_t)(int32_t)v;
        pcm[2 * i] = s;
        pcm[2 * i + 1] = s;
    }
}

/* adjust_gain — pyspecsdr.py:898-919.  power arrives as np.float32; AGC_TARGET_POWER - power is
 * evaluated in float32 (python int is weak). */
int pss_o_agc_step(float power_db, int idx, int n_gains)
{
    float diff = -30.0f - power_db;
    if (fabsf(diff) < 2.0f) return idx;
    if (diff > 0) {
        idx += 1;
        if (idx > n_gains - 1) idx = n_gains - 1;
    } else {
        idx -= 1;
        if (idx < 0) idx = 0;
    }
    return idx;
}

/* np.interp(np.linspace(0, len-1, W), np.arange(len), row): linspace = start + i*step with the last
 * point forced to stop; interp = slope*(x - xp[j]) + fp[j] (numpy compiled_base.c arr_interp). */
static double interp_row(const double *row, int len, int W, int i)
{
    double stop = (double)(len - 1);
    double x;
    if (W == 1) x = 0.0;
    else {
        double step = stop / (double)(W - 1);
        x = (i == W - 1) ? stop : (double)i * step;
    }
    if (x >= stop) return row[len - 1];
    int j = (int)x;
    double slope = (row[j + 1] - row[j]) / ((double)(j + 1) - (double)j);
    return slope * (x - (double)j) + row[j];
}

static void ring_minmax(const double *rows, long count, double *mn, double *mx)
{
    double lo = INFINITY, hi = -INFINITY;
    for (long i = 0; i < count; i++) {
        double v = rows[i];
        if (isfinite(v)) { if (v < lo) lo = v; if (v > hi) hi = v; }
    }
    *mn = lo;
    *mx = hi;
}

/* draw_waterfall — pyspecsdr.py:1342-1406 (ring :1351-1353, min/max :1356-1358, interp :1378-1382,
 * quantise :1386-1398; no zero-range guard).  Screen row y shows the y-th newest ring row. */
void pss_o_waterfall_cells(const double *rows, int n_rows, int len, int disp_h, int disp_w,
                           int8_t *glyph, int8_t *colour)
{
    double mn, mx;
    ring_minmax(rows, (long)n_rows * len, &mn, &mx);
    memset(glyph, -1, (size_t)disp_h * disp_w);
    memset(colour, -1, (size_t)disp_h * disp_w);
    for (int y = 0; y < n_rows && y < disp_h; y++) {
        const double *row = rows + (long)(n_rows - 1 - y) * len;
        for (int x = 0; x < disp_w; x++) {
            double v = interp_row(row, len, disp_w, x);
            if (!isfinite(v)) continue;
            double nv = (v - mn) / (mx - mn);
            int ci = (int)(nv * 5);
            int g = nv > 0.75 ? 3 : nv > 0.5 ? 2 : nv > 0.25 ? 1 : 0;
            glyph[y * disp_w + x] = (int8_t)g;
            colour[y * disp_w + x] = (int8_t)ci;
        }
    }
}

/* draw_gradient_waterfall — pyspecsdr.py:1640-1716: the same ring, min/max and resampling as draw_waterfall, a zero-range
 * guard (:1657-1659), character index int(norm*8) into ' ._-=+*#@' (:1686-1691) and colour index int(norm*5) (:1694). */
void pss_o_gradient_cells(const double *rows, int n_rows, int len, int disp_h, int disp_w, int8_t *glyph, int8_t *colour)
{
    double mn, mx;
    ring_minmax(rows, (long)n_rows * len, &mn, &mx);
    double range = mx - mn;
    if (range == 0) range = 1;
    memset(glyph, -1, (size_t)disp_h * disp_w);
    memset(colour, -1, (size_t)disp_h * disp_w);
    for (int y = 0; y < n_rows && y < disp_h; y++) {
        const double *row = rows + (long)(n_rows - 1 - y) * len;
        for (int x = 0; x < disp_w; x++) {
            double v = interp_row(row, len, disp_w, x);
            if (!isfinite(v)) continue;
            double nv = (v - mn) / range;
            glyph[y * disp_w + x] = (int8_t)(int)(nv * 8);
            colour[y * disp_w + x] = (int8_t)(int)(nv * 5);
        }
    }
}

/* draw_surface_plot — pyspecsdr.py:1567-1616: row min/max with a zero-range guard (:1575-1580), resampling to
 * max_w - 8 columns (:1583-1587), magnitude int(value*20) (:1593), then '#' cells marching up-left at 45 degrees
 * (:1595-1596) on the WHOLE screen grid [max_h][max_w], colour pair 1 + y % 5 (:1601); later (x, y) overwrite. 0 = empty. */
void pss_o_surface_cells(const double *row, int len, int max_h, int max_w, int8_t *colour)
{
    const double COS45 = 0x1.6a09e667f3bcdp-1, SIN45 = 0x1.6a09e667f3bccp-1; /* np.cos / np.sin(np.radians(45)) */
    double mn, mx;
    ring_minmax(row, len, &mn, &mx);
    double range = mx - mn;
    if (range == 0) range = 1;
    const int disp_w = max_w - 8;
    memset(colour, 0, (size_t)max_h * max_w);
    double *nrm = (double *)malloc(sizeof(double) * len);
    for (int i = 0; i < len; i++) nrm[i] = (row[i] - mn) / range;
    for (int x = 0; x < disp_w; x++) {
        const double value = interp_row(nrm, len, disp_w, x);
        if (!isfinite(value)) continue;
        const int mag = (int)(value * 20);
        for (int y = 0; y < mag; y++) {
            const int sx = (int)((double)x - (double)y * COS45) + 8;
            const int sy = (int)((double)(max_h - 2) - (double)y * SIN45);
            if (sx >= 0 && sx < max_w && sy >= 2 && sy < max_h - 1) colour[sy * max_w + sx] = (int8_t)(1 + y % 5);
        }
    }
    free(nrm);
}

/* draw_vector_display — pyspecsdr.py:1718-1752: x = int(center_x + i*scale), y = int(center_y - q*scale) in float32 (the
 * samples are np.float32 scalars, the Python ints are weak), '.' where the dot lands on the screen.  grid[max_h][max_w]. */
void pss_o_vector_cells(const float *iq, int n, int max_h, int max_w, int8_t *grid)
{
    const int cx = max_w / 2, cy = max_h / 2, scale = (max_w < max_h ? max_w : max_h) / 4;
    memset(grid, 0, (size_t)max_h * max_w);
    for (int k = 0; k < n; k++) {
        const float fx = (float)cx + iq[2 * k] * (float)scale, fy = (float)cy - iq[2 * k + 1] * (float)scale;
        if (!isfinite(fx) || !isfinite(fy)) continue; /* the reference raises here; nothing is drawn */
        const int x = (int)fx, y = (int)fy;
        if (x >= 0 && x < max_w && y >= 0 && y < max_h) grid[y * max_w + x] = 1;
    }
}

/* draw_persistence — pyspecsdr.py:1512-1564 (ring :1521-1523, range guard :1528-1530,
 * alpha/colour :1544-1545, y-cell :1555-1556).  Later traces overwrite earlier ones. */
void pss_o_persistence_cells(const double *rows, int n_rows, int len, int disp_h, int disp_w,
                             int8_t *colour)
{
    double mn, mx;
    ring_minmax(rows, (long)n_rows * len, &mn, &mx);
    double range = mx - mn;
    if (range == 0) range = 1;
    memset(colour, 0, (size_t)disp_h * disp_w);
    for (int i = 0; i < n_rows; i++) {
        double alpha = pow(0.7, (double)(10 - i));
        int cp = (int)(1 + (5 * (1 - alpha)));
        const double *row = rows + (long)i * len;
        for (int x = 0; x < disp_w; x++) {
            double v = interp_row(row, len, disp_w, x);
            if (!isfinite(v)) continue;
            double nv = (v - mn) / range;
            int y = (int)((1 - nv) * (disp_h - 1));
            if (y >= 0 && y < disp_h) colour[y * disp_w + x] = (int8_t)cp;
        }
    }
}

/* draw_spectrogram — pyspecsdr.py:398-498: noise floor = 20th percentile (np.percentile, method 'linear': virtual index
 * (n-1)*0.2, numpy _lerp), display range :424-427, clip + x**0.7 :442-445, np.interp to the display width :448-452, bar
 * height int(value*H) :457-458, glyph / colour by strength and position in the bar :471-490.  glyph: 0 '.', 1 '-', 2 '=',
 * 3 '#', 4 ' '; colour = curses pair number (1 = cleared cell); -1 where nothing was drawn (non-finite column). */
void pss_o_spectrogram_cells(const double *row, int len, int disp_h, int disp_w, int8_t *glyph, int8_t *colour,
                             double *disp_min, double *disp_max)
{
    double *fin = (double *)malloc(sizeof(double) * len), *pw = (double *)malloc(sizeof(double) * len);
    int nf = 0;
    for (int i = 0; i < len; i++)
        if (isfinite(row[i])) fin[nf++] = row[i];
    memset(glyph, -1, (size_t)disp_h * disp_w);
    memset(colour, -1, (size_t)disp_h * disp_w);
    if (nf == 0) { free(fin); free(pw); return; }
    qsort(fin, nf, sizeof(double), cmp_double);
    const double max_db = fin[nf - 1];
    const double vi = (double)(nf - 1) * 0.2;
    long lo = (long)floor(vi), hi = lo + 1;
    if (vi >= (double)(nf - 1)) { lo = nf - 1; hi = nf - 1; }
    if (hi > nf - 1) hi = nf - 1;
    const double g = vi - floor(vi), a = fin[lo], b = fin[hi], dba = b - a;
    double noise = a + dba * g;
    if (g >= 0.5) noise = b - dba * (1 - g);
    const double range = max_db - noise;
    const double dmin = noise - (range * 0.1), dmax = max_db + (range * 0.05);
    if (disp_min) *disp_min = dmin;
    if (disp_max) *disp_max = dmax;
    for (int i = 0; i < len; i++) {
        double v = (row[i] - dmin) / (dmax - dmin);
        v = v < 0 ? 0 : (v > 1 ? 1 : v);                       /* np.clip(., 0, 1) (NaN stays NaN) */
        pw[i] = pow(v, 0.7);
    }
    for (int x = 0; x < disp_w; x++) {
        const double value = interp_row(pw, len, disp_w, x);
        if (!isfinite(value)) continue;
        int height = (int)(value * disp_h);
        if (height > disp_h) height = disp_h;
        for (int y = 0; y < disp_h; y++) { glyph[y * disp_w + x] = 4; colour[y * disp_w + x] = 1; }
        for (int y = disp_h - height; y < disp_h; y++) {
            const double rel = height > 0 ? (double)(y - (disp_h - height)) / (double)height : 0.0;
            int gch, col;
            if (value > 0.8) { gch = rel > 0.5 ? 3 : 2; col = 14; }
            else if (value > 0.4) { gch = rel > 0.5 ? 2 : 1; col = 13; }
            else if (value > 0.2) { gch = rel > 0.5 ? 1 : 0; col = 12; }
            else if (rel > 0.7) { gch = 0; col = 11; }
            else { gch = 4; col = 10; }
            glyph[y * disp_w + x] = (int8_t)gch;
            colour[y * disp_w + x] = (int8_t)col;
        }
    }
    free(fin); free(pw);
}

/* bench.py cpu_baseline leg: the BASELINE.json headline path per frame, exactly what the reference's
 * loop does per read buffer (compute_fft + demodulate_signal(NFM) + int16), filters designed once. */
void pss_o_batch_spectrum_nfm(const float *iq, long n_frames, int n, double fs, int q, const double *taps,
                              const double *sos, const double *zi, float *db_out, int16_t *pcm_out,
                              int n_threads)
{
    int n_out = (int)(((long)n - 1 + q - 1) / q);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
    for (long f = 0; f < n_frames; f++) {
        double *db = (double *)malloc(sizeof(double) * n);
        double *au = (double *)malloc(sizeof(double) * (n_out + 1));
        pss_o_compute_fft(iq + 2 * f * n, n, db);
        for (int k = 0; k < n; k++) db_out[f * n + k] = (float)db[k];
        pss_o_demod_nfm(iq + 2 * f * n, n, fs, q, taps, sos, zi, au, 0, 0);
        pss_o_pcm16_stereo(au, n_out, pcm_out + 2 * f * n_out);
        free(db);
        free(au);
    }
}

/* The full BASELINE cfg-2 step per frame: compute_fft (float32 dB out), the caller's smoothing + median clamp
 * (pyspecsdr.py:2278-2283; float32 rows out, finite extremes per row), NFM -> int16 stereo.  post_out / lo_out / hi_out may be
 * NULL (then this is pss_o_batch_spectrum_nfm). */
void pss_o_batch_spectrum_post_nfm(const float *iq, long n_frames, int n, double fs, int q, const double *taps,
                                   const double *sos, const double *zi, float *db_out, float *post_out, float *lo_out,
                                   float *hi_out, int16_t *pcm_out, int n_threads)
{
    int n_out = (int)(((long)n - 1 + q - 1) / q);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
    /* scratch once per thread, contiguous blocks of frames per thread (schedule(static)) */
    double *db = (double *)malloc(sizeof(double) * n);
    double *po = (double *)malloc(sizeof(double) * n);
    double *au = (double *)malloc(sizeof(double) * (n_out + 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (long f = 0; f < n_frames; f++) {
        pss_o_compute_fft(iq + 2 * f * n, n, db);
        for (int k = 0; k < n; k++) db_out[f * n + k] = (float)db[k];
        if (post_out) {
            pss_o_postprocess(db, n, po);
            float lo = INFINITY, hi = -INFINITY;
            for (int k = 0; k < n - 4; k++) {
                const float v = (float)po[k];
                post_out[f * (n - 4) + k] = v;
                if (isfinite(v)) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
            }
            if (lo_out) lo_out[f] = lo;
            if (hi_out) hi_out[f] = hi;
        }
        pss_o_demod_nfm(iq + 2 * f * n, n, fs, q, taps, sos, zi, au, 0, 0);
        pss_o_pcm16_stereo(au, n_out, pcm_out + 2 * f * n_out);
    }
    free(db);
    free(po);
    free(au);
    }
}

/* Batched waterfall accumulator: for every frame the newest display line (y = 0) of draw_waterfall (pyspecsdr.py:1342-1406)
 * with the history of the last `window` rows — min / max over the finite values of rows i-window+1 .. i (:1356-1358),
 * np.interp to disp_w (:1379-1383), glyph / colour (:1386-1398).  rows float32 [n_frames][len]; glyph / colour int8
 * [n_frames][disp_w], -1 where the value is not finite. */
void pss_o_waterfall_rows(const float *rows, long n_frames, int len, int window, int disp_w, int8_t *glyph, int8_t *colour,
                          int n_threads)
{
    double *rlo = (double *)malloc(sizeof(double) * (n_frames > 0 ? n_frames : 1));
    double *rhi = (double *)malloc(sizeof(double) * (n_frames > 0 ? n_frames : 1));
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
    for (long f = 0; f < n_frames; f++) {
        double lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < len; k++) {
            const double v = (double)rows[f * len + k];
            if (isfinite(v)) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        }
        rlo[f] = lo;
        rhi[f] = hi;
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
    for (long f = 0; f < n_frames; f++) {
        double lo = INFINITY, hi = -INFINITY;
        for (long p = f - (window - 1) < 0 ? 0 : f - (window - 1); p <= f; p++) {
            lo = rlo[p] < lo ? rlo[p] : lo;
            hi = rhi[p] > hi ? rhi[p] : hi;
        }
        const float *row = rows + f * len;
        const double stop = (double)(len - 1);
        for (int x = 0; x < disp_w; x++) {
            double xp, v;
            if (disp_w == 1) xp = 0.0;
            else {
                const double step = stop / (double)(disp_w - 1);
                xp = (x == disp_w - 1) ? stop : (double)x * step;
            }
            if (xp >= stop) v = (double)row[len - 1];
            else {
                const int j = (int)xp;
                const double slope = ((double)row[j + 1] - (double)row[j]) / ((double)(j + 1) - (double)j);
                v = slope * (xp - (double)j) + (double)row[j];
            }
            int8_t g = -1, ci = -1;
            if (isfinite(v)) {
                const double nv = (v - lo) / (hi - lo);
                ci = (int8_t)(int)(nv * 5);
                g = nv > 0.75 ? 3 : nv > 0.5 ? 2 : nv > 0.25 ? 1 : 0;
            }
            glyph[f * disp_w + x] = g;
            colour[f * disp_w + x] = ci;
        }
    }
    free(rlo);
    free(rhi);
}

/* Batched persistence accumulator: for every frame the row index of the NEWEST trace's '*' in every display column as draw_persistence
 * (pyspecsdr.py:1512-1564) places it with the history of the last `window` rows — min / max over the finite values of rows
 * i-window+1 .. i (:1525-1527), range 0 -> 1 (:1528-1530), np.interp to disp_w (:1547-1551), y = int((1 - norm) * (disp_h - 1))
 * (:1555-1556), drawn only inside the grid (:1557).  rows float32 [n_frames][len]; y int8 [n_frames][disp_w], -1 where nothing is drawn. */
void pss_o_persistence_rows(const float *rows, long n_frames, int len, int window, int disp_h, int disp_w, int8_t *ycell, int n_threads)
{
    double *rlo = (double *)malloc(sizeof(double) * (n_frames > 0 ? n_frames : 1));
    double *rhi = (double *)malloc(sizeof(double) * (n_frames > 0 ? n_frames : 1));
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
    for (long f = 0; f < n_frames; f++) {
        double lo = INFINITY, hi = -INFINITY;
        for (int k = 0; k < len; k++) {
            const double v = (double)rows[f * len + k];
            if (isfinite(v)) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        }
        rlo[f] = lo;
        rhi[f] = hi;
    }
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
    for (long f = 0; f < n_frames; f++) {
        double lo = INFINITY, hi = -INFINITY;
        for (long p = f - (window - 1) < 0 ? 0 : f - (window - 1); p <= f; p++) {
            lo = rlo[p] < lo ? rlo[p] : lo;
            hi = rhi[p] > hi ? rhi[p] : hi;
        }
        double range = hi - lo;
        if (range == 0) range = 1;
        const float *row = rows + f * len;
        const double stop = (double)(len - 1);
        for (int x = 0; x < disp_w; x++) {
            double xp, v;
            if (disp_w == 1) xp = 0.0;
            else {
                const double step = stop / (double)(disp_w - 1);
                xp = (x == disp_w - 1) ? stop : (double)x * step;
            }
            if (xp >= stop) v = (double)row[len - 1];
            else {
                const int j = (int)xp;
                const double slope = ((double)row[j + 1] - (double)row[j]) / ((double)(j + 1) - (double)j);
                v = slope * (xp - (double)j) + (double)row[j];
            }
            int8_t y8 = -1;
            if (isfinite(v)) {
                const int y = (int)((1 - (v - lo) / range) * (disp_h - 1));
                if (y >= 0 && y < disp_h) y8 = (int8_t)y;
            }
            ycell[f * disp_w + x] = y8;
        }
    }
    free(rlo);
    free(rhi);
}


/* The reference's own step per read buffer with its own row type — float64 from IQ to cells: compute_fft returns float64
 * (signal_processing.py:243-264), the caller smooths and clamps those rows (pyspecsdr.py:2278-2283), draw_waterfall normalises and
 * quantises them (:1342-1406).  db_out / post_out / pcm_out may be NULL; lo_out / hi_out: finite extremes of every post-processed row. */
void pss_o_batch_headline_f64(const float *iq, long n_frames, int n, double fs, int q, const double *taps, const double *sos,
                              const double *zi, double *db_out, double *post_out, double *lo_out, double *hi_out, int16_t *pcm_out,
                              int n_threads)
{
    int n_out = (int)(((long)n - 1 + q - 1) / q);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads)
#endif
    {
        double *db = (double *)malloc(sizeof(double) * n);
        double *po = (double *)malloc(sizeof(double) * n);
        double *au = (double *)malloc(sizeof(double) * (n_out + 1));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long f = 0; f < n_frames; f++) {
            pss_o_compute_fft(iq + 2 * f * n, n, db);
            if (db_out) memcpy(db_out + f * n, db, sizeof(double) * n);
            pss_o_postprocess(db, n, po);
            if (post_out) memcpy(post_out + f * (n - 4), po, sizeof(double) * (n - 4));
            double lo = INFINITY, hi = -INFINITY;
            for (int k = 0; k < n - 4; k++) {
                const double v = po[k];
                if (isfinite(v)) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
            }
            lo_out[f] = lo;
            hi_out[f] = hi;
            if (pcm_out) {
                pss_o_demod_nfm(iq + 2 * f * n, n, fs, q, taps, sos, zi, au, 0, 0);
                pss_o_pcm16_stereo(au, n_out, pcm_out + 2 * f * n_out);
            }
        }
        free(db); free(po); free(au);
    }
}

/* pss_o_waterfall_rows on float64 rows, with the rows' extremes supplied (row_lo / row_hi [n_frames], e.g. from pss_o_batch_headline_f64). */
void pss_o_waterfall_rows_f64(const double *rows, const double *row_lo, const double *row_hi, long n_frames, int len, int window, int disp_w,
                              int8_t *glyph, int8_t *colour, int n_threads)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
    for (long f = 0; f < n_frames; f++) {
        double lo = INFINITY, hi = -INFINITY;
        for (long p = f - (window - 1) < 0 ? 0 : f - (window - 1); p <= f; p++) {
            lo = row_lo[p] < lo ? row_lo[p] : lo;
            hi = row_hi[p] > hi ? row_hi[p] : hi;
        }
        const double *row = rows + f * len;
        for (int x = 0; x < disp_w; x++) {
            const double v = interp_row(row, len, disp_w, x);
            int8_t g = -1, ci = -1;
            if (isfinite(v)) {
                const double nv = (v - lo) / (hi - lo);
                ci = (int8_t)(int)(nv * 5);
                g = nv > 0.75 ? 3 : nv > 0.5 ? 2 : nv > 0.25 ? 1 : 0;
            }
            glyph[f * disp_w + x] = g;
            colour[f * disp_w + x] = ci;
        }
    }
}

/* pss_o_persistence_rows on float64 rows with the rows' extremes supplied (draw_persistence :1512-1564, newest trace per frame). */
void pss_o_persistence_rows_f64(const double *rows, const double *row_lo, const double *row_hi, long n_frames, int len, int window, int disp_h,
                                int disp_w, int8_t *ycell, int n_threads)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
    for (long f = 0; f < n_frames; f++) {
        double lo = INFINITY, hi = -INFINITY;
        for (long p = f - (window - 1) < 0 ? 0 : f - (window - 1); p <= f; p++) {
            lo = row_lo[p] < lo ? row_lo[p] : lo;
            hi = row_hi[p] > hi ? row_hi[p] : hi;
        }
        double range = hi - lo;
        if (range == 0) range = 1;
        const double *row = rows + f * len;
        for (int x = 0; x < disp_w; x++) {
            const double v = interp_row(row, len, disp_w, x);
            int8_t y8 = -1;
            if (isfinite(v)) {
                const int y = (int)((1 - (v - lo) / range) * (disp_h - 1));
                if (y >= 0 && y < disp_h) y8 = (int8_t)y;
            }
            ycell[f * disp_w + x] = y8;
        }
    }
}
