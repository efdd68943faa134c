// pss_design.cpp — host-side filter design for the demodulators (plain C++, no GPU code).
//
// The reference re-designs its filters with SciPy on every call (≈76 % of its NFM time):
//   firwin(65, 15000/(fs/2))                 signal_processing.py:107   (NFM low-pass)
//   decimate(x, q) -> cheby1(8, 0.05, 0.8/q) signal_processing.py:112   (scipy _signaltools.py:4831)
//   sosfilt_zi(sos)                          scipy _signaltools.py:4718 (inside sosfiltfilt)
//   firwin(65, 3000/fs)                      signal_processing.py:203/208 (SSB)
//   butter(5, [300,3000]/11025, 'band')      signal_processing.py:188-191 -> :39-41 (AM, fixed)
// Here each is designed once per sample rate and cached in the context.  The algorithms follow SciPy's
// (firwin: windowed sinc scaled to unit DC gain; cheby1: cheb1ap -> lp2lp_zpk -> bilinear_zpk -> zpk2sos
// with 'nearest' pairing; sosfilt_zi: per-section lfilter_zi scaled by the running DC gain).  SciPy evaluates
// sin/cos/sinh through NumPy's SIMD loops, so coefficients agree to a few ulp rather than bit-for-bit;
// tests/test_design.py bounds the difference and pss_set_nfm_filters() lets a caller inject SciPy's tables.
#include "../../include/pss.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <vector>

typedef std::complex<double> cplx;

// NumPy's arithmetic, restated where it is not the obvious one (this unit is compiled with -ffp-contract=off):
//   np.add.reduce on a contiguous float64 array: pairwise summation — eight running sums over blocks of eight, combined as
//   ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)), the remainder added one by one; halves above 128 elements;
//   complex128 multiply: (ar br - ai bi, ar bi + ai br), no fused multiply-add; complex128 divide: Smith's algorithm
//   (ratio of the divisor's smaller to its larger component, one reciprocal).
namespace {
double np_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double r = 0.0;
        for (long i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        long i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}
inline std::complex<double> np_cmul(std::complex<double> a, std::complex<double> b)
{
    return std::complex<double>(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real());
}
inline std::complex<double> np_cdiv(std::complex<double> a, std::complex<double> b)
{
    const double ar = a.real(), ai = a.imag(), br = b.real(), bi = b.imag();
    if (std::fabs(br) >= std::fabs(bi)) {
        if (br == 0.0 && bi == 0.0) return std::complex<double>(ar / std::fabs(br), ai / std::fabs(bi));
        const double rat = bi / br, scl = 1.0 / (br + bi * rat);
        return std::complex<double>((ar + ai * rat) * scl, (ai - ar * rat) * scl);
    }
    const double rat = br / bi, scl = 1.0 / (bi + br * rat);
    return std::complex<double>((ar * rat + ai) * scl, (ai * rat - ar) * scl);
}
// glibc's complex functions (what NumPy calls: npy_csqrt / npy_cexp / npy_csinh forward to them).  Not std::sqrt / exp / sinh on
// std::complex: in a HIP translation unit clang's <complex> wrapper turns libstdc++'s C99 forwarding off and the generic formulas
// land an ulp away (found on scipy.signal.butter's band-pass poles: 584 of 806 designs differed).
inline std::complex<double> libm_csqrt(std::complex<double> v)
{
    const double _Complex r = __builtin_csqrt(__builtin_complex(v.real(), v.imag()));
    return std::complex<double>(__real__ r, __imag__ r);
}
inline std::complex<double> libm_cexp(std::complex<double> v)
{
    const double _Complex r = __builtin_cexp(__builtin_complex(v.real(), v.imag()));
    return std::complex<double>(__real__ r, __imag__ r);
}
inline std::complex<double> libm_csinh(std::complex<double> v)
{
    const double _Complex r = __builtin_csinh(__builtin_complex(v.real(), v.imag()));
    return std::complex<double>(__real__ r, __imag__ r);
}

// ---- NumPy's float64 tan and exp -------------------------------------------------------------------------------------------
// Under its AVX512_SKX dispatch NumPy evaluates np.tan and np.exp on float64 through Intel's SVML (shipped in NumPy's own sources,
// numpy/_core/src/umath/svml: __svml_tan8_ha and __svml_exp8_ha — the "high accuracy" entry points), NOT through libm: 0.5 % of
// tan's and 5 % of exp's results differ from glibc's by an ulp, which is what separated this library's Butterworth tables and
// de-emphasis coefficient from SciPy's at some sample rates.  The two routines' main paths restated operation by operation (every
// FMA, the 16-entry tables of tan(j pi / 16) and 2^(j / 16) as high + low parts, VRCP14PD — the same 64-entry table as VRCP14PS,
// indexed by the top 16 mantissa bits, checked against the instruction on 2x10^7 operands — with its refinement, exp's first FMA in
// round-toward-zero); compared with np.tan / np.exp on 2.2x10^7 arguments (uniform on (-1000, 1000), (0, pi/2), small arguments,
// every de-emphasis argument of 2x10^6 sample rates): no differing bit (tests/test_abi_and_design.py keeps a 10^6-argument version).
// Outside the main paths' domains (|x| >= 65536 for tan, |x| >= 708 for exp, NaN) libm answers.
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



inline double u2d(uint64_t u) { double d; std::memcpy(&d, &u, 8); return d; }
inline uint64_t d2u(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }
double rcp14pd(double x)
{
    const uint64_t u = d2u(x), sign = u & 0x8000000000000000ull, e = (u >> 52) & 0x7ff, m = u & 0xfffffffffffffull;
    if (m == 0) return u2d(sign | ((2046ull - e) << 52));
    const uint32_t idx = (uint32_t)(m >> 46), low = (uint32_t)(m >> 36) & 1023u;
    const uint32_t v = (RCP14_A[idx] - (uint32_t)RCP14_B[idx] * low) >> 9;
    return u2d(sign | ((2045ull - e) << 52) | ((uint64_t)(v & 0xffffu) << 36));
}
const uint64_t EXP_TH[16] = {0x3ff0000000000000ull, 0x3ff0b5586cf9890full, 0x3ff172b83c7d517bull, 0x3ff2387a6e756238ull, 0x3ff306fe0a31b715ull, 0x3ff3dea64c123422ull,
                             0x3ff4bfdad5362a27ull, 0x3ff5ab07dd485429ull, 0x3ff6a09e667f3bcdull, 0x3ff7a11473eb0187ull, 0x3ff8ace5422aa0dbull, 0x3ff9c49182a3f090ull,
                             0x3ffae89f995ad3adull, 0x3ffc199bdd85529cull, 0x3ffd5818dcfba487ull, 0x3ffea4afa2a490daull};
const uint64_t EXP_TL[16] = {0x0ull, 0x3c979aa65d837b6dull, 0xbc801b15eaa59348ull, 0x3c968efde3a8a894ull, 0x3c834d754db0abb6ull, 0x3c859f48a72a4c6dull,
                             0x3c7690cebb7aafb0ull, 0x3c9063e1e21c5409ull, 0xbc93b3efbf5e2228ull, 0xbc7b32dcb94da51dull, 0x3c8db72fc1f0eab4ull, 0x3c71affc2b91ce27ull,
                             0x3c8c1a7792cb3387ull, 0x3c736eae30af0cb3ull, 0x3c74a385a63d07a7ull, 0xbc8ff7128fd391f0ull};
const uint64_t TAN_TH[16] = {0x8000000000000000ull, 0x3fc975f5e0553158ull, 0x3fda827999fcef32ull, 0x3fe561b82ab7f990ull, 0x3ff0000000000000ull, 0x3ff7f218e25a7461ull,
                             0x4003504f333f9de6ull, 0x40141bfee2424771ull, 0xffefffffffffffffull, 0xc0141bfee2424771ull, 0xc003504f333f9de6ull, 0xbff7f218e25a7461ull,
                             0xbff0000000000000ull, 0xbfe561b82ab7f990ull, 0xbfda827999fcef32ull, 0xbfc975f5e0553158ull};
const uint64_t TAN_TL[16] = {0x8000000000000000ull, 0x3c2ef5d367441946ull, 0x3c708b2fb1366ea9ull, 0x3c87a8c52172b675ull, 0x0ull, 0x3c9419fa6954928full,
                             0x3ca21165f626cdd5ull, 0x3c810706fed37f0eull, 0xfca0000000000000ull, 0xbc810706fed37f0eull, 0xbca21165f626cdd5ull, 0xbc9419fa6954928full,
                             0x0ull, 0xbc87a8c52172b675ull, 0xbc708b2fb1366ea9ull, 0xbc2ef5d367441946ull};
}  // namespace

// np.exp(x) for a float64 x (NumPy's AVX512_SKX dispatch)
double pss_np_exp(double x)
{
    if (!(std::fabs(x) < 708.0)) return std::exp(x);
    const double L2E = u2d(0x3ff71547652b82feull), SH = u2d(0x42f8000000003ff0ull), L2H = u2d(0x3fe62e42fefa39efull), L2L = u2d(0x3c7abc9e3b39803full);
    const double c0 = u2d(0x3f57411836940c04ull), c1 = u2d(0x3f81101cbbc265c0ull), c2 = u2d(0x3fa55557242d68feull), c3 = u2d(0x3fc5555553939732ull),
                 c4 = u2d(0x3fe000000000d008ull), c5 = u2d(0x3fefffffffffff70ull);
    double z = std::fma(x, L2E, SH);
    if (std::fma(x, L2E, SH - z) < 0.0) z = std::nextafter(z, 0.0);      // the routine's first FMA rounds toward zero (z > 0)
    const double N = z - SH;
    const int j = (int)(d2u(z) & 15);
    const double Th = u2d(EXP_TH[j]), Tl = u2d(EXP_TL[j]);
    double r = std::fma(-N, L2H, x);
    r = std::fma(-L2L, N, r);
    const double rm = u2d(d2u(r) & 0xbfffffffffffffffull);
    const double r2 = rm * rm;
    double A = std::fma(c0, rm, c1);
    const double B = std::fma(c2, rm, c3), C = std::fma(c4, rm, c5);
    A = std::fma(r2, A, B);
    A = std::fma(r2, A, C);
    double p = std::fma(A, rm, Tl);
    p = std::fma(Th, p, Th);
    return std::ldexp(p, (int)std::floor(N));
}

// np.tan(x) for a float64 x (NumPy's AVX512_SKX dispatch)
double pss_np_tan(double x)
{
    if (!(std::fabs(x) < 65536.0)) return std::tan(x);
    const double IP = u2d(0x40145f306dc9c883ull), SH = u2d(0x4338000000000000ull), P1 = u2d(0x3fc921fb54442d18ull), P2 = u2d(0x3c61a62633000000ull),
                 P3 = u2d(0x3a645c06e0e68948ull);
    const double c0 = u2d(0x3fd55555555555dcull), c1 = u2d(0x3fc11111110b0802ull), c2 = u2d(0x3faba1ba489d25caull), c3 = u2d(0x3f9664ab664efba9ull),
                 c4 = u2d(0x3f825cccc7c9fa5dull);
    const double z = std::fma(x, IP, SH), N = z - SH;
    const double r1 = std::fma(-N, P1, x), r2 = std::fma(-N, P2, r1), R = std::fma(-N, P3, r2);
    const double e1 = std::fma(-P2, N, r1 - r2), e2 = std::fma(P3, N, R - r2);
    const double Rl = e1 - e2;
    const int j = (int)(d2u(z) & 15);
    const double Th = u2d(TAN_TH[j]), Tl = u2d(TAN_TL[j]);
    const double R2 = R * R;
    double p = std::fma(c4, R2, c3);
    p = std::fma(R2, p, c2);
    p = std::fma(R2, p, c1);
    p = std::fma(R2, p, c0);
    const double t9 = -std::fma(R2, p * R, Rl);
    const double Ph = R - t9, Pl = (R - Ph) - t9;
    const double numh = Ph + Th;
    const double numl = ((Ph - (numh - Th)) + Tl) + Pl;
    const double denh = std::fma(-Ph, Th, 1.0);
    const double denl = std::fma(Ph, Tl, std::fma(Pl, Th, std::fma(Ph, Th, denh - 1.0)));
    double rc = rcp14pd(denh);
    const double e = std::fma(denl, rc, std::fma(-denh, rc, 1.0));
    rc = std::fma(e, rc, rc);
    const double q = numh * rc;
    const double t0 = std::fma(-q, denl, std::fma(q, denh, -numh)) - numl;
    return std::fma(-rc, t0, q);
}

extern "C" int pss_h_np_f64(int op, const double *x, long n, double *out)
{
    if ((op != 0 && op != 1) || n < 0 || (n > 0 && (!x || !out))) return PSS_E_ARG;
    for (long i = 0; i < n; i++) out[i] = op == 0 ? pss_np_tan(x[i]) : pss_np_exp(x[i]);
    return PSS_OK;
}

// scipy.signal.firwin(numtaps, cutoff) (window='hamming', pass_zero=True, scale=True): every operation in SciPy's order —
// np.sinc (pi * where(x == 0, 1e-20, x); sin(y) / y), general_cosine's fac = linspace(-pi, pi, M) and w = 0.54 cos(0 fac) +
// (1 - 0.54) cos(fac), the scale by np.sum(h * cos(0)) (pairwise).  NumPy's float64 sin / cos are libm's (checked on the build image:
// np.sin == math.sin on 200 000 arguments), so the taps equal SciPy's bit for bit (tests: 400 cutoffs).
extern "C" int pss_design_firwin(int numtaps, double cutoff, double *taps)
{
    if (numtaps < 1 || !taps) return PSS_E_ARG;
    if (!(cutoff > 0.0 && cutoff < 1.0)) return PSS_E_CUTOFF;  // scipy: "Invalid cutoff frequency"
    const double alpha = 0.5 * (numtaps - 1);
    const double step = numtaps > 1 ? (M_PI - (-M_PI)) / (double)(numtaps - 1) : 0.0;
    std::vector<double> hc(numtaps);
    for (int i = 0; i < numtaps; i++) {
        const double m = (double)i - alpha;
        const double x = cutoff * m;
        const double y = M_PI * (x == 0.0 ? 1.0e-20 : x);  // np.sinc
        const double h = cutoff * (std::sin(y) / y);
        double fac = (double)i * step + (-M_PI);             // np.linspace: arange * step + start, the last element set to stop
        if (i == numtaps - 1 && numtaps > 1) fac = M_PI;
        double w = 0.0;
        w += 0.54 * std::cos(0.0 * fac);
        w += (1.0 - 0.54) * std::cos(fac);
        taps[i] = h * w;
        hc[i] = taps[i] * std::cos(M_PI * m * 0.0);          // scale_frequency = 0
    }
    const double s = np_pairwise_sum(hc.data(), numtaps);
    for (int i = 0; i < numtaps; i++) taps[i] /= s;
    return PSS_OK;
}

// scipy.signal.cheby1(order, rp, wn, output='sos') (what scipy.signal.decimate designs: order 8, rp 0.05, wn 0.8 / q): cheb1ap ->
// pre-warp -> lp2lp_zpk -> bilinear_zpk -> zpk2sos(pairing='nearest'), every operation in SciPy's / NumPy's order (complex sinh =
// glibc's csinh, np.abs = hypot, Python's float ** int = pow, complex products and quotients as above).  NumPy evaluates tan and
// arcsinh through SVML, which is NOT libm in general (0.5 % / 13 % of random arguments differ) — on the arguments this chain can
// produce they agree: arcsinh(1 / eps) for (8, 0.05), and tan(pi 0.8 / q / 2) for every q = 2 .. 4095 (checked on the build image), so
// the sections equal SciPy's bit for bit for every decimation factor (tests: q = 2 .. 1199).
extern "C" int pss_design_cheby1_sos(int order, double rp_db, double wn, double *sos)
{
    if (order < 2 || (order & 1) || !sos || !(wn > 0.0 && wn < 1.0) || !(rp_db > 0)) return PSS_E_ARG;
    const int N = order;
    // cheb1ap
    const double eps = std::sqrt(std::pow(10.0, 0.1 * rp_db) - 1.0);
    const double mu = 1.0 / N * std::asinh(1.0 / eps);
    std::vector<cplx> p(N);
    cplx kprod;
    for (int i = 0; i < N; i++) {
        const double m = (double)(-N + 1 + 2 * i);
        const double theta = M_PI * m / (double)(2 * N);
        p[i] = -libm_csinh(cplx(mu, theta));
        kprod = i == 0 ? -p[0] : np_cmul(kprod, -p[i]);
    }
    double k = kprod.real() / std::sqrt(1.0 + eps * eps);  // N even
    // pre-warp, lp2lp_zpk
    const double fs = 2.0;
    const double warped = 2.0 * fs * pss_np_tan(M_PI * wn / fs);
    for (int i = 0; i < N; i++) p[i] = cplx(warped * p[i].real(), warped * p[i].imag());
    k = k * std::pow(warped, (double)N);
    // bilinear_zpk (no finite zeros: all N digital zeros land on -1)
    const double fs2 = 2.0 * fs;
    cplx den;
    for (int i = 0; i < N; i++) {
        const cplx d(fs2 - p[i].real(), -p[i].imag());
        den = i == 0 ? d : np_cmul(den, d);
        p[i] = np_cdiv(cplx(fs2 + p[i].real(), p[i].imag()), d);
    }
    k = k * np_cdiv(cplx(1.0, 0.0), den).real();
    // zpk2sos, pairing='nearest': _cplxreal keeps one pole per conjugate pair ((zp + conj(zn)) / 2 = zp), sorted by real part; the pole
    // closest to the unit circle goes to the LAST section; every section gets the zero pair (-1, -1): np.poly -> [1, 2, 1] and
    // [1, -p1 - conj(p1), p1 conj(p1)] by np.convolve's plain dot loop; the gain goes on section 0.
    std::vector<cplx> pc;
    for (int i = 0; i < N; i++)
        if (p[i].imag() > 0) pc.push_back(p[i]);
    if ((int)pc.size() != N / 2) return PSS_E_ARG;
    std::stable_sort(pc.begin(), pc.end(), [](const cplx &a, const cplx &b) { return a.real() < b.real(); });
    const int nsec = N / 2;
    for (int si = nsec - 1; si >= 0; si--) {
        int worst = 0;
        double best = 0.0;
        for (size_t j = 0; j < pc.size(); j++) {
            const double d = std::fabs(1.0 - std::hypot(pc[j].real(), pc[j].imag()));
            if (j == 0 || d < best) { best = d; worst = (int)j; }
        }
        const cplx p1 = pc[worst];
        pc.erase(pc.begin() + worst);
        double *row = sos + 6 * si;
        row[0] = 1.0; row[1] = 2.0; row[2] = 1.0;
        row[3] = 1.0;
        row[4] = (0.0 + (1.0 * (-p1.real()) - 0.0 * p1.imag())) + ((-p1.real()) * 1.0 - (-p1.imag()) * 0.0);
        row[5] = 0.0 + ((-p1.real()) * (-p1.real()) - (-p1.imag()) * p1.imag());
    }
    sos[0] *= k; sos[1] *= k; sos[2] *= k;
    return PSS_OK;
}

// scipy.signal.butter(N, Wn, btype='low'|'band', output='sos') (_filter_design.py: iirfilter -> buttap ->
// lp2lp_zpk | lp2bp_zpk -> bilinear_zpk -> zpk2sos(pairing='nearest')).  wn_low <= 0 selects the low-pass with cutoff
// wn_high (what the reference's bandpass_filter does, signal_processing.py:37-39).  sos: ceil(N/2) rows (low) or N rows
// (band); *nsec receives the row count.
namespace {
inline bool is_real(const cplx &v) { return v.imag() == 0.0; }
// zpk2sos's inner pairing loop on the "cplxreal" lists (one member per conjugate pair, imag > 0, then the reals)
int zpk2sos_nearest(std::vector<cplx> z, std::vector<cplx> p, double k, double *sos, int nsec)
{
    auto nearest_idx = [](const std::vector<cplx> &fro, cplx to, int which) {  // which: 0 real, 1 complex, 2 any
        int best = -1;
        double bd = 0;
        for (size_t j = 0; j < fro.size(); j++) {
            if (which == 0 && !is_real(fro[j])) continue;
            if (which == 1 && is_real(fro[j])) continue;
            const cplx df(fro[j].real() - to.real(), fro[j].imag() - to.imag());
            double d = std::hypot(df.real(), df.imag());   // np.abs(complex) = hypot
            if (best < 0 || d < bd) { best = (int)j; bd = d; }
        }
        return best;
    };
    auto count_real = [](const std::vector<cplx> &v) { int c = 0; for (auto &x : v) c += is_real(x); return c; };
    auto put = [&](int si, const cplx *zz, int nz, cplx p1, cplx p2) {
        double *row = sos + 6 * si;
        // zpk2tf -> np.poly: [1, -(r1 + r2), r1 r2]
        if (nz == 2) {
            row[0] = 1.0; row[1] = -(zz[0] + zz[1]).real(); row[2] = (zz[0] * zz[1]).real();
        } else { row[0] = 1.0; row[1] = -zz[0].real(); row[2] = 0.0; }
        row[3] = 1.0; row[4] = -(p1 + p2).real(); row[5] = (p1 * p2).real();
    };
    for (int si = nsec - 1; si >= 0; si--) {
        int pi = 0;
        double best = 1e300;
        for (size_t j = 0; j < p.size(); j++) {
            double d = std::fabs(1.0 - std::hypot(p[j].real(), p[j].imag()));
            if (d < best) { best = d; pi = (int)j; }
        }
        cplx p1 = p[pi];
        p.erase(p.begin() + pi);
        if (is_real(p1) && count_real(p) == 0) {  // first-order section
            int zi = nearest_idx(z, p1, 2);
            if (zi < 0) return PSS_E_ARG;
            cplx z1 = z[zi];
            z.erase(z.begin() + zi);
            cplx zz[2] = {z1, cplx(0, 0)};
            put(si, zz, 2, p1, cplx(0, 0));
        } else if (!is_real(p1) && count_real(z) == 1) {
            int zi = nearest_idx(z, p1, 1);
            if (zi < 0) return PSS_E_ARG;
            cplx z1 = z[zi];
            z.erase(z.begin() + zi);
            cplx zz[2] = {z1, std::conj(z1)};
            put(si, zz, 2, p1, std::conj(p1));
        } else {
            cplx p2;
            if (is_real(p1)) {
                int pj = -1;
                double bd = 0;
                for (size_t j = 0; j < p.size(); j++) {
                    if (!is_real(p[j])) continue;
                    double d = std::fabs(std::hypot(p[j].real(), p[j].imag()) - 1.0);
                    if (pj < 0 || d < bd) { pj = (int)j; bd = d; }
                }
                if (pj < 0) return PSS_E_ARG;
                p2 = p[pj];
                p.erase(p.begin() + pj);
            } else p2 = std::conj(p1);
            if (z.empty()) return PSS_E_ARG;
            int zi = nearest_idx(z, p1, 2);
            cplx z1 = z[zi];
            z.erase(z.begin() + zi);
            if (!is_real(z1)) {
                cplx zz[2] = {z1, std::conj(z1)};
                put(si, zz, 2, p1, p2);
            } else if (!z.empty()) {
                int zj = nearest_idx(z, p1, 0);
                if (zj < 0) return PSS_E_ARG;
                cplx z2 = z[zj];
                z.erase(z.begin() + zj);
                cplx zz[2] = {z1, z2};
                put(si, zz, 2, p1, p2);
            } else {
                cplx zz[1] = {z1};
                put(si, zz, 1, p1, p2);
            }
        }
    }
    sos[0] *= k; sos[1] *= k; sos[2] *= k;
    return PSS_OK;
}
}  // namespace

extern "C" int pss_design_butter_sos(int order, double wn_low, double wn_high, double *sos, int *nsec_out)
{
    const int N = order;
    if (N < 1 || N > 16 || !sos) return PSS_E_ARG;
    const bool band = wn_low > 0.0;
    if (!(wn_high > 0.0 && wn_high < 1.0) || (band && !(wn_low < wn_high))) return PSS_E_CUTOFF;  // scipy ValueError
    // buttap: p = -exp(1j * pi * m / (2 N)) — NumPy forms (1j pi) m as a complex product and divides by the complex 2 N (Smith's
    // algorithm: a multiplication by the rounded reciprocal), then glibc's cexp
    std::vector<cplx> p(N), z;
    for (int i = 0; i < N; i++) {
        const double m = (double)(-N + 1 + 2 * i);
        const cplx arg = np_cdiv(np_cmul(cplx(0.0, M_PI), cplx(m, 0.0)), cplx((double)(2 * N), 0.0));
        p[i] = -libm_cexp(arg);
    }
    double k = 1.0;
    const double fs = 2.0, fs2 = 2.0 * fs;
    int degree = N;
    if (!band) {
        const double warped = 2.0 * fs * pss_np_tan(M_PI * wn_high / fs);
        for (auto &v : p) v = np_cmul(cplx(warped, 0.0), v);      // lp2lp_zpk: wo * p
        k = k * std::pow(warped, (double)degree);
    } else {
        const double w0 = 2.0 * fs * pss_np_tan(M_PI * wn_low / fs), w1 = 2.0 * fs * pss_np_tan(M_PI * wn_high / fs);
        const double bw = w1 - w0, wo = std::sqrt(w0 * w1);
        // lp2bp_zpk: p_lp = p * bw / 2 (a complex product, then Smith's division by 2 + 0j: the signs of the zeros matter to csqrt's
        // branch); p_lp ** 2 is np.square, whose real part is ONE fused operation fma(re, re, -(im im)) (probed on 2000 values at
        // every vector length); wo ** 2 is Python's float power = pow(wo, 2); np.sqrt(complex) = glibc's csqrt
        std::vector<cplx> pb(2 * N);
        const double wo2 = std::pow(wo, 2.0);
        for (int i = 0; i < N; i++) {
            const cplx pl = np_cdiv(np_cmul(p[i], cplx(bw, 0.0)), cplx(2.0, 0.0));
            const cplx sq(std::fma(pl.real(), pl.real(), -(pl.imag() * pl.imag())) - wo2, (pl.real() * pl.imag() + pl.imag() * pl.real()) - 0.0);
            const cplx r = libm_csqrt(sq);
            pb[i] = cplx(pl.real() + r.real(), pl.imag() + r.imag());
            pb[N + i] = cplx(pl.real() - r.real(), pl.imag() - r.imag());
        }
        p = pb;
        z.assign(degree, cplx(0, 0));
        k = k * std::pow(bw, (double)degree);
    }
    // bilinear_zpk: (fs2 + x) / (fs2 - x) by Smith's division, the gain k real(prod(fs2 - z) / prod(fs2 - p)) with sequential plain products
    cplx num(1.0, 0.0), den(1.0, 0.0);
    bool first = true;
    for (auto &v : z) {
        const cplx d(fs2 - v.real(), -v.imag());
        num = first ? d : np_cmul(num, d);
        first = false;
        v = np_cdiv(cplx(fs2 + v.real(), v.imag()), d);
    }
    const bool have_z = !first;
    first = true;
    for (auto &v : p) {
        const cplx d(fs2 - v.real(), -v.imag());
        den = first ? d : np_cmul(den, d);
        first = false;
        v = np_cdiv(cplx(fs2 + v.real(), v.imag()), d);
    }
    const int deg2 = (int)p.size() - (int)z.size();
    for (int i = 0; i < deg2; i++) z.push_back(cplx(-1.0, 0.0));
    k = k * np_cdiv(have_z ? num : cplx(1.0, 0.0), den).real();
    // zpk2sos front matter: equalise lengths, make the count even, reduce to one member per conjugate pair + reals
    if (p.size() % 2 == 1) { p.push_back(cplx(0, 0)); z.push_back(cplx(0, 0)); }
    const int nsec = (int)p.size() / 2;
    auto cplxreal = [](const std::vector<cplx> &v) {
        std::vector<cplx> c, r;
        for (auto &x : v) {
            if (std::fabs(x.imag()) <= 100 * 2.220446049250313e-16 * std::hypot(x.real(), x.imag())) r.push_back(cplx(x.real(), 0.0));
            else if (x.imag() > 0) c.push_back(x);
        }
        std::sort(r.begin(), r.end(), [](const cplx &a, const cplx &b) { return a.real() < b.real(); });
        std::sort(c.begin(), c.end(), [](const cplx &a, const cplx &b) {
            return a.real() != b.real() ? a.real() < b.real() : std::fabs(a.imag()) < std::fabs(b.imag()); });
        c.insert(c.end(), r.begin(), r.end());
        return c;
    };
    int r = zpk2sos_nearest(cplxreal(z), cplxreal(p), k, sos, nsec);
    if (r) return r;
    if (nsec_out) *nsec_out = nsec;
    return PSS_OK;
}

extern "C" int pss_design_sosfilt_zi(const double *sos, int nsec, double *zi)
{
    if (!sos || !zi || nsec < 1) return PSS_E_ARG;
    double scale = 1.0;
    for (int s = 0; s < nsec; s++) {
        const double *c = sos + 6 * s;
        double b0 = c[0], b1 = c[1], b2 = c[2], a1 = c[4], a2 = c[5];
        // (I - A^T) zi = B, A = companion(a):  [[1 + a1, -1], [a2, 1]] zi = [b1 - a1 b0, b2 - a2 b0]
        double A00 = 1.0 + a1, A01 = -1.0, A10 = a2, A11 = 1.0;
        double B0 = b1 - a1 * b0, B1 = b2 - a2 * b0;
        if (std::fabs(A10) > std::fabs(A00)) {  // partial pivoting (LAPACK dgesv)
            std::swap(A00, A10); std::swap(A01, A11); std::swap(B0, B1);
        }
        double l = A10 * (1.0 / A00);  // LAPACK dgetf2 scales the column by the reciprocal pivot
        double U11 = A11 - l * A01;
        double y1 = std::fma(-l, B0, B1);  // dgetrs' forward substitution runs on OpenBLAS's fused kernels (found on 1198 decimator designs: the unfused form misses one)
        double x1 = y1 / U11;
        double x0 = (B0 - A01 * x1) / A00;
        zi[2 * s] = scale * x0;
        zi[2 * s + 1] = scale * x1;
        scale *= (b0 + b1 + b2) / (1.0 + a1 + a2);
    }
    return PSS_OK;
}

// butter(5, [300/11025, 3000/11025], btype='band', output='sos') as produced by SciPy 1.15.3 — the AM
// band-pass never changes (the reference designs it for fs = 22050 whatever the real rate is,
// signal_processing.py:188), so it ships as a table.
extern "C" int pss_am_bandpass_sos(double *sos)
{
    static const double T[5][6] = {
        {0x1.8a839d7e945aep-9, 0x1.8a839d7e945aep-8, 0x1.8a839d7e945aep-9, 0x1.0000000000000p+0, -0x1.e54fb79358c15p-1, 0x1.49d9364acce79p-2},
        {0x1.0000000000000p+0, 0x1.0000000000000p+1, 0x1.0000000000000p+0, 0x1.0000000000000p+0, -0x1.1ca90978f7b15p+0, 0x1.59f47a2b56d53p-1},
        {0x1.0000000000000p+0, 0x0.0p+0, -0x1.0000000000000p+0, 0x1.0000000000000p+0, -0x1.5e8583d4ea52ap+0, 0x1.b1cd32fd38885p-2},
        {0x1.0000000000000p+0, -0x1.0000000000000p+1, 0x1.0000000000000p+0, 0x1.0000000000000p+0, -0x1.dc1f5dd7d8e1bp+0, 0x1.bca3b46626609p-1},
        {0x1.0000000000000p+0, -0x1.0000000000000p+1, 0x1.0000000000000p+0, 0x1.0000000000000p+0, -0x1.f2eb692bfdfbep+0, 0x1.e99774ee38750p-1},
    };
    if (!sos) return PSS_E_ARG;
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 6; j++) sos[6 * i + j] = T[i][j];
    return PSS_OK;
}
