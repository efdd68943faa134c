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
#include <vector>

typedef std::complex<double> cplx;

extern "C" int pss_design_firwin(int numtaps, double cutoff, double *taps)
{
    if (numtaps < 1 || !taps) return PSS_E_ARG;
    if (!(cutoff > 0.0 && cutoff < 1.0)) return PSS_E_CUTOFF;  // scipy: "Invalid cutoff frequency"
    const double alpha = 0.5 * (numtaps - 1);
    double s = 0.0;
    for (int i = 0; i < numtaps; i++) {
        double m = (double)i - alpha;
        double x = cutoff * m;
        double y = M_PI * (x == 0.0 ? 1.0e-20 : x);  // np.sinc
        double h = cutoff * (std::sin(y) / y);
        // general_cosine(M, [0.54, 0.46], sym=True): fac = linspace(-pi, pi, M)
        double fac = (numtaps == 1) ? -M_PI : -M_PI + (double)i * (2.0 * M_PI / (double)(numtaps - 1));
        if (i == numtaps - 1 && numtaps > 1) fac = M_PI;
        double w = 0.54 + 0.46 * std::cos(fac);
        taps[i] = h * w;
        s += taps[i];  // scale_frequency = 0 -> cos(0) = 1
    }
    for (int i = 0; i < numtaps; i++) taps[i] /= s;
    return PSS_OK;
}

extern "C" int pss_design_cheby1_sos(int order, double rp_db, double wn, double *sos)
{
    if (order < 2 || (order & 1) || !sos || !(wn > 0.0 && wn < 1.0) || !(rp_db > 0)) return PSS_E_ARG;
    const int N = order;
    // cheb1ap
    double eps = std::sqrt(std::pow(10.0, 0.1 * rp_db) - 1.0);
    double mu = 1.0 / N * std::asinh(1.0 / eps);
    std::vector<cplx> p(N);
    cplx kprod(1.0, 0.0);
    for (int i = 0; i < N; i++) {
        double m = (double)(-N + 1 + 2 * i);
        double theta = M_PI * m / (2.0 * N);
        p[i] = -std::sinh(cplx(mu, theta));
        kprod *= -p[i];
    }
    double k = kprod.real() / std::sqrt(1.0 + eps * eps);  // N even
    // pre-warp, lp2lp_zpk
    const double fs = 2.0;
    double warped = 2.0 * fs * std::tan(M_PI * wn / fs);
    for (int i = 0; i < N; i++) p[i] *= warped;
    k *= std::pow(warped, (double)N);
    // bilinear_zpk (no finite zeros: all N digital zeros land on -1)
    const double fs2 = 2.0 * fs;
    cplx den(1.0, 0.0);
    for (int i = 0; i < N; i++) {
        den *= (fs2 - p[i]);
        p[i] = (fs2 + p[i]) / (fs2 - p[i]);
    }
    k *= (cplx(1.0, 0.0) / den).real();
    // zpk2sos, pairing='nearest': keep one pole per conjugate pair (imag > 0), the pole closest to the unit
    // circle goes to the LAST section; every section gets the zero pair (-1, -1); gain on section 0.
    std::vector<cplx> pc;
    for (int i = 0; i < N; i++)
        if (p[i].imag() > 0) pc.push_back(p[i]);
    if ((int)pc.size() != N / 2) return PSS_E_ARG;
    int nsec = N / 2;
    for (int si = nsec - 1; si >= 0; si--) {
        int worst = 0;
        double best = 1e300;
        for (size_t j = 0; j < pc.size(); j++) {
            double d = std::fabs(1.0 - std::abs(pc[j]));
            if (d < best) { best = d; worst = (int)j; }
        }
        cplx p1 = pc[worst];
        pc.erase(pc.begin() + worst);
        double *row = sos + 6 * si;
        row[0] = 1.0; row[1] = 2.0; row[2] = 1.0;
        row[3] = 1.0;
        row[4] = -(p1.real() + p1.real());
        row[5] = p1.real() * p1.real() + p1.imag() * p1.imag();
    }
    sos[0] *= k; sos[1] *= k; sos[2] *= k;
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
        double y1 = B1 - l * B0;
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
