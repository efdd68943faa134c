// pss_device.h — device-side float32/float64 primitives shared by the HIP kernels (gfx950).
//
// These are the bit-level building blocks that make the int16 audio identical to the reference's
// NumPy/SciPy path (xqtr/PySpecSDR signal_processing.py).  Every multiply-add that the reference's
// native code fuses is written as an explicit fma(); the translation unit is compiled with
// -ffp-contract=off so nothing else is fused.
#pragma once
#include <hip/hip_runtime.h>
#pragma clang fp contract(off)
#include <stdint.h>

#include "pss_npf32.h"   // f2u / u2f, log10f_np, cabsf_np

namespace pss {

// VRCP14PS bit-exact model (x86 AVX-512 reciprocal approximation): 64-segment piecewise-linear in the
// top 16 mantissa bits.  Table derived and exhaustively verified by tools/derive_rcp14.c.
// {intercept, slope} per segment, one 8-byte load per lookup
static __constant__ uint2 RCP14_AB[64] = {
    {67107072u, 1009u}, {66074112u, 977u}, {65073664u, 949u}, {64102400u, 921u},
    {63159040u, 893u}, {62244608u, 869u}, {61354752u, 843u}, {60491264u, 821u},
    {59650560u, 797u}, {58833920u, 777u}, {58038272u, 755u}, {57264640u, 735u},
    {56511488u, 717u}, {55778048u, 699u}, {55062784u, 681u}, {54365184u, 663u},
    {53686016u, 647u}, {53022976u, 631u}, {52377088u, 617u}, {51745536u, 601u},
    {51129600u, 587u}, {50528000u, 573u}, {49940992u, 561u}, {49366272u, 547u},
    {48805376u, 535u}, {48257024u, 523u}, {47721728u, 513u}, {47196672u, 501u},
    {46683904u, 491u}, {46181632u, 479u}, {45690368u, 469u}, {45209344u, 459u},
    {44739072u, 451u}, {44277504u, 441u}, {43826176u, 433u}, {43382784u, 423u},
    {42949120u, 415u}, {42523904u, 407u}, {42106880u, 399u}, {41698048u, 391u},
    {41297920u, 385u}, {40903936u, 377u}, {40517888u, 369u}, {40139520u, 363u},
    {39768320u, 357u}, {39402752u, 349u}, {39044608u, 343u}, {38692864u, 337u},
    {38347520u, 331u}, {38008064u, 325u}, {37674496u, 319u}, {37347840u, 315u},
    {37025280u, 309u}, {36708608u, 303u}, {36398080u, 299u}, {36091648u, 293u},
    {35791360u, 289u}, {35495680u, 285u}, {35204352u, 279u}, {34919168u, 275u},
    {34638080u, 271u}, {34361088u, 267u}, {34088192u, 263u}, {33819392u, 259u}};

// `tab`: the segment table — RCP14_AB itself (constant memory: a vector load of ~300 clocks even when it hits), or a copy the kernel
// keeps in LDS (k_nfm_fwd, k_spectrum_r16<DISC>: the lookup sits in the middle of every sample's dependent chain)
__device__ __forceinline__ float rcp14f(float x, const uint2 *tab = RCP14_AB)
{
    uint32_t u = f2u(x), sign = u & 0x80000000u, e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    uint32_t idx = m >> 17, low = (m >> 7) & 1023u;
    const uint2 ab = tab[idx];
    uint32_t v = (ab.x - ab.y * low) >> 9;
    uint32_t r = sign | ((253u - e) << 23) | ((v & 0xffffu) << 7);
    uint32_t r0 = sign | ((254u - e) << 23);
    return u2f(m == 0 ? r0 : r);
}

// Operands outside [2^-125, 2^123) (zeros, NaN, inf, denormals, huge): kept out of line so the float64 libm
// fallback does not bloat the hot kernels' register allocation.
__device__ __noinline__ float atan2f_svml_rare(float y, float x)
{
    const float PIO2 = 0x1.921fb6p+0f, PI = 0x1.921fb6p+1f;
    uint32_t xb = f2u(x), yb = f2u(y);
    uint32_t axb = xb & 0x7fffffffu, ayb = yb & 0x7fffffffu;
    uint32_t sx = xb & 0x80000000u, sy = yb & 0x80000000u;
    float ax = u2f(axb), ay = u2f(ayb);
    if (x != x || y != y) return x + y;
    if (axb == 0 || ayb == 0) {  // the routine's vector fix-up for zero operands
        float v = (!(ay < ax) && !(axb == 0 && ayb == 0)) ? PIO2 : 0.0f;
        v = u2f(f2u(v) | sx);
        if (sx) v = v + PI;
        return u2f(f2u(v) | sy);
    }
    return (float)atan2((double)y, (double)x);  // SVML's scalar "rare" helper works in double (pinned on random bit patterns: atan2f_bits.npz)
}

// numpy.arctan2(float32) under NumPy's AVX512_SKX dispatch == Intel SVML __svml_atan2f16 (np.angle at
// signal_processing.py:94).  Main path for 2^-125 <= |x|,|y| < 2^123; zero operands follow the routine's
// vector fix-up; NaN/inf/denormal/huge operands use IEEE special values / a double-precision fallback.
// The routine's main path, branch-free; `inr` tells whether the operands were in its range (otherwise the value is meaningless and
// atan2f_svml_rare has the answer).  Split out so that a batch of independent samples can be evaluated as interleaved straight-line
// chains with ONE (wave-uniform, almost never taken) fix-up branch behind them.
__device__ __forceinline__ float atan2f_svml_main(float y, float x, bool &inr, const uint2 *tab = RCP14_AB)
{
    const float PIO2 = 0x1.921fb6p+0f, PI = 0x1.921fb6p+1f;
    uint32_t xb = f2u(x), yb = f2u(y);
    uint32_t axb = xb & 0x7fffffffu, ayb = yb & 0x7fffffffu;
    uint32_t sx = xb & 0x80000000u, sy = yb & 0x80000000u;
    float ax = u2f(axb), ay = u2f(ayb);
    // main path: 2^-125 <= min(|x|, |y|) and max(|x|, |y|) < 2^123, tested on the ordered pair the routine needs anyway (two
    // float compares; NaN operands fail them, zeros and denormals the first, inf and huge values the second)
    bool k1 = ay < ax;
    float a = k1 ? ay : -ax;
    float b = k1 ? ax : ay;
    inr = (fabsf(a) >= 0x1p-125f) && (b < 0x1p123f);
    float base = k1 ? 0.0f : PIO2;
    float r0 = rcp14f(b, tab);
    float e = __fmaf_rn(-b, r0, 1.0f);
    float r1 = __fmaf_rn(r0, e, r0);
    float q0 = __fmul_rn(a, r1);
    float rem = __fmaf_rn(-b, q0, a);
    float q = __fmaf_rn(rem, r1, q0);
    float s = __fmul_rn(q, q);
    float s2 = __fmul_rn(s, s);
    float pa = __fmaf_rn(s2, 0x1.64598p-9f, 0x1.578708p-5f);
    float pb = __fmaf_rn(s2, -0x1.fe4c62p-7f, -0x1.30ec52p-4f);
    pa = __fmaf_rn(pa, s2, 0x1.b2c8e8p-4f);
    pb = __fmaf_rn(pb, s2, -0x1.22c3fp-3f);
    pa = __fmaf_rn(pa, s2, 0x1.996f3ep-3f);
    pb = __fmaf_rn(pb, s2, -0x1.555492p-2f);
    pa = __fmaf_rn(pa, s2, 1.0f);
    float p = __fmaf_rn(pb, s, pa);
    float r = __fmaf_rn(p, q, base);
    r = u2f(f2u(r) | sx);
    if (x <= 0.0f) r = __fadd_rn(r, PI);
    return u2f(f2u(r) | sy);
}

__device__ __forceinline__ float atan2f_svml(float y, float x, const uint2 *tab = RCP14_AB)
{
    bool inr;
    // (the range test comes first in the instruction stream: the compiler hoists it and skips the rest when it fails)
    const float r = atan2f_svml_main(y, x, inr, tab);
    if (__builtin_expect(!inr, 0)) return atan2f_svml_rare(y, x);
    return r;
}

// FM discriminator sample: float32(angle(a * conj(b))) * float32(fs/2pi)   (signal_processing.py:94,97)
// a = samples[i+1], b = samples[i].  NumPy's complex64 multiply is the FMA form; `swapped` selects the
// operand order NumPy's temporary elision produces for frames with N-1 >= 32768.
__device__ __forceinline__ void disc_product(float2 a, float2 b, bool swapped, float &re, float &im)
{
    float c = b.x, d = -b.y;
    re = __fmaf_rn(a.x, c, -__fmul_rn(a.y, d));
    im = swapped ? __fmaf_rn(a.y, c, __fmul_rn(a.x, d)) : __fmaf_rn(a.x, d, __fmul_rn(a.y, c));
}
__device__ __forceinline__ float disc_sample(float2 a, float2 b, float kscale, bool swapped, const uint2 *tab = RCP14_AB)
{
    float re, im;
    disc_product(a, b, swapped, re, im);
    return __fmul_rn(atan2f_svml(im, re, tab), kscale);
}
// the same without the branch to the rare path: `ok` = false where the caller must redo the sample with disc_sample
__device__ __forceinline__ float disc_sample_main(float2 a, float2 b, float kscale, bool swapped, bool &ok, const uint2 *tab = RCP14_AB)
{
    float re, im;
    disc_product(a, b, swapped, re, im);
    return __fmul_rn(atan2f_svml_main(im, re, ok, tab), kscale);
}

// Inner product in the accumulation order of OpenBLAS ddot (kernel/x86_64/ddot.c + ddot_microk_skylakex-2.c), which is
// what np.convolve -> cblas_ddot executes inside scipy.signal.lfilter's FIR branch (signal_processing.py:108):
//   n1 = n & -16 elements go through 4 accumulators of 8 lanes (32 per step, fused multiply-add from +0.0), folded
//   8 -> 4 lanes, then 4 accumulators of 4 lanes (16 per step); s[l] = ((a0[l]+a1[l])+a2[l])+a3[l];
//   dot = (s0+s2)+(s1+s3); the n - n1 tail elements are added one by one with fma(y, x, dot).
// Three renderings of this tree live here and in the kernels: ddot_skx_lane (per-lane length, predicated),
// ddot_skx_uniform (wave-uniform length, rolled loops) and fir65_batch / fir_batch (n = 65, several consecutive
// outputs at once, taps as SGPR operands).  X(j), Y(j): accessors for j = 0..n-1.

// ddot tree with a PER-LANE length n <= 65, fully predicated (no divergent loops): used for the
// left-edge outputs of the FIR (window shorter than 65) where every lane of a wavefront has a different n.
// Phases of the ddot kernel for n <= 65: up to two 32-element steps on 4x8 accumulators (n >= 32, n >= 64),
// fold 8->4, at most one 16-element step on 4x4 accumulators, horizontal sum, up to 15 tail fmas.
template <class FX, class FY>
__device__ __forceinline__ double ddot_skx_lane(FX X, FY Y, int n)
{
    const int n1 = n & -16, n32 = n1 & ~31;
    const bool has32 = n32 >= 32, has64 = n32 >= 64, has16 = (n1 - n32) == 16;
    double a[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int l = 0; l < 4; l++) {
            double lo = 0.0, hi = 0.0;
            {
                const int j = has32 ? 8 * k + l : 0;  // clamp the index when the phase is inactive
                double tlo = __fma_rn(X(j), Y(j), 0.0), thi = __fma_rn(X(j + (has32 ? 4 : 0)), Y(j + (has32 ? 4 : 0)), 0.0);
                const int j2 = has64 ? 32 + 8 * k + l : 0;
                double ulo = __fma_rn(X(j2), Y(j2), tlo), uhi = __fma_rn(X(j2 + (has64 ? 4 : 0)), Y(j2 + (has64 ? 4 : 0)), thi);
                lo = has64 ? ulo : tlo;
                hi = has64 ? uhi : thi;
            }
            double folded = __dadd_rn(lo, hi);
            a[k][l] = has32 ? folded : 0.0;
            const int j3 = has16 ? n32 + 4 * k + l : 0;
            double v = __fma_rn(X(j3), Y(j3), a[k][l]);
            a[k][l] = has16 ? v : a[k][l];
        }
    double s[4];
#pragma unroll
    for (int l = 0; l < 4; l++) s[l] = __dadd_rn(__dadd_rn(__dadd_rn(a[0][l], a[1][l]), a[2][l]), a[3][l]);
    double dot = __dadd_rn(__dadd_rn(s[0], s[2]), __dadd_rn(s[1], s[3]));
    if (n1 == 0) dot = 0.0;
#pragma unroll
    for (int t = 0; t < 15; t++) {
        const int j = (n1 + t < n) ? n1 + t : 0;
        double v = __fma_rn(Y(j), X(j), dot);
        dot = (n1 + t < n) ? v : dot;
    }
    return dot;
}

// ddot tree for a WAVE-UNIFORM length n <= 65 with few live registers (loops stay rolled): used by
// the fused NFM kernel's IIR wave for the 64 left-edge FIR outputs, where lane = frame and all lanes share n.
template <class FX, class FY>
__device__ __forceinline__ double ddot_skx_uniform(FX X, FY Y, int n)
{
    const int n1 = n & -16, n32 = n1 & ~31;
    double dot = 0.0;
    if (n1) {
        double s[4];
#pragma unroll 1
        for (int l = 0; l < 4; l++) {
            double sl = 0.0;
#pragma unroll 1
            for (int k = 0; k < 4; k++) {
                double a = 0.0;
                if (n32) {
                    double lo = 0.0, hi = 0.0;
#pragma unroll 1
                    for (int i = 0; i < n32; i += 32) {
                        lo = __fma_rn(X(i + 8 * k + l), Y(i + 8 * k + l), lo);
                        hi = __fma_rn(X(i + 8 * k + l + 4), Y(i + 8 * k + l + 4), hi);
                    }
                    a = __dadd_rn(lo, hi);
                }
                if (n1 > n32) a = __fma_rn(X(n32 + 4 * k + l), Y(n32 + 4 * k + l), a);
                sl = (k == 0) ? a : __dadd_rn(sl, a);
            }
            s[l] = sl;
        }
        dot = __dadd_rn(__dadd_rn(s[0], s[2]), __dadd_rn(s[1], s[3]));
    }
#pragma unroll 1
    for (int i = n1; i < n; i++) dot = __fma_rn(Y(i), X(i), dot);
    return dot;
}

// Real part of OpenBLAS zdotu (zdot_microk_haswell-2.c) with a real second operand — the complex FIR at
// signal_processing.py:204/:209.
template <class FX, class FY>
__device__ __forceinline__ double zdot_re_skx(FX X, FY Y, int n)
{
    int n1 = n & -8, i = 0;
    double dot = 0.0;
    if (n1) {
        double acc[4][2];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int p = 0; p < 2; p++) acc[a][p] = __fma_rn(X(2 * a + p), Y(2 * a + p), 0.0);
        for (i = 8; i < n1; i += 8) {
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int p = 0; p < 2; p++) acc[a][p] = __fma_rn(X(i + 2 * a + p), Y(i + 2 * a + p), acc[a][p]);
        }
        double c0 = __dadd_rn(__dadd_rn(acc[0][0], acc[1][0]), __dadd_rn(acc[2][0], acc[3][0]));
        double c1 = __dadd_rn(__dadd_rn(acc[0][1], acc[1][1]), __dadd_rn(acc[2][1], acc[3][1]));
        dot = __dadd_rn(c0, c1);
    }
    for (; i < n; i++) dot = __fma_rn(X(i), Y(i), dot);
    return dot;
}

// zdot tree (real part, real second operand) with a PER-LANE length n <= 65, fully predicated: up to eight 8-element
// steps on 4x2 accumulators, c_p = (a0p+a1p)+(a2p+a3p), dot = c0+c1, up to 7 tail fmas (one when n = 65).
template <class FX, class FY>
__device__ __forceinline__ double zdot_re_skx_lane(FX X, FY Y, int n)
{
    const int n1 = n & -8;
    double acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int p = 0; p < 2; p++) acc[a][p] = 0.0;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const bool on = 8 * it < n1;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const int j = on ? 8 * it + 2 * a + p : 0;
                double v = __fma_rn(X(j), Y(j), acc[a][p]);
                acc[a][p] = on ? v : acc[a][p];
            }
    }
    double c0 = __dadd_rn(__dadd_rn(acc[0][0], acc[1][0]), __dadd_rn(acc[2][0], acc[3][0]));
    double c1 = __dadd_rn(__dadd_rn(acc[0][1], acc[1][1]), __dadd_rn(acc[2][1], acc[3][1]));
    double dot = n1 ? __dadd_rn(c0, c1) : 0.0;
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int j = (n1 + t < n) ? n1 + t : 0;
        double v = __fma_rn(X(j), Y(j), dot);
        dot = (n1 + t < n) ? v : dot;
    }
    return dot;
}

// One DF2T biquad step exactly as scipy's _sosfilt (un-fused float64, this association).
struct Biquad { double b0, b1, b2, a1, a2; };
__device__ __forceinline__ double biquad_step(const Biquad &c, double x, double &z0, double &z1)
{
    double xn = __dadd_rn(__dmul_rn(c.b0, x), z0);
    z0 = __dadd_rn(__dsub_rn(__dmul_rn(c.b1, x), __dmul_rn(c.a1, xn)), z1);
    z1 = __dsub_rn(__dmul_rn(c.b2, x), __dmul_rn(c.a2, xn));
    return xn;
}

// The same step for a section whose numerator is exactly [1, 2, 1] (every section but the first of SciPy's
// cheby1 / butter low-pass SOS): 1.0*x == x and 2.0*x == x+x exactly, so dropping the multiplies is bit-identical.
__device__ __forceinline__ double biquad_step_121(const Biquad &c, double x, double &z0, double &z1)
{
    double xn = __dadd_rn(x, z0);
    z0 = __dadd_rn(__dsub_rn(__dadd_rn(x, x), __dmul_rn(c.a1, xn)), z1);
    z1 = __dsub_rn(x, __dmul_rn(c.a2, xn));
    return xn;
}

// np.int16(v * 32767): truncation toward zero, NaN -> 0 (io_manager.py:26, audio_processing.py:37)
__device__ __forceinline__ int16_t pcm16(double a)
{
    double v = __dmul_rn(a, 32767.0);
    return (v != v) ? (int16_t)0 : (int16_t)(int)v;
}

}  // namespace pss
