// pss_npf32.h — NumPy's float32 log10 and abs(complex64) as the AVX512_SKX dispatch evaluates them, bit for bit; shared by the
// demodulator unit (pss_demod.hip, through pss_device.h) and the spectrum unit (pss_fft.hip: the scanner's float32 dB rows).
// The tables are `static`: each translation unit carries its own copy.  HIP's __fmul_rn / __fadd_rn are plain * and + inside header
// functions (round-to-nearest is the default mode), which a unit compiled with the default -ffp-contract=fast (pss_fft.hip)
// contracts into fused multiply-adds after inlining — v_fmaak_f32 a, a, 1e-10 was what the first version compiled to: each
// function therefore switches contraction off for its own body and spells its products and sums with operators; sqrtf / __fdiv_rn must be correctly rounded
// (-fhip-fp32-correctly-rounded-divide-sqrt on both units, pyspecsdr_amd/build.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pss {

#ifndef PSS_F2U_DEFINED
#define PSS_F2U_DEFINED
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
#endif

// np.log10 of a float32 (array or scalar) under NumPy's AVX512_SKX dispatch = Intel SVML __svml_log10f16 (shipped inside
// NumPy's _multiarray_umath): m = mantissa scaled into [0.75, 1.5), k = the matching exponent, r = m - 1, a degree-4
// polynomial in r whose coefficients are looked up by the top four mantissa bits of m (16 intervals), result
// fma(p, r, k * log10(2)) — all float32, round to nearest.  The model below was compared with np.log10 on EVERY positive
// finite float32 (2 139 095 039 values, denormals included): no differing bit.  +-0 -> -inf, x < 0 -> NaN, +inf -> +inf,
// NaN -> NaN as the library's special-value path returns them.
static __constant__ uint32_t L10_C0[16] = {0xbdc9ae9bu, 0xbda6fcf4u, 0xbd8bac76u, 0xbd6bca30u, 0xbd48a99bu, 0xbd2c0a9fu, 0xbd1480dbu, 0xbd00faf2u,
                                    0xbe823aa9u, 0xbe656348u, 0xbe4afbb9u, 0xbe346895u, 0xbe20ffffu, 0xbe103a0bu, 0xbe01a91cu, 0xbde9e84eu};
static __constant__ uint32_t L10_C1[16] = {0x3e13d888u, 0x3e10a87cu, 0x3e0b95c3u, 0x3e057f0bu, 0x3dfde038u, 0x3df080d9u, 0x3de34c1eu, 0x3dd68333u,
                                    0x3dac6e8eu, 0x3dd54a51u, 0x3df30f40u, 0x3e04235du, 0x3e0b7033u, 0x3e102c90u, 0x3e12ebadu, 0x3e141ff8u};
static __constant__ uint32_t L10_C2[16] = {0xbe5e5a9bu, 0xbe5e2677u, 0xbe5d83f5u, 0xbe5c6016u, 0xbe5abd0bu, 0xbe58a6fdu, 0xbe562e02u, 0xbe5362f8u,
                                    0xbe68e27cu, 0xbe646747u, 0xbe619a73u, 0xbe5ff05au, 0xbe5f0570u, 0xbe5e92d0u, 0xbe5e662bu, 0xbe5e5c08u};
static __constant__ uint32_t L10_C3[16] = {0x3ede5bd8u, 0x3ede5b45u, 0x3ede57d8u, 0x3ede4eb1u, 0x3ede3d37u, 0x3ede2166u, 0x3eddf9d9u, 0x3eddc5bbu,
                                    0x3ede08edu, 0x3ede32e7u, 0x3ede4967u, 0x3ede5490u, 0x3ede597fu, 0x3ede5b50u, 0x3ede5bcau, 0x3ede5bd9u};

__device__ __forceinline__ float log10f_np(float x)
{
#pragma clang fp contract(off)
    const uint32_t b = f2u(x);
    if (x != x) return x;
    if ((b & 0x7fffffffu) == 0u) return -INFINITY;
    if (b & 0x80000000u) return NAN;
    if (b == 0x7f800000u) return x;
    int e = (int)(b >> 23);
    uint32_t man = b & 0x7fffffu;
    if (e == 0) {                                     // denormal: normalise the mantissa
        const int sh = __clz((int)man) - 8;           // leading one to bit 23
        man = (man << sh) & 0x7fffffu;
        e = 1 - sh;
    }
    int k = e - 127;
    uint32_t mb;
    if (man >= 0x400000u) { mb = (126u << 23) | man; k += 1; }   // mantissa >= 1.5: halve it
    else mb = (127u << 23) | man;
    const int idx = (int)(mb >> 19) & 15;
    const float r = u2f(mb) - 1.0f;
    float p = __fmaf_rn(u2f(L10_C0[idx]), r, u2f(L10_C1[idx]));
    p = __fmaf_rn(p, r, u2f(L10_C2[idx]));
    p = __fmaf_rn(p, r, u2f(L10_C3[idx]));
    const float kl = (float)k * u2f(0x3e9a209bu);
    return __fmaf_rn(p, r, kl);
}

// numpy.abs(complex64), AVX512F loop: mx * sqrt(fma(r, r, 1)), r = mn / mx (IEEE div and sqrt).
// NB: sqrtf()/operator/ are correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt, set in build.py);
// __fsqrt_rn() is NOT (it lowers to the native approximate v_sqrt_f32).
__device__ __forceinline__ float cabsf_np(float re, float im)
{
#pragma clang fp contract(off)
    float a = fabsf(re), b = fabsf(im);
    float mx = a > b ? a : b, mn = a > b ? b : a;
    if (mx == 0.0f) return 0.0f;
    const float r = mn / mx;
    const float h = sqrtf(__fmaf_rn(r, r, 1.0f));
    return mx * h;
}

// One bin of the scanner's power spectrum (pyspecsdr.py:2542-2543, :1050-1051): 10 * np.log10(np.abs(spectrum) ** 2 + 1e-10) with
// spectrum = np.fft.fft(complex64) — which NumPy 2.2 computes in DOUBLE and rounds to complex64 (identical bits to
// fft(x.astype(complex128)).astype(complex64) for every length tried) — so: round the float64 bin to float32 components, then
// float32 all the way (npy_hypotf form of np.abs, x * x, + float32(1e-10), SVML log10, * 10).
// the four coefficient sets side by side: one 16-byte load per evaluation in the hot scanner epilogue
static __constant__ uint4 L10_PACK[16] = {
    {0xbdc9ae9bu, 0x3e13d888u, 0xbe5e5a9bu, 0x3ede5bd8u}, {0xbda6fcf4u, 0x3e10a87cu, 0xbe5e2677u, 0x3ede5b45u},
    {0xbd8bac76u, 0x3e0b95c3u, 0xbe5d83f5u, 0x3ede57d8u}, {0xbd6bca30u, 0x3e057f0bu, 0xbe5c6016u, 0x3ede4eb1u},
    {0xbd48a99bu, 0x3dfde038u, 0xbe5abd0bu, 0x3ede3d37u}, {0xbd2c0a9fu, 0x3df080d9u, 0xbe58a6fdu, 0x3ede2166u},
    {0xbd1480dbu, 0x3de34c1eu, 0xbe562e02u, 0x3eddf9d9u}, {0xbd00faf2u, 0x3dd68333u, 0xbe5362f8u, 0x3eddc5bbu},
    {0xbe823aa9u, 0x3dac6e8eu, 0xbe68e27cu, 0x3ede08edu}, {0xbe656348u, 0x3dd54a51u, 0xbe646747u, 0x3ede32e7u},
    {0xbe4afbb9u, 0x3df30f40u, 0xbe619a73u, 0x3ede4967u}, {0xbe346895u, 0x3e04235du, 0xbe5ff05au, 0x3ede5490u},
    {0xbe20ffffu, 0x3e0b7033u, 0xbe5f0570u, 0x3ede597fu}, {0xbe103a0bu, 0x3e102c90u, 0xbe5e92d0u, 0x3ede5b50u},
    {0xbe01a91cu, 0x3e12ebadu, 0xbe5e662bu, 0x3ede5bcau}, {0xbde9e84eu, 0x3e141ff8u, 0xbe5e5c08u, 0x3ede5bd9u}};

// tab: L10_PACK itself (constant memory: every lookup a vector load in the middle of the bin's dependent chain) or a copy the kernel
// keeps in LDS (the scanner kernels: 16 entries)
__device__ __forceinline__ float scan_db_np(double re, double im, const uint4 *tab = L10_PACK)
{
#pragma clang fp contract(off)
    // np.abs(complex64): cabsf_np without its early return (a select instead: the scanner evaluates this for every bin)
    const float ar = fabsf((float)re), ai = fabsf((float)im);
    const float mx = ar > ai ? ar : ai, mn = ar > ai ? ai : ar;
    const float r = mn / mx;
    const float h = sqrtf(__fmaf_rn(r, r, 1.0f));
    const float a = mx == 0.0f ? 0.0f : mx * h;
    const float sq = a * a;                 // separate statements under contract(off): a fused a * a + 1e-10 would differ
    const float p = sq + 1e-10f;
    // log10f_np for p >= 1e-10 (never zero, negative or denormal here); +inf and NaN pass through
    const uint32_t b = f2u(p);
    const uint32_t man = b & 0x7fffffu;
    const bool up = man >= 0x400000u;
    const uint32_t mb = (up ? (126u << 23) : (127u << 23)) | man;
    const int k = (int)(b >> 23) - 127 + (up ? 1 : 0);
    const uint4 c = tab[(mb >> 19) & 15u];
    const float q = u2f(mb) - 1.0f;
    float t = __fmaf_rn(u2f(c.x), q, u2f(c.y));
    t = __fmaf_rn(t, q, u2f(c.z));
    t = __fmaf_rn(t, q, u2f(c.w));
    const float kl = (float)k * u2f(0x3e9a209bu);
    const float lg = b >= 0x7f800000u ? p : __fmaf_rn(t, q, kl);
    return 10.0f * lg;
}

}  // namespace pss
