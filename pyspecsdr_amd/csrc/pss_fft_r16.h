// pss_fft_r16.h — register-resident radix-16 FFT kernel for the spectrum path (N = 256 * R3, R3 in {1,2,4,8,16}).
//
// Decomposition N = 16 (n2) x 16 (m2) x R3 (m1), T = N/16 threads per frame, 16 points per thread:
//   stage 1  thread n1 loads x[n1 + T*n2] (n2 = 0..15; each n2 is one coalesced row), applies the Hamming window,
//            runs a 16-point DFT over n2 in registers and multiplies by W_N^(n1*k2).
//   exchange 1 through LDS (E1[k2][n1], rows padded by 4 so the strided ds_read_b128 below is conflict-free)
//   stage 2  thread (k2, m1) gathers y[n1 = m1 + R3*m2][k2], runs a 16-point DFT over m2, multiplies by W_T^(m1*j2).
//   exchange 2 through LDS (E2[m1][bf], bf = 16*j2 + k2, planes padded by 2)
//   stage 3  thread rho takes 16/R3 radix-R3 butterflies bf = rho + T*c, whose outputs are bins
//            k = 256*j1 + bf — so for fixed (j1, c) the T threads of a frame hold T CONSECUTIVE bins and the
//            dB row is written with coalesced stores directly from registers (no staging, no digit reversal).
// All butterflies are float64 (see pss_fft.hip for why); the stage-1 twiddles and the window are per-thread
// constants and stay in registers while the workgroup loops over frames.
#pragma once
#include <hip/hip_runtime.h>

#include "pss_npf32.h"
#include "pss_device.h"

namespace pss_r16 {

// Complex product with explicit fused multiply-adds (4 instructions instead of 6).  pss_device.h switches contraction off for
// everything that includes it — the demodulators' bit-exactness contract — so nothing in the transforms was fused until round 3;
// their results are tolerance-bound (or rounded to float32 with ~1e-15 of margin), and the float64 pipe is what paces them.
#ifdef PSS_EXP_NOFMA
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double power_of(double2 X) { return X.x * X.x + X.y * X.y + 1e-10; }
#else
__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ double power_of(double2 X) { return fma(X.x, X.x, fma(X.y, X.y, 1e-10)); }   // |X|^2 + 1e-10
#endif
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

constexpr int brev(int i, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; b++) r |= ((i >> b) & 1) << (bits - 1 - b);
    return r;
}
constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }

// multiply by W_len^j = exp(-2 pi i j / len) for compile-time (j, len), len <= 16
template <int J, int LEN>
__device__ __forceinline__ double2 twc(double2 a)
{
    constexpr int j16 = (J * (16 / LEN)) & 15;  // as a 16th root index
    constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, H = 0.70710678118654752440;
    if constexpr (j16 == 0) return a;
    else if constexpr (j16 == 4) return make_double2(a.y, -a.x);
    else if constexpr (j16 == 8) return make_double2(-a.x, -a.y);
    else if constexpr (j16 == 12) return make_double2(-a.y, a.x);
    else if constexpr (j16 == 2) return make_double2((a.x + a.y) * H, (a.y - a.x) * H);
    else if constexpr (j16 == 6) return make_double2((a.y - a.x) * H, -(a.x + a.y) * H);
    else if constexpr (j16 == 1) return cmul(a, make_double2(C1, -S1));
    else if constexpr (j16 == 3) return cmul(a, make_double2(S1, -C1));
    else if constexpr (j16 == 5) return cmul(a, make_double2(-S1, -C1));
    else if constexpr (j16 == 7) return cmul(a, make_double2(-C1, -S1));
    else return a;  // j >= len/2 never occurs in a DIF butterfly
}

// In-register DIF FFT of R points (R = 2,4,8,16); result X[k] is left at v[brev(k)].
template <int R, int LEN = R, int BASE = 0>
struct Dif {
    template <int J>
    static __device__ __forceinline__ void bf(double2 (&v)[R])
    {
        if constexpr (J < LEN / 2) {
            double2 a = v[BASE + J], b = v[BASE + J + LEN / 2];
            v[BASE + J] = cadd(a, b);
            v[BASE + J + LEN / 2] = twc<J, LEN>(csub(a, b));
            bf<J + 1>(v);
        }
    }
    static __device__ __forceinline__ void run(double2 (&v)[R])
    {
        if constexpr (LEN >= 2) {
            bf<0>(v);
            Dif<R, LEN / 2, BASE>::run(v);
            Dif<R, LEN / 2, BASE + LEN / 2>::run(v);
        }
    }
};

template <int R>
__device__ __forceinline__ void fft_reg(double2 (&v)[R])
{
    if constexpr (R > 1) Dif<R>::run(v);
}

// 10 log10(pw), pw = |X|^2 + 1e-10 (>= 1e-10, float64), evaluated to float64 accuracy and rounded ONCE to float32.
// compute_fft's own result is float64 (np.abs / ** 2 / np.log10 on a complex128 transform, signal_processing.py:250-262); a
// float64-accurate evaluation rounded to float32 equals the float32 rounding of the reference's value except where that value
// lies within the two evaluations' ~1e-15 of a float32 rounding boundary (~1e-8 of the bins) — the previous float32 evaluation
// (hardware log2) was 1-2 float32 ulp off on most bins, at about the same instruction count.
//   pw = 2^e z, z folded into [0.75, 1.5);  c = 0.75 + i / 128 the nearest of 97 centres (z = 1 is one: relative accuracy is
//   kept where dB -> 0);  r = z / c - 1 (|r| <= 0.0053) with 1 / c from the table;  log10 pw = e log10 2 + log10 c + ln(1 + r) / ln 10.
// DB_TAB[i] = {double(1 / c_i), -log10 of THAT double} (tools/make_db_table.py: extended precision), so the table's rounding cancels.
static __constant__ double2 DB_TAB[97] = {
    {0x1.5555555555555p+0, -0x1.ffbfc2bbc7802p-4},
    {0x1.51d07eae2f815p+0, -0x1.ed50a4a26eafbp-4},
    {0x1.4e5e0a72f0539p+0, -0x1.db11ed766abf2p-4},
    {0x1.4afd6a052bf5bp+0, -0x1.c902a19e65114p-4},
    {0x1.47ae147ae147bp+0, -0x1.b721cd17157e3p-4},
    {0x1.446f86562d9fbp+0, -0x1.a56e8325f5c87p-4},
    {0x1.4141414141414p+0, -0x1.93e7de0fc3e7fp-4},
    {0x1.3e22cbce4a902p+0, -0x1.828cfed29a212p-4},
    {0x1.3b13b13b13b14p+0, -0x1.715d0ce367afdp-4},
    {0x1.3813813813814p+0, -0x1.605735ee985f4p-4},
    {0x1.3521cfb2b78c1p+0, -0x1.4f7aad9bbcbaep-4},
    {0x1.323e34a2b10bfp+0, -0x1.3ec6ad5407866p-4},
    {0x1.2f684bda12f68p+0, -0x1.2e3a740b7800dp-4},
    {0x1.2c9fb4d812ca0p+0, -0x1.1dd5460c8b170p-4},
    {0x1.29e4129e4129ep+0, -0x1.0d966cc6500f8p-4},
    {0x1.27350b8812735p+0, -0x1.fafa6d397efdbp-5},
    {0x1.2492492492492p+0, -0x1.db11ed766abf1p-5},
    {0x1.21fb78121fb78p+0, -0x1.bb7209d1e24e4p-5},
    {0x1.1f7047dc11f70p+0, -0x1.9c197abf00dd3p-5},
    {0x1.1cf06ada2811dp+0, -0x1.7d070145f4fd8p-5},
    {0x1.1a7b9611a7b96p+0, -0x1.5e3966b7e9294p-5},
    {0x1.1811811811812p+0, -0x1.3faf7c6630614p-5},
    {0x1.15b1e5f75270dp+0, -0x1.21681b5c8c213p-5},
    {0x1.135c81135c811p+0, -0x1.0362241e638eap-5},
    {0x1.1111111111111p+0, -0x1.cb38fccd8bfdap-6},
    {0x1.0ecf56be69c90p+0, -0x1.902c31d62a847p-6},
    {0x1.0c9714fbcda3bp+0, -0x1.559bd2406c3c1p-6},
    {0x1.0a6810a6810a7p+0, -0x1.1b85d6044e9bbp-6},
    {0x1.0842108421084p+0, -0x1.c3d0837784c3ap-7},
    {0x1.0624dd2f1a9fcp+0, -0x1.51824c7587eb5p-7},
    {0x1.0410410410410p+0, -0x1.c03a80ae5e038p-8},
    {0x1.0204081020408p+0, -0x1.be76bd77b4fb5p-9},
    {0x1.0000000000000p+0, 0x0.0p+0},
    {0x1.fc07f01fc07f0p-1, 0x1.bafd47221ed34p-9},
    {0x1.f81f81f81f820p-1, 0x1.b9476a4fcd0f3p-8},
    {0x1.f44659e4a4271p-1, 0x1.49b085144368ep-7},
    {0x1.f07c1f07c1f08p-1, 0x1.b5e908eb13789p-7},
    {0x1.ecc07b301ecc0p-1, 0x1.10a83a8446c7fp-6},
    {0x1.e9131abf0b767p-1, 0x1.45f4f5acb8be3p-6},
    {0x1.e573ac901e574p-1, 0x1.7adc3df3b1ff3p-6},
    {0x1.e1e1e1e1e1e1ep-1, 0x1.af5f92b00e611p-6},
    {0x1.de5d6e3f8868ap-1, 0x1.e3806acbd0593p-6},
    {0x1.dae6076b981dbp-1, 0x1.0ba01a816ffffp-5},
    {0x1.d77b654b82c34p-1, 0x1.25502c0fc3148p-5},
    {0x1.d41d41d41d41dp-1, 0x1.3ed1199a5e427p-5},
    {0x1.d0cb58f6ec074p-1, 0x1.58238eeb353dcp-5},
    {0x1.cd85689039b0bp-1, 0x1.71483427d2a97p-5},
    {0x1.ca4b3055ee191p-1, 0x1.8a3fadeb847f4p-5},
    {0x1.c71c71c71c71cp-1, 0x1.a30a9d609efedp-5},
    {0x1.c3f8f01c3f8f0p-1, 0x1.bba9a058dfd85p-5},
    {0x1.c0e070381c0e0p-1, 0x1.d41d5164facb7p-5},
    {0x1.bdd2b899406f7p-1, 0x1.ec6647eb5880bp-5},
    {0x1.bacf914c1bad0p-1, 0x1.02428c1f08014p-4},
    {0x1.b7d6c3dda338bp-1, 0x1.0e3d29d81165fp-4},
    {0x1.b4e81b4e81b4fp-1, 0x1.1a23445501814p-4},
    {0x1.b2036406c80d9p-1, 0x1.25f5215eb594ap-4},
    {0x1.af286bca1af28p-1, 0x1.31b3055c4711ap-4},
    {0x1.ac5701ac5701bp-1, 0x1.3d5d335c53178p-4},
    {0x1.a98ef606a63bep-1, 0x1.48f3ed1df48f9p-4},
    {0x1.a6d01a6d01a6dp-1, 0x1.5477731973e85p-4},
    {0x1.a41a41a41a41ap-1, 0x1.5fe80488af4fep-4},
    {0x1.a16d3f97a4b02p-1, 0x1.6b45df6f3e2c8p-4},
    {0x1.9ec8e951033d9p-1, 0x1.769140a2526fdp-4},
    {0x1.9c2d14ee4a102p-1, 0x1.81ca63d05a448p-4},
    {0x1.999999999999ap-1, 0x1.8cf183886480bp-4},
    {0x1.970e4f80cb872p-1, 0x1.9806d9414a20cp-4},
    {0x1.948b0fcd6e9e0p-1, 0x1.a30a9d609efebp-4},
    {0x1.920fb49d0e229p-1, 0x1.adfd07416be06p-4},
    {0x1.8f9c18f9c18fap-1, 0x1.b8de4d3ab3d97p-4},
    {0x1.8d3018d3018d3p-1, 0x1.c3aea4a5c6effp-4},
    {0x1.8acb90f6bf3aap-1, 0x1.ce6e41e463da3p-4},
    {0x1.886e5f0abb04ap-1, 0x1.d91d5866aa99ap-4},
    {0x1.8618618618618p-1, 0x1.e3bc1ab0e1a00p-4},
    {0x1.83c977ab2beddp-1, 0x1.ee4aba610f205p-4},
    {0x1.8181818181818p-1, 0x1.f8c9683468191p-4},
    {0x1.7f405fd017f40p-1, 0x1.019c2a064b487p-3},
    {0x1.7d05f417d05f4p-1, 0x1.06cbd67a6c3b7p-3},
    {0x1.7ad2208e0ecc3p-1, 0x1.0bf3d0937c41dp-3},
    {0x1.78a4c8178a4c8p-1, 0x1.11142f0811357p-3},
    {0x1.767dce434a9b1p-1, 0x1.162d082ac9d10p-3},
    {0x1.745d1745d1746p-1, 0x1.1b3e71ec94f7ap-3},
    {0x1.724287f46debcp-1, 0x1.204881dee8777p-3},
    {0x1.702e05c0b8170p-1, 0x1.254b4d35e7d3dp-3},
    {0x1.6e1f76b4337c7p-1, 0x1.2a46e8ca7ba29p-3},
    {0x1.6c16c16c16c17p-1, 0x1.2f3b691c5a000p-3},
    {0x1.6a13cd1537290p-1, 0x1.3428e2540096ep-3},
    {0x1.6816816816817p-1, 0x1.390f6844a0b82p-3},
    {0x1.661ec6a5122f9p-1, 0x1.3def0e6dfdf85p-3},
    {0x1.642c8590b2164p-1, 0x1.42c7e7fe3fc02p-3},
    {0x1.623fa77016240p-1, 0x1.479a07d3b6410p-3},
    {0x1.6058160581606p-1, 0x1.4c65807e93337p-3},
    {0x1.5e75bb8d015e7p-1, 0x1.512a644296c3ep-3},
    {0x1.5c9882b931057p-1, 0x1.55e8c518b10f9p-3},
    {0x1.5ac056b015ac0p-1, 0x1.5aa0b4b0988fap-3},
    {0x1.58ed2308158edp-1, 0x1.5f52447255c93p-3},
    {0x1.571ed3c506b3ap-1, 0x1.63fd857fc49bap-3},
    {0x1.5555555555555p-1, 0x1.68a288b60b7fdp-3}};

// the float64 value itself (the reference's own row type, pss_spectrum_db_f64): within ~1e-14 dB of np.log10's
__device__ __forceinline__ double db64_of_exact(double pw, const double2 *tab = DB_TAB)
{
    const unsigned hi = (unsigned)__double2hiint(pw), lo = (unsigned)__double2loint(pw);
    const bool up = (hi & 0xfffffu) >= 0x80000u;
    const int e = (int)(hi >> 20) - 1023 + (up ? 1 : 0);
    const double z = __hiloint2double((int)((hi & 0xfffffu) | (up ? 0x3fe00000u : 0x3ff00000u)), (int)lo);
    const int i = (int)fma(z, 128.0, -95.5);
    const double2 tc = tab[i];
    const double r = fma(z, tc.x, -1.0);
    double p = fma(r, -0x1.5555555555555p-3, 0x1.999999999999ap-3);
    p = fma(p, r, -0.25);
    p = fma(p, r, 0x1.5555555555555p-2);
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.0);
    double res = fma(p * r, 0x1.bcb7b1526e50ep-2, tc.y);
    res = fma((double)e, 0x1.34413509f79ffp-2, res);
    return hi >= 0x7ff00000u ? pw : 10.0 * res;                          // +inf, NaN
}

__device__ __forceinline__ float db_of_exact(double pw, const double2 *tab = DB_TAB)
{
    const unsigned hi = (unsigned)__double2hiint(pw), lo = (unsigned)__double2loint(pw);
    const bool up = (hi & 0xfffffu) >= 0x80000u;                         // mantissa >= 1.5: halve
    const int e = (int)(hi >> 20) - 1023 + (up ? 1 : 0);                 // pw > 0: no sign bit to mask
    const double z = __hiloint2double((int)((hi & 0xfffffu) | (up ? 0x3fe00000u : 0x3ff00000u)), (int)lo);
    const int i = (int)fma(z, 128.0, -95.5);                             // round((z - 0.75) * 128), 0 .. 96
    const double2 tc = tab[i];
    const double r = fma(z, tc.x, -1.0);
    double p = fma(r, -0x1.5555555555555p-3, 0x1.999999999999ap-3);      // -1/6, 1/5
    p = fma(p, r, -0.25);
    p = fma(p, r, 0x1.5555555555555p-2);                                 // 1/3
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.0);
    double res = fma(p * r, 0x1.bcb7b1526e50ep-2, tc.y);                 // ln(1 + r) / ln 10 + log10 c
    res = fma((double)e, 0x1.34413509f79ffp-2, res);                     // + e log10 2
    const float out = (float)(10.0 * res);
    return hi >= 0x7ff00000u ? (float)pw : out;                          // +inf, NaN
}

// The same quantity evaluated in float32 (hardware log2; an atanh series near pw = 1 keeps the RELATIVE error ~1e-7 where dB -> 0):
// within 1-2 float32 ulp of the value above — far inside the 1e-4 relative contract of compute_fft's rows — at about half the
// issue slots.  This is the default; option "db_exact" = 1 selects db_of_exact (spectrum kernel 0.17 -> 0.20 ms at cfg 2, bench step + 3 %).
__device__ __forceinline__ float db_of_fast(double pw)
{
    const float t = (float)(pw - 1.0);
    const float far = 3.0102999566398120f * __log2f((float)pw);
    const float s = t * __builtin_amdgcn_rcpf(2.0f + t);
    const float s2 = s * s;
#ifdef PSS_EXP_NOFMA
    const float p = s * (2.0f + s2 * (0.66666667f + s2 * (0.4f + s2 * (0.28571429f + s2 * 0.22222222f))));
#else
    const float p = s * fmaf(s2, fmaf(s2, fmaf(s2, fmaf(s2, 0.22222222f, 0.28571429f), 0.4f), 0.66666667f), 2.0f);
#endif
    return fabsf(t) < 0.25f ? 4.342944819032518f * p : far;
}

// flags of the spectrum kernels (wave-uniform): bit 0 = scanner rows by NumPy's float32 chain, bit 1 = db_of_exact
constexpr int FLAG_SCAN_EXACT = 1, FLAG_DB_EXACT = 2;
__device__ __forceinline__ float db_of(double pw, int flags = 0, const double2 *tab = DB_TAB)
{
    return (flags & FLAG_DB_EXACT) ? db_of_exact(pw, tab) : db_of_fast(pw);
}

// Buffer addressing: resource descriptor (scalar base + size) + ONE per-lane byte offset + a scalar offset per access.  With
// plain pointers the compiler materialises a 64-bit per-lane address for every row of the frame, hoists the 48 of them out
// of the persistent frame loop and spills them (measured: 180 VGPRs of spills in a 128-VGPR kernel).
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float2 buf_load_f2(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    const v2u_t v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__device__ __forceinline__ void buf_store_f32(__amdgpu_buffer_rsrc_t r, int voff, int soff, float x)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_f64(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x)
{
    v2u_t v;
    v.x = (unsigned)__double2loint(x);
    v.y = (unsigned)__double2hiint(x);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// DPP control "wave_rol:1" (gfx9): every lane reads its upper neighbour, lane 63 reads lane 0
constexpr int DPP_NEXT_LANE = 0x134;

template <int LOG_R3>
struct Cfg {
    static constexpr int R3 = 1 << LOG_R3;
    static constexpr int T = 16 * R3;          // threads per frame
    static constexpr int N = 16 * T;           // FFT length
    static constexpr int FPW = 256 / T;        // frames per 256-thread workgroup
    // Row / plane paddings of the two exchanges, in complex elements (16 bytes = one 4-bank slot), from the per-instruction lane
    // groups of MI355X_MICROARCH.md (LDS): ds_read_b128 is served in four NON-contiguous groups of 16 lanes ({0-3, 12-15, 20-27}, ...)
    // against 64 banks (16 slots), ds_write_b128 in eight groups of 8 CONSECUTIVE lanes against 32 banks (8 slots).
    //  * exchange-1 reads: lane t takes row t / R3, column t % R3 + R3 m2 -> slot (t / R3) pad1 + t % R3 (mod 16); with pad1 = R3
    //    (mod 16) that is t (mod 16), and every read group holds 16 different residues;
    //  * exchange-2 writes: lane t stores to plane t % R3, column t / R3 + 16 j2 -> slot (t % R3) pad2 + t / R3 (mod 8), all
    //    different over 8 consecutive lanes iff pad2 = 8 / R3 (R3 <= 8) or odd (R3 = 16);
    //  * the stage-2 twiddle rows W_T^(m1 j2) are TW2S = 17 apart (16 apart, the R3 rows of a lane group met on one slot: an
    //    R3-way conflict on each of the 15 reads).
    // Rounds 1-2 used pad1 = 4, pad2 = 2, rows of 16 for every length: SQ_LDS_BANK_CONFLICT was 29 / 58 % of the LDS cycles at
    // 1024 / 2048 points (profiles/r03_lds_bank_conflicts.txt has both layouts).
#ifdef PSS_EXP_OLDPAD
    static constexpr int E1_STRIDE = T + 4;
    static constexpr int E2_STRIDE = 256 + 2;
    static constexpr int TW2S = 16;
#else
    static constexpr int E1_STRIDE = T + (R3 == 1 ? 4 : R3 % 16);                     // complex elements per k2 row
    static constexpr int E2_STRIDE = 256 + (R3 == 1 ? 2 : R3 <= 8 ? 8 / R3 : 1);     // complex elements per m1 plane
    static constexpr int TW2S = 17;                                                   // complex elements per stage-2 twiddle row
#endif
    static constexpr int TW2 = R3 * TW2S;                                             // stage-2 twiddle table, complex elements
    static __device__ __forceinline__ int tw2_slot(int i) { return (i / 16) * TW2S + (i % 16); }   // of entry m1 * 16 + j2
    static constexpr int EX = (16 * E1_STRIDE > R3 * E2_STRIDE) ? 16 * E1_STRIDE : R3 * E2_STRIDE;  // per frame
    static constexpr size_t LDS = (size_t)FPW * EX * sizeof(double2) + (size_t)TW2 * sizeof(double2);
};

// Stages 1b..3 for one frame whose 16 (windowed) stage-1 inputs are already in v[]: 16-point DFT, twiddle, exchange,
// 16-point DFT, twiddle, exchange, radix-R3 butterflies.  emit(i, kp, X) is called for the thread's 16 results, kp being
// the frequency index inside this N = 256*R3 transform.  Contains three workgroup barriers; the caller adds the one
// that separates consecutive frames.
// Synchronisation between an LDS exchange's writes and reads.  WAVE_LOCAL: the T <= 64 threads of a frame are lanes of ONE
// wavefront (k_spectrum_r16 at N <= 1024: tid = frame * T + t); a wavefront's LDS instructions execute in order, so only
// the compiler has to be kept from reordering them — no s_barrier, and the frames of a workgroup no longer wait for
// each other three times per transform.
template <bool WAVE_LOCAL>
__device__ __forceinline__ void frame_sync()
{
    if constexpr (WAVE_LOCAL) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int LOG_R3, bool WAVE_LOCAL = false, class Emit>
__device__ __forceinline__ void r16_core(double2 (&v)[16], double2 *ex, const double2 (&tw1)[16], const double2 *tw2,
                                         int t, Emit emit)
{
    using C = Cfg<LOG_R3>;
    constexpr int R3 = C::R3, T = C::T;
    const int k2s = t / R3, m1s = t % R3;  // stage-2 role
    fft_reg<16>(v);
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
        double2 y = v[brev(k2, 4)];
        if (k2) y = cmul(y, tw1[k2]);
        ex[k2 * C::E1_STRIDE + t] = y;
    }
    frame_sync<WAVE_LOCAL>();
#pragma unroll
    for (int m2 = 0; m2 < 16; m2++) v[m2] = ex[k2s * C::E1_STRIDE + m1s + R3 * m2];
    frame_sync<WAVE_LOCAL>();
#ifndef PSS_EXP_TW2_AT_USE
    // the stage-2 twiddles W_T^(m1 j2) come from LDS: requested TWD outputs ahead of their use, the first ones before the
    // butterflies (read at the use, each of the 15 products waited for an LDS round trip with six instructions to cover it)
    constexpr int TWD = 4;
    const double2 *twp = tw2 + m1s * C::TW2S;
    double2 tq[TWD];
    if constexpr (R3 > 1) {
#pragma unroll
        for (int d = 0; d < TWD; d++) tq[d] = twp[1 + d];
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    fft_reg<16>(v);
#ifdef PSS_EXP_TW2_AT_USE
#pragma unroll
    for (int j2 = 0; j2 < 16; j2++) {
        double2 z = v[brev(j2, 4)];
        if (R3 > 1) z = cmul(z, tw2[m1s * C::TW2S + j2]);
        ex[m1s * C::E2_STRIDE + 16 * j2 + k2s] = z;
    }
#else
    {
#pragma unroll
        for (int j2 = 0; j2 < 16; j2++) {   // j2 = 0 is a product with 1
            double2 z = v[brev(j2, 4)];
            if constexpr (R3 > 1) {
                if (j2 >= 1) {
                    z = cmul(z, tq[(j2 - 1) % TWD]);
                    if (j2 + TWD < 16) tq[(j2 - 1) % TWD] = twp[j2 + TWD];
                }
            }
            ex[m1s * C::E2_STRIDE + 16 * j2 + k2s] = z;
        }
    }
#endif
    frame_sync<WAVE_LOCAL>();
#pragma unroll
    for (int c = 0; c < 16 / R3; c++) {
        double2 b[R3];
#pragma unroll
        for (int m1 = 0; m1 < R3; m1++) b[m1] = ex[m1 * C::E2_STRIDE + t + T * c];
        fft_reg<R3>(b);
#pragma unroll
        for (int j1 = 0; j1 < R3; j1++) emit(c * R3 + j1, 256 * j1 + t + T * c, b[brev(j1, LOG_R3)]);
    }
}

// The same transform with the two LDS exchanges done one COMPONENT at a time (real parts, then imaginary parts): the
// exchange buffer shrinks to EX doubles per frame, which doubles the workgroups a CU can hold (the kernel is LDS-capacity
// bound at 2 workgroups per CU otherwise) at the price of twice the barriers.  Same arithmetic, same results.
template <int LOG_R3, bool WAVE_LOCAL = false, class Emit>
__device__ __forceinline__ void r16_core_split(double2 (&v)[16], double *ex, const double2 (&tw1)[16], const double2 *tw2,
                                               int t, Emit emit)
{
    using C = Cfg<LOG_R3>;
    constexpr int R3 = C::R3, T = C::T;
    const int k2s = t / R3, m1s = t % R3;  // stage-2 role
    fft_reg<16>(v);
    double2 y[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) {
        y[k2] = v[brev(k2, 4)];
        if (k2) y[k2] = cmul(y[k2], tw1[k2]);
    }
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) ex[k2 * C::E1_STRIDE + t] = y[k2].x;
    frame_sync<WAVE_LOCAL>();
#pragma unroll
    for (int m2 = 0; m2 < 16; m2++) v[m2].x = ex[k2s * C::E1_STRIDE + m1s + R3 * m2];
    frame_sync<WAVE_LOCAL>();
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) ex[k2 * C::E1_STRIDE + t] = y[k2].y;
    frame_sync<WAVE_LOCAL>();
#pragma unroll
    for (int m2 = 0; m2 < 16; m2++) v[m2].y = ex[k2s * C::E1_STRIDE + m1s + R3 * m2];
    frame_sync<WAVE_LOCAL>();
    fft_reg<16>(v);
#pragma unroll
    for (int j2 = 0; j2 < 16; j2++) {
        y[j2] = v[brev(j2, 4)];
        if (R3 > 1) y[j2] = cmul(y[j2], tw2[m1s * C::TW2S + j2]);
    }
#pragma unroll
    for (int j2 = 0; j2 < 16; j2++) ex[m1s * C::E2_STRIDE + 16 * j2 + k2s] = y[j2].x;
    frame_sync<WAVE_LOCAL>();
#pragma unroll
    for (int c = 0; c < 16 / R3; c++)
#pragma unroll
        for (int m1 = 0; m1 < R3; m1++) v[c * R3 + m1].x = ex[m1 * C::E2_STRIDE + t + T * c];
    frame_sync<WAVE_LOCAL>();
#pragma unroll
    for (int j2 = 0; j2 < 16; j2++) ex[m1s * C::E2_STRIDE + 16 * j2 + k2s] = y[j2].y;
    frame_sync<WAVE_LOCAL>();
#pragma unroll
    for (int c = 0; c < 16 / R3; c++)
#pragma unroll
        for (int m1 = 0; m1 < R3; m1++) v[c * R3 + m1].y = ex[m1 * C::E2_STRIDE + t + T * c];
#pragma unroll
    for (int c = 0; c < 16 / R3; c++) {
        double2 b[R3];
#pragma unroll
        for (int m1 = 0; m1 < R3; m1++) b[m1] = v[c * R3 + m1];
        fft_reg<R3>(b);
#pragma unroll
        for (int j1 = 0; j1 < R3; j1++) emit(c * R3 + j1, 256 * j1 + t + T * c, b[brev(j1, LOG_R3)]);
    }
}

// EXACT: dB rows by db_of_exact (compile-time: the float64 evaluation next to the float32 one in ONE kernel costs both of them
// registers — measured 0.41 -> 0.45 ms for the default path at 131072 x 1024)
// ONE (frames of two or more wavefronts, i.e. N >= 2048): one frame per workgroup instead of 256 / T — the three exchange barriers of a
// transform then hold up the frame's own wavefronts only, not the other frame's as well.
// D64 (with EXACT): `db` points at float64 rows — the reference's own row type (compute_fft returns float64): the value db_of_exact rounds,
// stored unrounded, 8 bytes per bin (pss_spectrum_db_f64, the cell-exact pipeline pss_frame_pipeline_nfm_f64).
template <int LOG_R3, bool SCAN, bool SPLIT = false, bool PREFETCH = false, bool EXACT = false, bool ONE = false, bool D64 = false>
__global__ __launch_bounds__(256) void k_spectrum_r16(const float2 *__restrict__ iq, float *__restrict__ db,
                                                      const double2 *__restrict__ tw, const double *__restrict__ win,
                                                      long n_frames, float *__restrict__ peak, double *__restrict__ bw,
                                                      int *__restrict__ count, double bin_hz, int flags)
{
    using C = Cfg<LOG_R3>;
    constexpr int R3 = C::R3, T = C::T, N = C::N, FPW = ONE ? 1 : C::FPW;
    static_assert(!ONE || (T >= 128 && !SCAN && !SPLIT), "one frame per workgroup: compute_fft rows of frames of >= 2 wavefronts");
    static_assert(!D64 || (EXACT && !SCAN), "float64 rows: compute_fft rows by the float64 evaluation");
    constexpr int OB = D64 ? 8 : 4;            // bytes per output bin
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex_all = reinterpret_cast<double2 *>(smem);
    double2 *tw2 = SPLIT ? reinterpret_cast<double2 *>(smem + (size_t)FPW * C::EX * sizeof(double))
                         : ex_all + (size_t)FPW * C::EX;  // W_T^(m1*j2), [m1][j2]
    __shared__ float red_f[4];
    __shared__ int red_i[4];
    __shared__ uint4 l10[SCAN ? 16 : 1];       // the float32 log10 model's coefficient sets (pss_npf32.h): LDS copy for the scanner's exact rows
    const int tid = threadIdx.x;
    if (SCAN && tid < 16) l10[tid] = pss::L10_PACK[tid];
    const int fl = tid / T;   // frame slot inside the workgroup
    const int t = tid % T;    // thread inside the frame: n1 in stage 1, (k2, m1) in stage 2, rho in stage 3
    double2 *ex = ex_all + (size_t)fl * C::EX;
    // per-thread constants
    double2 tw1[16];
    double w[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)t * k2];  // W_N^(n1*k2), n1*k2 < N
#pragma unroll
    for (int n2 = 0; n2 < 16; n2++) w[n2] = SCAN ? 1.0 : win[t + T * n2];
    if (tid < R3 * 16) {
        const int m1 = tid / 16, j2 = tid % 16;
        tw2[C::tw2_slot(tid)] = tw[(size_t)(m1 * j2) * 16];  // W_T^(m1 j2) = W_N^(16 m1 j2)
    }
    __syncthreads();
    const long groups = (n_frames + FPW - 1) / FPW;
    // PREFETCH: the next frame's samples are requested before this frame is transformed and converted when it comes up.
    // Costs ~100 VGPRs (the unsplit kernel is LDS-bound at two wavefronts per SIMD, where 256 are available).
    float2 nx[16];
    auto fetch = [&](long g) {
        const long f = g * FPW + fl;
#ifdef PSS_EXP_SPEC_L2IQ   // timing experiment: every load hits the cache
        const float2 *x = iq + (size_t)(f < n_frames ? (f & 15) : 0) * N;
#else
        const float2 *x = iq + (size_t)(f < n_frames ? f : 0) * N;
#endif
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) nx[n2] = x[t + T * n2];
    };
    if (PREFETCH && (long)blockIdx.x < groups) fetch(blockIdx.x);
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long f = g * FPW + fl;
        const bool valid = f < n_frames;
        double2 v[16];
        if constexpr (!PREFETCH) fetch(g);
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) v[n2] = make_double2((double)nx[n2].x * w[n2], (double)nx[n2].y * w[n2]);
        if (PREFETCH && g + gridDim.x < groups) fetch(g + gridDim.x);
        // Row stores.  PREFETCH kernels: through a buffer resource over the workgroup's FPW rows — frames past the end (and
        // db == NULL) fall outside its size and the hardware drops them, so the 16 stores are UNCONDITIONAL in the instruction
        // stream.  With `if (out)` around them the compiler has to assume at the loop head that the prefetched loads may be the
        // youngest memory operations and waits for vmcnt(0) — i.e. for the acknowledgement of the stores just issued — once per
        // frame; now it waits for the loads only (vmcnt counts in order: the 16 younger stores stay in flight).  1024 / 2048
        // points: +3.5 / +4 % paired; without prefetch the loads ARE the youngest operations and plain pointers are 4-6 % faster.
        float *out = (!PREFETCH && db && valid) ? db + (size_t)f * N * (OB / 4) : nullptr;
        const long f_first = g * FPW;
        const long rows_here = n_frames - f_first < FPW ? n_frames - f_first : FPW;
        const __amdgpu_buffer_rsrc_t ro = make_rsrc((PREFETCH && db) ? db + (size_t)f_first * N * (OB / 4) : nullptr, (PREFETCH && db) ? (unsigned)(rows_here * N * OB) : 0u);
        const int ro_lane = (fl * N + t) * OB;
        float lmax = -INFINITY;
        float dbv[16];
        auto emit = [&](int i, int k, double2 X) {
            if constexpr (D64) {
                const double d64 = db64_of_exact(power_of(X));
                if constexpr (PREFETCH) buf_store_f64(ro, ro_lane, (((k - t) + N / 2) & (N - 1)) * 8, d64);
                else if (out) reinterpret_cast<double *>(out)[(k + N / 2) & (N - 1)] = d64;
                return;
            }
            // compute_fft: float64 all the way, dB rounded to float32; scanner slice: NumPy's complex64 spectrum + float32 chain
            float d;
            if constexpr (SCAN) d = (flags & FLAG_SCAN_EXACT) ? pss::scan_db_np(X.x, X.y, l10) : db_of_fast(power_of(X));
            else d = EXACT ? db_of_exact(power_of(X)) : db_of_fast(power_of(X));
#ifdef PSS_EXP_SPEC_NODB
            d = (float)X.x + (float)X.y;
#endif
            // fftshift; T consecutive bins per store instruction (k - t is a multiple of T, so the row offset is a compile-time constant)
            if constexpr (PREFETCH) buf_store_f32(ro, ro_lane, (((k - t) + N / 2) & (N - 1)) * 4, d);
            else if (out) out[(k + N / 2) & (N - 1)] = d;
            dbv[i] = d;
            lmax = fmaxf(lmax, d);
        };
        constexpr bool WL = T <= 64;  // a frame's threads are lanes of one wavefront: no workgroup barriers in the transform
        if (SPLIT) r16_core_split<LOG_R3, WL>(v, reinterpret_cast<double *>(smem) + (size_t)fl * C::EX, tw1, tw2, t, emit);
        else r16_core<LOG_R3, WL>(v, ex, tw1, tw2, t, emit);
        if (SCAN) {
            // per-frame peak and 20-dB-down bin count (pyspecsdr.py:2546-2552); T threads own one frame
            float m = lmax;
            for (int off = (T < 64 ? T : 64) / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if (T > 64) {
                if ((tid & 63) == 0) red_f[tid >> 6] = m;
                __syncthreads();
                m = red_f[(fl * T) >> 6];
                for (int wv = 1; wv < T / 64; wv++) m = fmaxf(m, red_f[((fl * T) >> 6) + wv]);
            }
            const float thr = m - 20.0f;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) cnt += dbv[i] > thr;
            for (int off = (T < 64 ? T : 64) / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
            if (T > 64) {
                if ((tid & 63) == 0) red_i[tid >> 6] = cnt;
                __syncthreads();
                cnt = 0;
                for (int wv = 0; wv < T / 64; wv++) cnt += red_i[((fl * T) >> 6) + wv];
            }
            if (valid && t == 0) {
                peak[f] = m;
                if (bw) bw[f] = (double)cnt * bin_hz;
                if (count) count[f] = cnt;
            }
        }
        frame_sync<T <= 64 && !SCAN>();  // the next frame overwrites this frame's exchange buffer (SCAN's reductions use barriers)
    }
}

// N = R * 4096, R in {2,4,8,16}: a radix-R decimation-in-frequency pre-pass, then R register-resident 4096-point
// transforms (r16_core<4>) one after the other:
//   X[R k' + r] = FFT_4096( y_r )[k'],  y_r[n] = W_N^(n r) * sum_q x[n + 4096 q] w[n + 4096 q] W_R^(q r).
// One 256-thread workgroup per frame.  The pre-pass reads the frame once (each thread: 16 R-point register DFTs) and
// parks y_r[] in a per-workgroup scratch (N complex float64, L2-resident); bins of one sub-transform land R apart, so
// a dB row is completed by R interleaved store passes that L2 merges.
template <int LOG_R, bool WINDOW>
__global__ __launch_bounds__(256) void k_spectrum_r16_big(const float2 *__restrict__ iq, float *__restrict__ db,
                                                          const double2 *__restrict__ tw, const double *__restrict__ win,
                                                          long n_frames, double2 *__restrict__ scratch, int flags)
{
    using C = Cfg<4>;
    constexpr int T = C::T, NS = C::N, R = 1 << LOG_R, N = R * NS;  // 256 threads, 4096-point sub-transforms
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex = reinterpret_cast<double2 *>(smem);
    double2 *tw2 = ex + C::EX;
    const int t = threadIdx.x;
    double2 *scr = scratch + (size_t)blockIdx.x * N;
    double2 tw1[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)t * k2 * R];  // W_4096^(n1 k2) = W_N^(R n1 k2)
    tw2[C::tw2_slot(t)] = tw[(size_t)((t / 16) * (t % 16)) * 16 * R];     // W_256^(m1 j2) = W_N^(16 R m1 j2)
    __syncthreads();
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
        const float2 *x = iq + (size_t)f * N;
        float *out = db + (size_t)f * N;
        // pre-pass
#pragma unroll 4
        for (int j = 0; j < 16; j++) {
            const int n = t + T * j;
            double2 a[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                float2 s = x[n + NS * q];
                const double wv = WINDOW ? win[n + NS * q] : 1.0;
                a[q] = make_double2((double)s.x * wv, (double)s.y * wv);
            }
            fft_reg<R>(a);
#pragma unroll
            for (int r = 0; r < R; r++) {
                double2 y = a[brev(r, LOG_R)];
                if (r) y = cmul(y, tw[(size_t)n * r]);  // W_N^(n r), n r < N
                scr[(size_t)r * NS + n] = y;
            }
        }
        __syncthreads();
        for (int r = 0; r < R; r++) {
            double2 v[16];
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) v[n2] = scr[(size_t)r * NS + t + T * n2];
            r16_core<4>(v, ex, tw1, tw2, t, [&](int, int kp, double2 X) {
                const int k = R * kp + r;
                out[(k + N / 2) & (N - 1)] = db_of(power_of(X), flags);
            });
            __syncthreads();
        }
    }
}

// The N = R * 4096 transform (R = 8, 16: N = 32768, 65536) around caller-supplied load / store functors, complex float64
// in and out: load(f, idx) -> element idx of frame f, store(f, k, X) receives bin k.  Same scheme as k_spectrum_r16_big
// (radix-R pre-pass into a per-workgroup scratch, then R register-resident 4096-point transforms).
template <int LOG_R, class Load, class Store>
__global__ __launch_bounds__(256) void k_big_g(Load load, Store store, const double2 *__restrict__ tw, long n_frames,
                                               double2 *__restrict__ scratch)
{
    using C = Cfg<4>;
    constexpr int T = C::T, NS = C::N, R = 1 << LOG_R, N = R * NS;
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex = reinterpret_cast<double2 *>(smem);
    double2 *tw2 = ex + C::EX;
    const int t = threadIdx.x;
    double2 *scr = scratch + (size_t)blockIdx.x * N;
    double2 tw1[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)t * k2 * R];
    tw2[C::tw2_slot(t)] = tw[(size_t)((t / 16) * (t % 16)) * 16 * R];
    __syncthreads();
    for (long f = blockIdx.x; f < n_frames; f += gridDim.x) {
#pragma unroll 2
        for (int j = 0; j < 16; j++) {
            const int n = t + T * j;
            double2 a[R];
#pragma unroll
            for (int q = 0; q < R; q++) a[q] = load(f, (size_t)(n + NS * q));
            fft_reg<R>(a);
#pragma unroll
            for (int r = 0; r < R; r++) {
                double2 y = a[brev(r, LOG_R)];
                if (r) y = cmul(y, tw[(size_t)n * r]);
                scr[(size_t)r * NS + n] = y;
            }
        }
        __syncthreads();
        for (int r = 0; r < R; r++) {
            double2 v[16];
#pragma unroll
            for (int n2 = 0; n2 < 16; n2++) v[n2] = scr[(size_t)r * NS + t + T * n2];
            r16_core<4>(v, ex, tw1, tw2, t, [&](int, int kp, double2 X) { store(f, (size_t)(R * kp + r), X); });
            __syncthreads();
        }
    }
}

// N = 256 * NS, NS in {512, 1024, 2048, 4096} (N = 131072 ... 1048576, the reference's largest read buffers,
// pyspecsdr.py:2236 with SAMPLES = 9..12): two kernels around a float64 scratch Y[frame][r][n]:
//   pass 1  for every column n < NS: 256-point transform over q of x[n + NS q] w[n + NS q], times W_N^(n r)  -> Y[r][n]
//   pass 2  for every r < 256: NS-point transform of the contiguous row Y[r][.]  -> bin 256 k' + r, dB, fftshift.
// Pass 1 maps 16 adjacent columns to adjacent LANES (thread = 16 * t + column), so loads are 128-byte and stores
// 256-byte contiguous although each transform walks the frame with stride NS; the per-transform LDS regions are offset
// by one element so that the 16 columns hit different banks.
__global__ __launch_bounds__(256) void k_huge_p1(const float2 *__restrict__ iq, const double2 *__restrict__ tw,
                                                 const double *__restrict__ win, double2 *__restrict__ Y, int NS, long n_frames)
{
    using C = Cfg<0>;  // 256-point transforms, 16 threads each
    constexpr int EXP = C::EX + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex_all = reinterpret_cast<double2 *>(smem);
    const int tid = threadIdx.x, fl = tid & 15, t = tid >> 4;
    const size_t N = (size_t)256 * NS;
    const int cpf = NS / 16;  // column groups per frame
    double2 tw1[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)(t * k2) * NS];  // W_256^(t k2) = W_N^(NS t k2)
    double2 *ex = ex_all + (size_t)fl * EXP;
    const long total = n_frames * cpf;
    for (long g = blockIdx.x; g < total; g += gridDim.x) {
        const long f = g / cpf;
        const int col = (int)(g - f * cpf) * 16 + fl;
        const float2 *x = iq + (size_t)f * N;
        double2 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const size_t idx = (size_t)col + (size_t)NS * (t + 16 * j);
            const float2 sv = x[idx];
            const double wv = win[idx];
            v[j] = make_double2((double)sv.x * wv, (double)sv.y * wv);
        }
        double2 *Yf = Y + (size_t)f * N;
        r16_core<0>(v, ex, tw1, nullptr, t, [&](int, int r, double2 X) {
            if (r) X = cmul(X, tw[(size_t)col * r]);  // W_N^(n r), n r < N
            Yf[(size_t)r * NS + col] = X;
        });
        __syncthreads();
    }
}

template <int LOG_R3>
__global__ __launch_bounds__(256) void k_huge_p2(const double2 *__restrict__ Y, float *__restrict__ db,
                                                 const double2 *__restrict__ tw, long n_rows, int flags)
{
    using C = Cfg<LOG_R3>;
    constexpr int R3 = C::R3, T = C::T, NS = C::N, FPW = C::FPW;
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex_all = reinterpret_cast<double2 *>(smem);
    double2 *tw2 = ex_all + (size_t)FPW * C::EX;
    const int tid = threadIdx.x, fl = tid / T, t = tid % T;
    double2 *ex = ex_all + (size_t)fl * C::EX;
    const size_t N = (size_t)256 * NS;
    double2 tw1[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)(t * k2) * 256];  // W_NS^(t k2) = W_N^(256 t k2)
    if (tid < R3 * 16) {
        const int m1 = tid / 16, j2 = tid % 16;
        tw2[C::tw2_slot(tid)] = tw[(size_t)(m1 * j2) * 16 * 256];
    }
    __syncthreads();
    const long groups = (n_rows + FPW - 1) / FPW;  // n_rows = n_frames * 256, a multiple of FPW
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long row = g * FPW + fl;             // row = f * 256 + r
        const long f = row >> 8;
        const int r = (int)(row & 255);
        const double2 *y = Y + (size_t)row * NS;
        double2 v[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) v[n2] = y[t + T * n2];
        float *out = db + (size_t)f * N;
        r16_core<LOG_R3>(v, ex, tw1, tw2, t, [&](int, int kp, double2 X) {
            const size_t k = (size_t)256 * kp + r;
            out[(k + N / 2) & (N - 1)] = db_of(power_of(X), flags);
        });
        __syncthreads();
    }
}

// ---- the same two-pass N = 256 * NS float64 transform around caller-supplied load / store functors -----------------
// (Bluestein's algorithm for frame lengths that are not a power of two: pss_fft.hip).  load(f, idx) -> element idx of
// frame f's length-N input; store(f, k, X) receives bin k.
template <class Load>
__global__ __launch_bounds__(256) void k_huge_p1_g(Load load, const double2 *__restrict__ tw, double2 *__restrict__ Y, int NS,
                                                   long n_frames)
{
    using C = Cfg<0>;
    constexpr int EXP = C::EX + 1;
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex_all = reinterpret_cast<double2 *>(smem);
    const int tid = threadIdx.x, fl = tid & 15, t = tid >> 4;
    const size_t N = (size_t)256 * NS;
    const int cpf = NS / 16;
    double2 tw1[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)(t * k2) * NS];
    double2 *ex = ex_all + (size_t)fl * EXP;
    const long total = n_frames * cpf;
    for (long g = blockIdx.x; g < total; g += gridDim.x) {
        const long f = g / cpf;
        const int col = (int)(g - f * cpf) * 16 + fl;
        double2 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = load(f, (size_t)col + (size_t)NS * (t + 16 * j));
        double2 *Yf = Y + (size_t)f * N;
        r16_core<0>(v, ex, tw1, nullptr, t, [&](int, int r, double2 X) {
            if (r) X = cmul(X, tw[(size_t)col * r]);
            Yf[(size_t)r * NS + col] = X;
        });
        __syncthreads();
    }
}

template <int LOG_R3, class Store>
__global__ __launch_bounds__(256) void k_huge_p2_g(const double2 *__restrict__ Y, Store store, const double2 *__restrict__ tw,
                                                   long n_rows)
{
    using C = Cfg<LOG_R3>;
    constexpr int R3 = C::R3, T = C::T, NS = C::N, FPW = C::FPW;
    extern __shared__ __align__(16) unsigned char smem[];
    double2 *ex_all = reinterpret_cast<double2 *>(smem);
    double2 *tw2 = ex_all + (size_t)FPW * C::EX;
    const int tid = threadIdx.x, fl = tid / T, t = tid % T;
    double2 *ex = ex_all + (size_t)fl * C::EX;
    double2 tw1[16];
    tw1[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int k2 = 1; k2 < 16; k2++) tw1[k2] = tw[(size_t)(t * k2) * 256];
    if (tid < R3 * 16) {
        const int m1 = tid / 16, j2 = tid % 16;
        tw2[C::tw2_slot(tid)] = tw[(size_t)(m1 * j2) * 16 * 256];
    }
    __syncthreads();
    const long groups = (n_rows + FPW - 1) / FPW;
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long row = g * FPW + fl;
        const long f = row >> 8;
        const int r = (int)(row & 255);
        const double2 *y = Y + (size_t)row * NS;
        double2 v[16];
#pragma unroll
        for (int n2 = 0; n2 < 16; n2++) v[n2] = y[t + T * n2];
        r16_core<LOG_R3>(v, ex, tw1, tw2, t, [&](int, int kp, double2 X) { store(f, (size_t)256 * kp + r, X); });
        __syncthreads();
    }
}

}  // namespace pss_r16
