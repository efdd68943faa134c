// pss_wfm_fused.h — fused WFM forward kernel (included by pss_demod.hip; same -ffp-contract=off rules).
//
// demodulate_wfm (signal_processing.py:119-176) up to and including the FORWARD half of the zero-phase decimator is
// causal, so one lane walks one frame once: discriminator -> {LP 15k, BP pilot -> 1-pole -> sign, BP 23..53k} ->
// x(2*pilot) -> LP 15k -> L/R matrix -> de-emphasis -> forward pass of sosfiltfilt(cheby1) over the odd-extended
// channel.  One wavefront per tile of 64 frames (lane = frame); the forward-filtered channels go straight to HBM in
// the transposed layout the backward kernel reads (Y[2*tile + channel][position][lane], 512-byte rows), so the
// de-emphasised audio u[] never exists in memory.
//  * input: every 16 steps the wave loads the next 16 samples of its 64 frames with coalesced 8-byte loads (16 lanes
//    per frame) and transposes them through LDS, instead of each lane walking its own row;
//  * SciPy's odd extension needs u[1..27] before the first filter step and u[M-28..M-2] after the last: the last 28
//    outputs of both channels live in an LDS ring ([slot][channel][lane], conflict-free), the decimator is primed
//    from it at step 27 and flushed from it after the last step;
//  * ~50 filter coefficients are live in the loop: the Butterworth sets stay scalar operands, the decimator set is
//    pinned in VGPRs (all-scalar spilled the 102-entry SGPR file, all-vector pushed filter state into AGPRs).
#pragma once

namespace wfmf {

using namespace pss;

constexpr int CH = 16;                 // input samples staged per chunk
constexpr int XSTR = TILE + 1;         // float2 row stride of the transposed input chunk
constexpr int RING = EDGE + 1;         // 28 most recent outputs per channel
constexpr size_t LDS_BYTES = (size_t)CH * XSTR * sizeof(float2) + (size_t)RING * 2 * TILE * sizeof(double) + 64 * sizeof(uint2);

__device__ __forceinline__ double vreg(double v)
{
    asm volatile("" : "+v"(v));  // pin the (wave-uniform) value in a vector register
    return v;
}
struct BqV { double b0, b1, b2, a1, a2; };
__device__ __forceinline__ BqV to_v(const Biquad &c) { return BqV{vreg(c.b0), vreg(c.b1), vreg(c.b2), vreg(c.a1), vreg(c.a2)}; }

// sosfilt's DF2T step (scipy _sosfilt.pyx) with the exact shortcuts of biquad_num (pss_demod.hip)
template <int K>
__device__ __forceinline__ double bq(const BqV &c, double x, double &z0, double &z1)
{
    if (K == NUM_GEN) {
        const double xn = __dadd_rn(__dmul_rn(c.b0, x), z0);
        z0 = __dadd_rn(__dsub_rn(__dmul_rn(c.b1, x), __dmul_rn(c.a1, xn)), z1);
        z1 = __dsub_rn(__dmul_rn(c.b2, x), __dmul_rn(c.a2, xn));
        return xn;
    }
    const double xn = __dadd_rn(x, z0);
    const double t = __dmul_rn(c.a1, xn), u = __dmul_rn(c.a2, xn);
    double v, w;
    if (K == NUM_121) { v = __dadd_rn(x, x); w = x; }
    else if (K == NUM_1M21) { v = -__dadd_rn(x, x); w = x; }
    else if (K == NUM_10M1) { v = __dmul_rn(c.b1, x); w = -x; }
    else { v = x; w = __dmul_rn(c.b2, x); }
    z0 = __dadd_rn(__dsub_rn(v, t), z1);
    z1 = __dsub_rn(w, u);
    return xn;
}

// SPEC: the Butterworth rows have SciPy's usual numerator shapes; B121: decimator sections 1..3 are [1, 2, 1].
// CORR: `iq` holds the frames as read and `scal` their iq_correction scalars (k_iqcorr's pre-pass, [n_frames][8]): every sample is corrected
// (signal_processing.py:55-80, iqc_apply: the bits of pss_iq_correction's output) as it leaves the transposed chunk — the corrected copy of
// the batch (8 bytes per sample written and read back) never exists.
template <bool SPEC, bool B121, bool CORR = false>
__global__ __launch_bounds__(TILE) void k_wfm_fwd(const float2 *__restrict__ iq, double *__restrict__ Y, int n,
                                                  long n_frames, int swapped, WfmCoef wc, NfmCoef dc, const float *__restrict__ scal)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);                                   // [CH][XSTR]
    double *ring = reinterpret_cast<double *>(smem + (size_t)CH * XSTR * sizeof(float2));  // [RING][2][TILE]
    uint2 *ltab = reinterpret_cast<uint2 *>(ring + (size_t)RING * 2 * TILE);          // the discriminator's reciprocal table (rcp14f): in every step's chain
    const int lane = threadIdx.x;
    ltab[lane] = pss::RCP14_AB[lane];   // one wavefront per workgroup: visible to itself in program order
    const long tile = blockIdx.x;
    const long f0 = tile * TILE;
    IqcScal kc{1.0f, 1.0f, 0.0f, 1.0f, 1.0f};
    if constexpr (CORR) {
        const long fr = f0 + lane < n_frames ? f0 + lane : n_frames - 1;
        const float4 a = *reinterpret_cast<const float4 *>(scal + (size_t)fr * 8);
        kc = IqcScal{a.x, a.y, a.z, a.w, scal[(size_t)fr * 8 + 4]};
    }
    const int M = n - 1;
    const long L = (long)M + 2 * EDGE;
    double *YL = Y + (size_t)(2 * tile) * L * TILE + lane, *YR = YL + (size_t)L * TILE;

    // ~50 coefficients are live in the loop: the three Butterworth sets (40 doubles) stay scalar (SGPR operands), the
    // decimator set is pinned in VGPRs — all in SGPRs spilled the scalar file, all in VGPRs pushed state into AGPRs
    BqV lp[3], pil[5], lmr[5], dec[4];
    auto as_is = [](const Biquad &c) { return BqV{c.b0, c.b1, c.b2, c.a1, c.a2}; };
#pragma unroll
    for (int s = 0; s < 3; s++) lp[s] = as_is(wc.lp[s]);
#pragma unroll
    for (int s = 0; s < 5; s++) { pil[s] = as_is(wc.pil[s]); lmr[s] = as_is(wc.lmr[s]); }
#pragma unroll
    for (int s = 0; s < 4; s++) dec[s] = to_v(dc.s[s]);
    const double b0d = wc.b0d, a1d = wc.a1d;

    double zlp[6] = {0, 0, 0, 0, 0, 0}, zl2[6] = {0, 0, 0, 0, 0, 0};
    double zpi[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, zlm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double zp1 = 0.0, zdl = 0.0, zdr = 0.0;
    double zfl[8], zfr[8];  // decimator forward state, left / right
    const double SIN_PI = 0x1.1a62633145c07p-53;  // np.sin(np.pi)

    auto lp3 = [&](double x, double *z) {
        x = bq<NUM_GEN>(lp[0], x, z[0], z[1]);
        x = bq<SPEC ? NUM_121 : NUM_GEN>(lp[1], x, z[2], z[3]);
        return bq<SPEC ? NUM_110 : NUM_GEN>(lp[2], x, z[4], z[5]);
    };
    auto bp5 = [&](const BqV *c, double x, double *z) {
        x = bq<NUM_GEN>(c[0], x, z[0], z[1]);
        x = bq<SPEC ? NUM_121 : NUM_GEN>(c[1], x, z[2], z[3]);
        x = bq<SPEC ? NUM_10M1 : NUM_GEN>(c[2], x, z[4], z[5]);
        x = bq<SPEC ? NUM_1M21 : NUM_GEN>(c[3], x, z[6], z[7]);
        return bq<SPEC ? NUM_1M21 : NUM_GEN>(c[4], x, z[8], z[9]);
    };
    auto dec4 = [&](double x, double *z) {
        x = bq<NUM_GEN>(dec[0], x, z[0], z[1]);
        x = bq<B121 ? NUM_121 : NUM_GEN>(dec[1], x, z[2], z[3]);
        x = bq<B121 ? NUM_121 : NUM_GEN>(dec[2], x, z[4], z[5]);
        return bq<B121 ? NUM_121 : NUM_GEN>(dec[3], x, z[6], z[7]);
    };
    long pos = 0;  // next position of the odd-extended sequence
    auto feed = [&](double ul, double ur) {
        YL[(size_t)pos * TILE] = dec4(ul, zfl);
        YR[(size_t)pos * TILE] = dec4(ur, zfr);
        pos++;
    };
#define RG(k, ch) ring[((size_t)((k) % RING) * 2 + (ch)) * TILE + lane]
    // prime the decimator once u[0..27] are in the ring: z = zi * ext[0]; ext[p] = 2 u[0] - u[27 - p], p = 0..26, then
    // u[0..27] themselves (scipy _arraytools.odd_ext, sosfiltfilt)
    auto prime = [&]() {
        const double l0 = RG(0, 0), r0 = RG(0, 1);
        const double el = __dsub_rn(__dmul_rn(2.0, l0), RG(EDGE, 0)), er = __dsub_rn(__dmul_rn(2.0, r0), RG(EDGE, 1));
#pragma unroll
        for (int k = 0; k < 8; k++) { const double zi = vreg(dc.zi[k]); zfl[k] = __dmul_rn(zi, el); zfr[k] = __dmul_rn(zi, er); }
#pragma unroll 1
        for (int pp = 0; pp < EDGE; pp++)
            feed(__dsub_rn(__dmul_rn(2.0, l0), RG(EDGE - pp, 0)), __dsub_rn(__dmul_rn(2.0, r0), RG(EDGE - pp, 1)));
#pragma unroll 1
        for (int k = 0; k <= EDGE; k++) feed(RG(k, 0), RG(k, 1));
    };

    // ---- input staging: lane -> (frame lane/16 + 4 j, sample lane%16), j = 0..15.  Chunks are ALIGNED groups of 16 samples
    // x[16 c .. 16 c + 15] (one 128-byte line per frame and chunk; the off-by-one window x[1 + 16 c ...] straddled two lines
    // and doubled the HBM fetch, rocprofv3 FETCH_SIZE).  Element e = 0 only seeds `prev`; element e >= 1 is step i = e - 1.
    const int ls = lane & 15, lf = lane >> 4;
    auto gload = [&](float2 (&b)[CH], int e0) {
#pragma unroll
        for (int j = 0; j < CH; j++) {
            long fr = f0 + lf + 4 * j;
            fr = fr < n_frames ? fr : n_frames - 1;
            const int e = e0 + ls;
            b[j] = e < n ? iq[(size_t)fr * n + e] : make_float2(0.0f, 0.0f);
        }
    };
    auto spill = [&](const float2 (&b)[CH]) {
#pragma unroll
        for (int j = 0; j < CH; j++) xs[ls * XSTR + lf + 4 * j] = b[j];
    };
    float2 prev = make_float2(0.0f, 0.0f);
    float2 nb[CH];
    gload(nb, 0);
    for (int e0 = 0; e0 < n; e0 += CH) {
        spill(nb);
        if (e0 + CH < n) gload(nb, e0 + CH);
        fused::lds_barrier();  // the transposed chunk is in LDS (LDS-only wait: the Y row stores stay in flight)
        const int cnt = (n - e0) < CH ? (n - e0) : CH;
        for (int t4 = 0; t4 < cnt; t4 += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t4 + u, i = e0 + t - 1;
                if (t < cnt) {
                    float2 cur = xs[t * XSTR + lane];
                    if constexpr (CORR) cur = iqc_apply(cur, kc);
                    if (i < 0) { prev = cur; continue; }
                    const double d = (double)disc_sample(cur, prev, 1.0f, swapped != 0, ltab);  // :122 (x1.0f is exact)
                    prev = cur;
                    const double a = lp3(d, zlp);                                                              // :126
                    const double p = bp5(pil, d, zpi);                                                         // :129
                    double m = bp5(lmr, d, zlm);                                                               // :133
                    const double y = lfilter1(1.0, -0.99, p, zp1);                                             // :130
                    const double pl = (y != y) ? y : ((y < 0.0 || (y == 0.0 && __builtin_signbit(y))) ? SIN_PI : 0.0);
                    m = __dmul_rn(m, __dmul_rn(2.0, pl));                                                      // :134
                    m = lp3(m, zl2);                                                                           // :137
                    const double l = __dmul_rn(__dadd_rn(a, m), 0.5), r = __dmul_rn(__dsub_rn(a, m), 0.5);     // :140-141
                    const double yl = lfilter1(b0d, a1d, l, zdl), yr = lfilter1(b0d, a1d, r, zdr);             // :148-149
                    RG(i, 0) = yl;
                    RG(i, 1) = yr;
                    if (i > EDGE) feed(yl, yr);
                    else if (i == EDGE) prime();  // steps 0..27 are in the ring
                }
            }
        }
        fused::lds_barrier();  // all lanes are done with xs before the next chunk overwrites it
    }
    // right extension: 2 u[M-1] - u[M-2-k], k = 0..26
    {
        const double lN = RG(M - 1, 0), rN = RG(M - 1, 1);
#pragma unroll 1
        for (int k = 0; k < EDGE; k++)
            feed(__dsub_rn(__dmul_rn(2.0, lN), RG(M - 2 - k, 0)), __dsub_rn(__dmul_rn(2.0, rN), RG(M - 2 - k, 1)));
    }
#undef RG
}

}  // namespace wfmf
