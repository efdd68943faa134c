// pss_ubench.h — role micro-benchmarks of the fused NFM forward kernel (compiled only with -DPSS_UBENCH: tools/build_variant.py ubench
// -DPSS_UBENCH; driven by tools/ubench_fwd.py).  Every wavefront of a k_nfm_fwd-shaped launch (1024 workgroups x 256 threads, the
// kernel's dynamic LDS: four workgroups per CU, four wavefronts per SIMD) runs ONE role of that kernel on synthetic on-chip data, with
// no barriers and no HBM traffic: what does each role cost when nothing but its own instruction stream limits it?
#pragma once

namespace fused {

// MODE 0: FIR worker (wave % 3 = the third J of the chunk): for_halves<J> over the LDS window, outputs to the u chunk buffer.
// MODE 1: IIR wavefront: 24 u values from LDS, 24 steps of the four-section pipeline, y rows to 16 recycled (L2-resident) rows.
// MODE 2: discriminator: 8 samples (9 complex loads from the frame's first 32 samples: cache hits) -> window block.
// MODE 3: roles as in the kernel (wave 0: IIR, waves 1..3: FIR + discriminator), still without barriers / HBM.
template <int MODE>
__global__ __launch_bounds__(WG, 1 + NFW) void k_ub_role(const float2 *__restrict__ iq, double *__restrict__ Y, int n, NfmCoef c, float kscale,
                                                         TapsArg taps, int chunks, const double *__restrict__ d_rev)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *win = reinterpret_cast<float *>(smem);
    double *ubuf = reinterpret_cast<double *>(smem + (size_t)TILE * WSTR * sizeof(float));
    uint2 *ltab = reinterpret_cast<uint2 *>(ubuf + (size_t)TILE * FC);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long tile = blockIdx.x;
    if (tid < 64) ltab[tid] = RCP14_AB[tid];
    for (int idx = tid; idx < TILE * WSTR; idx += WG) win[idx] = 1e-3f * (float)((idx * 2654435761u) >> 20);
    for (int idx = tid; idx < TILE * FC; idx += WG) ubuf[idx] = 1e-3 * (double)(idx & 255);
    __syncthreads();
    const float2 *xq = iq + (size_t)(tile * TILE + lane) * n;
    double *Yt = Y + (size_t)tile * 16 * TILE + lane;
    const bool iir_role = MODE == 1 || (MODE == 3 && wave == 0);
    if (iir_role) {
        Iir4 st;
#pragma unroll
        for (int i = 0; i < 8; i++) st.z[i] = c.zi[i];
        st.p0 = st.p1 = st.p2 = 0.0;
        long p = 0;
        double reg[FC];
        for (int ch = 0; ch < chunks; ch++) {
#pragma unroll
            for (int t = 0; t < FC; t++) reg[t] = ubuf[t * TILE + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int t = 0; t < FC; t++) { double v = pipe_step<true>(c, st, reg[t]); Yt[(size_t)((p++) & 15) * TILE] = v; }
        }
    } else {
        const int J = MODE == 3 ? wave - 1 : (MODE == 4 ? 0 : wave % NFW);   // MODE 4: every wavefront runs the SAME code (J = 0)
        float *row = win + lane * WSTR;
        int rot = 0;
        for (int ch = 0; ch < chunks; ch++) {
            if (MODE == 0 || MODE == 3 || MODE == 4) {
                int q = 8 + OPT * J + FC * rot;
                q = q >= WCOLS ? q - WCOLS : q;
#pragma unroll 1
                for (int h = 0; h < OPT / FBH; h++) {
                    int qh = q + FBH * h;
                    qh = qh >= WCOLS ? qh - WCOLS : qh;
                    double *us = ubuf + (OPT * J + FBH * h) * TILE + lane;
                    fir_ring_asm((unsigned)(uintptr_t)row, (unsigned)(uintptr_t)us, qh, d_rev);
                }
            }
            if (MODE == 2 || MODE == 3) {
                float2 pre[OPT + 1];
                float dn[OPT];
                const int tn = (ch * FC + OPT * J) & 31;
#pragma unroll
                for (int e = 0; e <= OPT; e++) pre[e] = xq[tn + e];
#pragma unroll
                for (int e = 0; e < OPT; e++) dn[e] = disc_sample(pre[e + 1], pre[e], kscale, false, ltab);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float4 *nb = reinterpret_cast<float4 *>(row + rot * FC + OPT * J);
#pragma unroll
                for (int e = 0; e < OPT / 4; e++) nb[e] = make_float4(dn[4 * e], dn[4 * e + 1], dn[4 * e + 2], dn[4 * e + 3]);
            }
            rot = (rot + 1) & (NB - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

}  // namespace fused

extern "C" int pss_ubench_role(pss_ctx *ctx, int mode, const float *d_iq, double *d_y, int n, double fs, int chunks, int reps, float *ms)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    PssNfmFilt *flt;
    int r = nfm_filters(ctx, fs, &flt);
    if (r) return r;
    const TapsArg targ = make_taps(flt->taps);
    NfmCoef c;
    for (int s = 0; s < 4; s++) {
        const double *row = flt->sos + 6 * s;
        c.s[s] = Biquad{row[0], row[1], row[2], row[4], row[5]};
    }
    for (int i = 0; i < 8; i++) c.zi[i] = flt->zi[i];
    const float kscale = (float)(fs / (2.0 * M_PI));
    const double *d_rev = nullptr;
    r = nfm_dev_taps(ctx, flt, &d_rev);
    if (r) return r;
    hipEvent_t e0, e1;
    PSS_HIP(ctx, hipEventCreate(&e0));
    PSS_HIP(ctx, hipEventCreate(&e1));
    auto go = [&](auto kern) {
        for (int i = 0; i < reps + 1; i++) {
            if (i == 1) hipEventRecord(e0, ctx->stream);
            hipLaunchKernelGGL(kern, dim3(1024), dim3(fused::WG), fused::LDS_BYTES, ctx->stream, reinterpret_cast<const float2 *>(d_iq), d_y, n,
                               c, kscale, targ, chunks, d_rev);
        }
        hipEventRecord(e1, ctx->stream);
    };
    if (mode == 0) go(fused::k_ub_role<0>);
    else if (mode == 1) go(fused::k_ub_role<1>);
    else if (mode == 2) go(fused::k_ub_role<2>);
    else if (mode == 4) go(fused::k_ub_role<4>);
    else go(fused::k_ub_role<3>);
    PSS_HIP(ctx, hipEventSynchronize(e1));
    PSS_HIP(ctx, hipEventElapsedTime(ms, e0, e1));
    *ms /= reps;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return pss_hip_check(ctx, hipGetLastError(), "ubench launch");
}

// start / end stamps of the most recent k_nfm_fwd launches' workgroups: out[4 i] = start, [4 i + 1] = end (100 MHz ticks), [4 i + 2] = HW_ID
extern "C" int pss_ubench_stamps(pss_ctx *ctx, unsigned long long *out, int n_wg, int clear)
{
    if (!ctx || n_wg > 4096) return PSS_E_ARG;
    PSS_GUARD(ctx);
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (out) PSS_HIP(ctx, hipMemcpyFromSymbol(out, HIP_SYMBOL(pss_dbg_stamps), sizeof(unsigned long long) * 4 * n_wg));
    if (clear) {
        static unsigned long long zeros[4 * 4096];
        PSS_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(pss_dbg_stamps), zeros, sizeof(zeros)));
    }
    return PSS_OK;
}
