// pss_nfm_fused.h — fused NFM kernel: discriminator + FIR + zero-phase Chebyshev decimator + normalise + int16
// for one tile of 64 frames per 256-thread workgroup (included by pss_demod.hip; same -ffp-contract=off rules).
//
// Why fused: with separate kernels the FIR output u[] (8.6 KB/frame float64) is written and read back, which —
// together with the forward-IIR output y[] that the backward pass needs — made the demodulator traffic-bound.
// Here u[] never leaves the chip:
//   wave 0      IIR wave, lane = frame (the recurrence is serial in time): consumes u chunk by chunk from LDS,
//               runs the 4-section skewed pipeline, streams y_fwd to HBM (transposed, 512-byte rows), then runs
//               the backward pass alone and emits the decimated, normalised int16 / float64 audio.
//   waves 1..3  FIR workers: thread (frame f, third j) produces 8 consecutive FIR outputs per chunk of 24 in the
//               exact OpenBLAS-ddot tree (taps as SGPR operands, 16 at a time), from a float32 discriminator
//               window in LDS (96 columns = 4 rotating blocks of 24, so nothing is ever shifted), and computes the
//               discriminator of the NEXT chunk's 8 samples from IQ prefetched before the FIR.
// Per chunk: two workgroup barriers.  LDS: window 64 x 97 x 4 B + one u chunk 64 x 24 x 8 B = 37 KB -> 4 workgroups
// (= all 4 tiles of a CU at BASELINE cfg 2) resident per CU, 4 waves per SIMD.
// The first 64 FIR outputs (windows shorter than 65 taps) and SciPy's odd extension are irregular: the prologue
// computes u[0..63] with the predicated per-lane ddot and parks them in a small global scratch (L2-resident), the
// workers park the last 28 outputs likewise, and the IIR wave builds the extension from those.
#pragma once

namespace fused {

using namespace pss;

constexpr int FC = 24;         // time steps per chunk
constexpr int NB = 4;          // window blocks
constexpr int WCOLS = NB * FC; // 96 window columns: logical column l <-> time 24 c - 8 + l at chunk c
constexpr int WSTR = 97;       // float row stride (odd: conflict-free for lane = frame at a fixed column)
constexpr int HEAD = 64;       // outputs produced by the prologue
constexpr int NFW = 3;         // FIR worker waves (measured slower: 4 x 6 outputs in 5-wave workgroups 0.89 vs 0.67 ms; 2 x 12 outputs in
                               // 3-wave workgroups with 168 VGPRs each 0.88 vs 0.60 ms)
constexpr int OPT = FC / NFW;  // FIR outputs per worker thread and chunk
constexpr int WG = 64 * (1 + NFW);
constexpr int HTAPS = 80;     // LDS copy of the reversed taps for the left-edge outputs: 65 taps + 15 zero slots (clamp-free tail loads)
constexpr size_t LDS_BYTES = (size_t)TILE * WSTR * sizeof(float) + (size_t)TILE * FC * sizeof(double) + HTAPS * sizeof(double) + 64 * sizeof(uint2);

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence that also waits for every
// outstanding GLOBAL store of the wave (vmcnt(0)); the IIR wave has 24 y_fwd row stores in flight per chunk, and
// draining them twice per chunk serialised the whole pipeline on the store round trip.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

struct Iir4 {
    // skewed 4-section pipeline, see iir4_pass in pss_demod.hip
    double z[8];
    double p0, p1, p2;
};

template <bool B121>
__device__ __forceinline__ double sec_step(const NfmCoef &c, Iir4 &st, int s, double x)
{
    return (B121 && s > 0) ? biquad_step_121(c.s[s], x, st.z[2 * s], st.z[2 * s + 1])
                           : biquad_step(c.s[s], x, st.z[2 * s], st.z[2 * s + 1]);
}

template <bool B121>
__device__ __forceinline__ double pipe_step(const NfmCoef &c, Iir4 &st, double x)
{
    const double xs[4] = {x, st.p0, st.p1, st.p2};
    double m[4], xn[4], t[4], u[4], v[4], w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool one = B121 && k > 0;
        m[k] = one ? xs[k] : __dmul_rn(c.s[k].b0, xs[k]);
        v[k] = one ? __dadd_rn(xs[k], xs[k]) : __dmul_rn(c.s[k].b1, xs[k]);
        w[k] = one ? xs[k] : __dmul_rn(c.s[k].b2, xs[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) xn[k] = __dadd_rn(m[k], st.z[2 * k]);
#pragma unroll
    for (int k = 0; k < 4; k++) { t[k] = __dmul_rn(c.s[k].a1, xn[k]); u[k] = __dmul_rn(c.s[k].a2, xn[k]); }
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] = __dsub_rn(v[k], t[k]); w[k] = __dsub_rn(w[k], u[k]); }
#pragma unroll
    for (int k = 0; k < 4; k++) { st.z[2 * k] = __dadd_rn(v[k], st.z[2 * k + 1]); st.z[2 * k + 1] = w[k]; }
    st.p0 = xn[0]; st.p1 = xn[1]; st.p2 = xn[2];
    return xn[3];
}

// FBH consecutive full-window FIR outputs (outputs 4H .. 4H+3 of the thread's 8) for worker third J.  Window column
// of x(c') (c' = 0..71 relative to the thread's first output) is l = 8 + 8 J + c'; blk[b] points at this frame's
// row inside logical block b.  Four outputs at a time keeps the accumulators at 16 doubles (the kernel must fit
// the 128-VGPR budget of 4 waves/SIMD).
constexpr int FBH = 4;
static_assert(FC % NFW == 0 && OPT % FBH == 0 && OPT <= 8, "chunk / worker split");
template <int J, int H>
__device__ __forceinline__ void fir_batch(const float *const (&blk)[NB], const double *__restrict__ yrev, double (&out)[OPT])
{
    auto x = [&](int cc) {
        const int l = 8 + OPT * J + FBH * H + cc;
        return (double)blk[l / FC][l % FC];
    };
    double s[FBH][4];
    // Four phases of 16 taps; k is a run-time loop index (keeps only 16 taps live in SGPRs, and converted window values
    // from one phase out of the next), but the column arithmetic must stay compile-time, so the cases are spelled out.
    // Phase 0 (which STARTS the sums) runs ahead of the loop: with all four cases in the loop the accumulators were copied
    // (16 v_mov_b64 per pass) where the starting case and the accumulating cases join.
    auto phase = [&](auto KC, int k) {
        constexpr int K = decltype(KC)::value;
        double ya[8], yb[8];
#pragma unroll
        for (int l = 0; l < 8; l++) { ya[l] = yrev[8 * k + l]; yb[l] = yrev[32 + 8 * k + l]; }
#pragma unroll
        for (int o = 0; o < FBH; o++) {
#pragma unroll
            for (int l = 0; l < 4; l++) {
                double lo = __fma_rn(x(8 * K + o + 32 + l), yb[l], __fma_rn(x(8 * K + o + l), ya[l], 0.0));
                double hi = __fma_rn(x(8 * K + o + 36 + l), yb[l + 4], __fma_rn(x(8 * K + o + l + 4), ya[l + 4], 0.0));
                double a = __dadd_rn(lo, hi);
                s[o][l] = (K == 0) ? a : __dadd_rn(s[o][l], a);
            }
        }
    };
    int k0 = 0;
    asm volatile("" : "+s"(k0));  // opaque zero: constant tap indices would be hoisted out of the chunk loop and held in SGPRs for good
    phase(std::integral_constant<int, 0>{}, k0);
#pragma unroll 1
    for (int k = 1; k < 4; k++) {
        if (k == 1) phase(std::integral_constant<int, 1>{}, k);
        else if (k == 2) phase(std::integral_constant<int, 2>{}, k);
        else phase(std::integral_constant<int, 3>{}, k);
    }
    const double y64 = yrev[64];
#pragma unroll
    for (int o = 0; o < FBH; o++) {
        double dot = __dadd_rn(__dadd_rn(s[o][0], s[o][2]), __dadd_rn(s[o][1], s[o][3]));
        out[FBH * H + o] = __fma_rn(y64, x(o + 64), dot);
    }
}

// The ddot tree (pss_device.h) for a WAVE-UNIFORM length n = 1..64: the left-edge FIR outputs, lane = frame.  x[j] is the
// lane's window row (float32 discriminator at time j), y[j] the tap that meets it (LDS, same address in every lane).
// One straight-line body per shape of the tree — S32 32-element steps (n >= 32; two only for n = 64) and an optional
// 16-element step — so that the LDS loads of a whole phase are in flight together; the n - n1 <= 15 tail elements are
// loaded ahead of their dependent FMA chain.  (The rolled ddot_skx_uniform this replaces paid one exposed LDS round trip
// per element: 0.045 ms of the kernel's 0.69.)
template <int S32, bool H16>
__device__ __forceinline__ double ddot_head_body(const float *__restrict__ x, const double *__restrict__ y)
{
    double s[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        double a[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            double v = 0.0;
            if (S32 >= 1) {
                double lo = __fma_rn((double)x[8 * k + l], y[8 * k + l], 0.0);
                double hi = __fma_rn((double)x[8 * k + l + 4], y[8 * k + l + 4], 0.0);
                if (S32 == 2) {
                    lo = __fma_rn((double)x[32 + 8 * k + l], y[32 + 8 * k + l], lo);
                    hi = __fma_rn((double)x[32 + 8 * k + l + 4], y[32 + 8 * k + l + 4], hi);
                }
                v = __dadd_rn(lo, hi);
            }
            if (H16) v = __fma_rn((double)x[32 * S32 + 4 * k + l], y[32 * S32 + 4 * k + l], v);
            a[k] = v;
        }
        s[l] = __dadd_rn(__dadd_rn(__dadd_rn(a[0], a[1]), a[2]), a[3]);
    }
    return __dadd_rn(__dadd_rn(s[0], s[2]), __dadd_rn(s[1], s[3]));
}

__device__ __forceinline__ double ddot_head(const float *__restrict__ x, const double *__restrict__ y, int n)
{
    const int n1 = n & -16;
    double dot = 0.0;
    if (n1 == 16) dot = ddot_head_body<0, true>(x, y);
    else if (n1 == 32) dot = ddot_head_body<1, false>(x, y);
    else if (n1 == 48) dot = ddot_head_body<1, true>(x, y);
    else if (n1 == 64) dot = ddot_head_body<2, false>(x, y);
    // tail operands: all loads ahead of the chain (y has 15 zero slots behind the taps, x is a window row: in bounds)
    float xt[15];
    double yt[15];
#pragma unroll
    for (int t = 0; t < 15; t++) { xt[t] = x[n1 + t]; yt[t] = y[n1 + t]; }
    const int cnt = n - n1;
#pragma unroll
    for (int t = 0; t < 15; t++)
        if (t < cnt) dot = __fma_rn(yt[t], (double)xt[t], dot);
    return dot;
}

template <int J, int H = 0, class Put>
__device__ __forceinline__ void for_halves(const float *const (&blk)[NB], const double *__restrict__ yrev, double (&out)[OPT], Put put)
{
    fir_batch<J, H>(blk, yrev, out);
    put(out, H);
    if constexpr ((H + 1) * FBH < OPT) for_halves<J, H + 1>(blk, yrev, out, put);
}

// SWAPPED: operand order of the discriminator's complex product for frames of 32 769 samples and more (disc_sample); a template
// parameter because as a run-time flag both orders were evaluated and selected per sample.
// DISC_IN (opt-in: options "disc_rows" / "disc_spectrum"): the discriminator was evaluated by the producer of `dsc` (k_disc_rows, or
// the 1024-point spectrum kernel, which holds every sample of the frame in registers anyway: k_spectrum_r16<..., DISC>): rows of `ld`
// floats per frame, d[t] at [t], zeros from t = n - 1 to the row's end (ld >= n + 2 FC, a multiple of 4: the fetch runs one chunk
// ahead).  The workers then fetch a chunk as 16-byte pieces, consecutive lanes on consecutive pieces of a frame (6 pieces of a frame,
// then the next frame: ~12 frames per load instruction) and only move them into the window, instead of 72 bytes of IQ per lane at a
// stride of one frame (64 different lines per load instruction) that the discriminator then waits for: 0.45 instead of 0.58-0.62 ms
// at cfg 2.  (With all arithmetic removed the IQ loads alone keep the default kernel at 0.45 ms, 0.17 ms without them.)
template <bool B121, bool SWAPPED = false, bool DISC_IN = false>
__global__ __launch_bounds__(WG, 1 + NFW) void k_nfm_fwd(const float2 *__restrict__ iq, double *__restrict__ Y,
                                                    double *__restrict__ Uh, double *__restrict__ Utl, int n,
                                                    long n_frames, NfmCoef c, float kscale, TapsArg taps,
                                                    const float *__restrict__ dsc = nullptr, int ld = 0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *win = reinterpret_cast<float *>(smem);                                   // [TILE][WSTR]
    double *ubuf = reinterpret_cast<double *>(smem + (size_t)TILE * WSTR * sizeof(float));  // [FC][TILE]
    double *ltaps = ubuf + (size_t)TILE * FC;                                        // reversed taps (left-edge dots)
    uint2 *ltab = reinterpret_cast<uint2 *>(ltaps + HTAPS);                          // the discriminator's reciprocal table (pss_device.h rcp14f)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: everything indexed by the role / chunk stays scalar
    const long tile = blockIdx.x;
    const int M = n - 1;                      // FIR outputs per frame (M >= 128 guaranteed by the caller)
    const long L = (long)M + 2 * EDGE;
    const int NC = (M - HEAD + FC - 1) / FC;  // worker chunks
    const long f = tile * TILE + lane;
    const long fr = f < n_frames ? f : n_frames - 1;  // masked lanes replay the last frame (results dropped)
    const float2 *xq = iq + (size_t)fr * n;
    double *Yt = Y + (size_t)tile * L * TILE + lane;
    double *Uht = Uh + (size_t)tile * HEAD * TILE + lane;
    double *Utt = Utl + (size_t)tile * (EDGE + 1) * TILE + lane;
#define YAT(p) Yt[(size_t)(p) * TILE]
    if (tid >= 128 && tid < 192) ltab[tid - 128] = RCP14_AB[tid - 128];
    if (tid < HTAPS) ltaps[tid] = tid < 65 ? taps.rev[tid] : 0.0;  // ltaps[m] = taps[64 - m]: output i pairs x[j] with ltaps[64 - i + j]
    // ---- prologue: discriminator of times 0..87 into logical columns 8..95 (physical = logical at chunk 0)
    for (int idx = tid; idx < TILE * WCOLS; idx += WG) {
        const int fl = idx / WCOLS, l = idx % WCOLS, t = l - 8;
        const long ff = tile * TILE + fl;
        const float2 *x = iq + (size_t)(ff < n_frames ? ff : n_frames - 1) * n;
        float d = 0.0f;
        if constexpr (DISC_IN) {
            if (t >= 0) d = dsc[(size_t)(ff < n_frames ? ff : n_frames - 1) * ld + t];
        } else {
            if (t >= 0 && t < M) d = disc_sample(x[t + 1], x[t], kscale, SWAPPED);
        }
        win[fl * WSTR + l] = d;
    }
    __syncthreads();
    // The first 64 FIR outputs have windows shorter than 65 samples, i.e. each its own ddot shape.  With lane = frame
    // the length is wave-uniform: the four waves take 16 outputs each (interleaved, so the dot lengths balance) and
    // park them in Uh (L2) for the IIR wave.
    {
        const float *row0 = win + lane * WSTR + 8;  // time t of this lane's frame at row0[t] (chunk-0 layout)
#pragma unroll 1
        for (int i = wave; i < HEAD; i += 1 + NFW) {
            Uht[(size_t)i * TILE] = ddot_head(row0, ltaps + (64 - i), i + 1);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (wave == 0) {
        // =============================== IIR wave ===============================
        Iir4 st;
        long p = 0;  // stream position of the next input; outputs lag by 3
        auto emit = [&](double v) { YAT(p - 3) = v; };
        // odd extension head + u[0..63] from the prologue (scipy odd_ext: ext[p] = 2u[0] - u[27-p])
        const double u0 = Uht[0];
        const double two_u0 = __dmul_rn(2.0, u0);
        auto head = [&](int pp) { return pp < EDGE ? __dsub_rn(two_u0, Uht[(size_t)(EDGE - pp) * TILE]) : Uht[(size_t)(pp - EDGE) * TILE]; };
        {
            const double x0 = head(0);
#pragma unroll
            for (int i = 0; i < 8; i++) st.z[i] = __dmul_rn(c.zi[i], x0);
            st.p0 = sec_step<B121>(c, st, 0, x0);
            { double t1 = sec_step<B121>(c, st, 1, st.p0); st.p0 = sec_step<B121>(c, st, 0, head(1)); st.p1 = t1; }
            { double t2 = sec_step<B121>(c, st, 2, st.p1); double t1 = sec_step<B121>(c, st, 1, st.p0);
              st.p0 = sec_step<B121>(c, st, 0, head(2)); st.p2 = t2; st.p1 = t1; }
            p = 3;
        }
#pragma unroll 4
        for (int pp = 3; pp < EDGE + HEAD; pp++) { double v = pipe_step<B121>(c, st, head(pp)); emit(v); p++; }
        // chunks from the workers
        double reg[FC];
        for (int ch = 0; ch <= NC; ch++) {
            if (ch >= 1) {
                const int cnt = (M - HEAD - (ch - 1) * FC) < FC ? (M - HEAD - (ch - 1) * FC) : FC;
                if (cnt == FC) {
#pragma unroll
                    for (int t = 0; t < FC; t++) { double v = pipe_step<B121>(c, st, reg[t]); emit(v); p++; }
                } else {
#pragma unroll
                    for (int t = 0; t < FC; t++)
                        if (t < cnt) { double v = pipe_step<B121>(c, st, reg[t]); emit(v); p++; }
                }
            }
            if (ch < NC) {
                lds_barrier();  // A: workers finished FIR(ch) -> ubuf
#pragma unroll
                for (int t = 0; t < FC; t++) reg[t] = ubuf[t * TILE + lane];
                lds_barrier();  // B
            } else {
                __syncthreads();  // final A/B: full fences, the workers' global Utt rows must be visible
                __syncthreads();
            }
        }
        // odd extension tail: ext[27+M+k] = 2u[M-1] - u[M-2-k]; Utt[r] = u[M-28+r]
        {
            const double two_uL = __dmul_rn(2.0, Utt[(size_t)EDGE * TILE]);
#pragma unroll 3
            for (int k = 0; k < EDGE; k++) {
                double v = pipe_step<B121>(c, st, __dsub_rn(two_uL, Utt[(size_t)(EDGE - 1 - k) * TILE]));
                emit(v); p++;
            }
        }
        // drain
        { double v = sec_step<B121>(c, st, 3, st.p2); emit(v); p++; st.p2 = sec_step<B121>(c, st, 2, st.p1); st.p1 = sec_step<B121>(c, st, 1, st.p0); }
        { double v = sec_step<B121>(c, st, 3, st.p2); emit(v); p++; st.p2 = sec_step<B121>(c, st, 2, st.p1); }
        { double v = sec_step<B121>(c, st, 3, st.p2); emit(v); p++; }
    } else {
        // =============================== FIR workers ===============================
        const int J = wave - 1;  // third of the chunk handled by this wave (wave-uniform)
        float *row = win + lane * WSTR;
        int rot = 0;  // physical block of logical block 0
        for (int ch = 0; ch <= NC; ch++) {
            if (ch < NC) {
                const int ibase = HEAD + ch * FC + OPT * J;  // first output index of this thread's batch
                const int tn = HEAD + (ch + 1) * FC + OPT * J;  // time of this thread's first new sample of the next chunk
                const float *blk[NB];
#pragma unroll
                for (int b = 0; b < NB; b++) blk[b] = row + ((b + rot) & (NB - 1)) * FC;
                auto put = [&](const double (&out)[OPT], int h) {
#pragma unroll
                    for (int o = FBH * h; o < FBH * h + FBH; o++) {
                        const int i = ibase + o;
                        ubuf[(OPT * J + o) * TILE + lane] = out[o];
                        if (i < M && i >= M - 1 - EDGE) Utt[(size_t)(i - (M - 1 - EDGE)) * TILE] = out[o];
                    }
                };
                {
                    double out[OPT];
                    if (J == 0) { for_halves<0>(blk, taps.rev, out, put); }
                    else if (J == 1) { for_halves<1>(blk, taps.rev, out, put); }
                    else if (J == 2 || NFW == 3) { for_halves<2>(blk, taps.rev, out, put); }
                    else { for_halves<NFW - 1>(blk, taps.rev, out, put); }
                }
                // discriminator of the next chunk's 8 samples (9 complex); kept in registers until the window block
                // is free.  The loads are issued after the FIR so they do not occupy registers across it — the other
                // three waves of the SIMD cover their latency.
                float dn[OPT];
                if constexpr (DISC_IN) {
                    // the next chunk's FC floats of every frame of the tile: TILE * FC / 4 pieces of 16 bytes over the 64 * NFW worker threads.
                    // Buffer addressing: the tile's rows as one resource (scalar), the chunk as the scalar offset, one 32-bit byte offset per
                    // lane — plain pointers became 64-bit per-lane addresses that were hoisted out of the chunk loop and spilled
                    constexpr int PPF = FC / 4, NP = TILE * PPF / (64 * NFW);
                    static_assert(FC % 4 == 0 && (TILE * PPF) % (64 * NFW) == 0 && NP * 4 == OPT, "piece split");
                    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
                    const int wt = tid - 64;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float *>(dsc + (size_t)tile * TILE * ld), 0, (int)(TILE * ld * sizeof(float)), 0x00020000);
                    const int last = (int)((n_frames - 1 - tile * TILE) < TILE - 1 ? (n_frames - 1 - tile * TILE) : TILE - 1);  // masked frames replay the last one
#pragma unroll
                    for (int r = 0; r < NP; r++) {
                        const int item = wt + 64 * NFW * r, fl = item / PPF, pc = item % PPF;
                        const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, ((fl < last ? fl : last) * ld + 4 * pc) * 4, (HEAD + (ch + 1) * FC) * 4, 0);
                        dn[4 * r] = __uint_as_float(v.x); dn[4 * r + 1] = __uint_as_float(v.y);
                        dn[4 * r + 2] = __uint_as_float(v.z); dn[4 * r + 3] = __uint_as_float(v.w);
                    }
                } else {
                    float2 pre[OPT + 1];
                    const int left = M - tn;  // discriminator samples of this batch inside the frame (one scalar per chunk: the bounds n / M themselves live in spilled SGPRs)
#pragma unroll
                    for (int e = 0; e <= OPT; e++) pre[e] = (e <= left) ? xq[tn + e] : make_float2(0.0f, 0.0f);
#pragma unroll
                    for (int e = 0; e < OPT; e++) dn[e] = (e < left) ? disc_sample(pre[e + 1], pre[e], kscale, SWAPPED, ltab) : 0.0f;
                }
                lds_barrier();  // A
                // the oldest block (logical 0) becomes the newest (logical 3 of the next chunk)
                if constexpr (DISC_IN) {
                    constexpr int PPF = FC / 4, NP = TILE * PPF / (64 * NFW);
                    const int wt = tid - 64;
#pragma unroll
                    for (int r = 0; r < NP; r++) {
                        const int item = wt + 64 * NFW * r, fl = item / PPF, pc = item % PPF;
                        float *nb = win + fl * WSTR + rot * FC + 4 * pc;
#pragma unroll
                        for (int k = 0; k < 4; k++) nb[k] = dn[4 * r + k];
                    }
                } else {
                    float *nb = row + rot * FC + OPT * J;
#pragma unroll
                    for (int e = 0; e < OPT; e++) nb[e] = dn[e];
                }
                rot = (rot + 1) & (NB - 1);
                lds_barrier();  // B
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();  // A
                __syncthreads();  // B
            }
        }
    }
#undef YAT
}


// Backward pass of sosfiltfilt over y_fwd (read in reverse from Y[tile][p][lane]), decimation [::q], peak
// normalisation, stereo int16 / float64 audio.  One wavefront per tile, lane = frame.
template <bool B121, bool WFM = false>
__global__ __launch_bounds__(TILE) void k_nfm_bwd(const double *__restrict__ Y, double *__restrict__ A, int n, int q,
                                                  int n_out, long n_frames, NfmCoef c, int16_t *__restrict__ pcm,
                                                  double *__restrict__ audio)
{
    const int lane = threadIdx.x;
    const long tile = blockIdx.x;
    const long f = tile * TILE + lane;
    const int M = n - 1;
    const long L = (long)M + 2 * EDGE;
    const double *Yt = Y + (size_t)tile * L * TILE + lane;
    double *At = A + (size_t)tile * n_out * TILE + lane;
#define YAT(p) Yt[(size_t)(p) * TILE]
    const double ylast = YAT(L - 1);
    double z[8];
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = __dmul_rn(c.zi[i], ylast);
    double mx = 0.0;
    bool nan = false;
    long next = EDGE + (long)(n_out - 1) * q;  // largest kept position (p = 27 + j q)
    int j = n_out - 1;
    iir4_pass<B121, IIR_CH>(c, z, L - EDGE, [&](long r) { return YAT(L - 1 - r); },
                            [&](double (&b)[IIR_CH], long r) {
#pragma unroll
                                for (int t = 0; t < IIR_CH; t++) b[t] = YAT(L - 1 - (r + t));
                            },
                            [&](long r, double v) {
                                if (L - 1 - r == next) {
                                    At[(size_t)j * TILE] = v;
                                    double av = fabs(v);
                                    nan = nan || (av != av);
                                    mx = av > mx ? av : mx;
                                    next -= q;
                                    j--;
                                }
                            });
    if (nan) mx = __builtin_nan("");
    if (WFM) {  // rows are (tile, channel, frame-in-tile): k_wfm_finalize normalises the two channels jointly
        audio[f] = mx;
        return;
    }
    if (f < n_frames) {
        for (int k = 0; k < n_out; k++) {
            double a = __dmul_rn(__ddiv_rn(At[(size_t)k * TILE], mx), 0.95);  // audio / max|audio| * 0.95
            if (audio) audio[(size_t)f * n_out + k] = a;
            if (pcm) {
                uint16_t s = (uint16_t)pcm16(a);
                reinterpret_cast<uint32_t *>(pcm)[(size_t)f * n_out + k] = (uint32_t)s | ((uint32_t)s << 16);
            }
        }
    }
#undef YAT
}

}  // namespace fused
