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
#include "pss_fir_ring_asm.h"

#ifdef PSS_UBENCH   // per-workgroup start / end times of k_nfm_fwd (s_memrealtime, 100 MHz): dispatch skew and tail (tools/ubench_fwd.py stamps)
__device__ unsigned long long pss_dbg_stamps[4 * 4096];
#endif

namespace fused {

using namespace pss;
// Progress balancing groups the workgroups of a launch by the compute unit they are expected to share: consecutive workgroups go to consecutive
// CUs while the grid fits the machine, so workgroup i is taken to run on CU i mod ncu (ncu = hipDeviceProp's multiProcessorCount, a kernel
// argument: 256 on an unpartitioned MI355X).  The grouping only steers s_setprio; a wrong guess costs balance, never results.

constexpr int FC = 24;         // time steps per chunk
constexpr int NB = 4;          // window blocks
constexpr int WCOLS = NB * FC; // 96 window columns: logical column l <-> time 24 c - 8 + l at chunk c
constexpr int WSTR = 100;      // float row stride: rows 16-byte aligned (the FIR reads 4 columns per ds_read_b128) and 25 x 16 bytes apart —
                               // odd in 16-byte units, so the 16 lanes a b128 access services together hit 16 different bank groups
constexpr int HEAD = 64;       // outputs produced by the prologue
constexpr int NFW = 3;         // FIR worker waves (measured slower: 4 x 6 outputs in 5-wave workgroups 0.89 vs 0.67 ms; 2 x 12 outputs in
                               // 3-wave workgroups with 168 VGPRs each 0.88 vs 0.60 ms)
constexpr int OPT = FC / NFW;  // FIR outputs per worker thread and chunk
constexpr int WG = 64 * (1 + NFW);
constexpr size_t LDS_BYTES = (size_t)TILE * WSTR * sizeof(float) + (size_t)TILE * FC * sizeof(double) + 64 * sizeof(uint2) + 16;   // + the workgroup's priority word

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a release fence that also waits for every
// outstanding GLOBAL store of the wave (vmcnt(0)); the IIR wave has 24 y_fwd row stores in flight per chunk, and
// draining them twice per chunk serialised the whole pipeline on the store round trip.
__device__ __forceinline__ void lds_barrier()
{
#ifdef PSS_EXP_NOBAR   // timing experiment only (results are wrong): no workgroup barriers inside the chunk loop
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
__device__ __forceinline__ void lds_barrier_b()
{
#ifdef PSS_EXP_NOBAR_B  // timing experiment only: barrier B dropped
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    lds_barrier();
#endif
}

// Balanced issue priority.  The SIMD's arbiter serves the OLDEST ready wavefront first: of the four workgroups a CU holds (they all
// start within 1.3 us of each other) the first-dispatched one runs at the pace of a workgroup that has the CU to itself and is done after
// 0.32 ms, the others after 0.40 / 0.50 / 0.59 ms (per-workgroup s_memrealtime stamps, tools/ubench_stamps.py) — a staircase whose last
// steps run with three, two, one wavefront per SIMD and hide no latency.  The user priority (s_setprio: ranked above age by the arbiter)
// therefore follows PROGRESS: once per chunk the IIR wavefront publishes the workgroup's chunk index in a per-CU word array (global
// memory, 16 bytes per CU), reads its three neighbours' and takes as priority the number of them that are ahead; the workers pick it
// up from LDS behind barrier A.  All four workgroups of a CU then finish within 2 % of each other (493 .. 502 us) and the launch is
// 4 % shorter than with the priority rotating on the real-time counter (every 41 us: 363 / 418 / 495 / 545 us), 9 % shorter than with
// none (paired A/B, tools/ab_fwd.py).  Workgroup i runs on CU i mod 256 while the grid fits the machine (HW_ID probe,
// tools/ubench/simd_placement_probe.hip); for other grids the words are merely a worse hint.
__device__ __forceinline__ void prio_rotate(int v)
{
    switch (v & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
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

constexpr int FBH = 4;        // outputs per FIR statement (pss_fir_ring_asm.h)
static_assert(FC % NFW == 0 && OPT % FBH == 0 && OPT <= 8, "chunk / worker split");

// The ddot tree (pss_device.h) for a WAVE-UNIFORM length n = 1..64: the left-edge FIR outputs, lane = frame.  x[j] is the
// lane's window row (float32 discriminator at time j), y[j] the tap that meets it (LDS, same address in every lane).
// One straight-line body per shape of the tree — S32 32-element steps (n >= 32; two only for n = 64) and an optional
// 16-element step — so that the LDS loads of a whole phase are in flight together; the n - n1 <= 15 tail elements are
// loaded ahead of their dependent FMA chain.  (The rolled ddot_skx_uniform this replaces paid one exposed LDS round trip
// per element: 0.045 ms of the kernel's 0.69.)
template <int S32, bool H16>
__device__ __forceinline__ double ddot_head_body(const float *__restrict__ row, const double *__restrict__ y)
{
    // the row's first 32 S32 + 16 H16 floats by 16-byte reads (the row stride of 25 x 16 bytes is conflict-free for ds_read_b128 and
    // 4-way conflicting for single floats at lane = frame)
    constexpr int NX = 32 * S32 + (H16 ? 16 : 0);
    float x[NX];
#pragma unroll
    for (int g = 0; g < NX / 4; g++) {
        const float4 v = *reinterpret_cast<const float4 *>(row + 4 * g);
        x[4 * g] = v.x; x[4 * g + 1] = v.y; x[4 * g + 2] = v.z; x[4 * g + 3] = v.w;
    }
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

// row: the lane's window row in LDS (16-byte aligned), x[j] = row[j]
__device__ __forceinline__ double ddot_head(const float *__restrict__ row, const double *__restrict__ y, int n)
{
    const int n1 = n & -16;
    double dot = 0.0;
    if (n1 == 16) dot = ddot_head_body<0, true>(row, y);
    else if (n1 == 32) dot = ddot_head_body<1, false>(row, y);
    else if (n1 == 48) dot = ddot_head_body<1, true>(row, y);
    else if (n1 == 64) dot = ddot_head_body<2, false>(row, y);
    // tail operands: all loads ahead of the chain (y has 15 zero slots behind the taps, x is a window row: in bounds)
    float xt[16];
    double yt[15];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 v = *reinterpret_cast<const float4 *>(row + n1 + 4 * g);
        xt[4 * g] = v.x; xt[4 * g + 1] = v.y; xt[4 * g + 2] = v.z; xt[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < 15; t++) yt[t] = y[n1 + t];
    const int cnt = n - n1;
#pragma unroll
    for (int t = 0; t < 15; t++)
        if (t < cnt) dot = __fma_rn(yt[t], (double)xt[t], dot);
    return dot;
}

// SWAPPED: operand order of the discriminator's complex product for frames of 32 769 samples and more (disc_sample); a template
// parameter because as a run-time flag both orders were evaluated and selected per sample.
// d_rev: the reversed taps in device memory (PssNfmFilt::d_rev; rev[j] = taps[64 - j], 15 zeros behind them): scalar tap loads of the
// FIR statement and of the left-edge dots (output i pairs x[j] with rev[64 - i + j]).
template <bool B121, bool SWAPPED = false>
__global__ __launch_bounds__(WG, 1 + NFW) void k_nfm_fwd(const float2 *__restrict__ iq, double *__restrict__ Y,
                                                    double *__restrict__ Uh, double *__restrict__ Utl, int n,
                                                    long n_frames, NfmCoef c, float kscale, const double *__restrict__ d_rev,
                                                    unsigned *prog, unsigned ncu, unsigned epoch, long tile0)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *win = reinterpret_cast<float *>(smem);                                   // [TILE][WSTR]
    double *ubuf = reinterpret_cast<double *>(smem + (size_t)TILE * WSTR * sizeof(float));  // [FC][TILE]
    uint2 *ltab = reinterpret_cast<uint2 *>(ubuf + (size_t)TILE * FC);
    int *lprio = reinterpret_cast<int *>(ltab + 64);                                // priority of the workgroup, handed from the IIR wavefront to the workers                        // the discriminator's reciprocal table (pss_device.h rcp14f)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // in an SGPR: everything indexed by the role / chunk stays scalar
    const long tile = tile0 + blockIdx.x;   // (tile0: first tile of this grid; the product launches the whole batch as one grid, tile0 = 0)
    const int slot = (int)(blockIdx.x / ncu) & 3;   // which of its CU's (up to) four workgroups this one is: consecutive workgroups go to consecutive CUs
    const int cu = (int)(blockIdx.x % ncu);
    const int M = n - 1;                      // FIR outputs per frame (M >= 128 guaranteed by the caller)
    const long L = (long)M + 2 * EDGE;
    const int NC = (M - HEAD + FC - 1) / FC;  // worker chunks
    const long f = tile * TILE + lane;
    const long fr = f < n_frames ? f : n_frames - 1;  // masked lanes replay the last frame (results dropped)
    double *Yt = Y + (size_t)tile * L * TILE + lane;
    double *Uht = Uh + (size_t)tile * HEAD * TILE + lane;
    double *Utt = Utl + (size_t)tile * (EDGE + 1) * TILE + lane;
#define YAT(p) Yt[(size_t)(p) * TILE]
#ifdef PSS_UBENCH
    if (threadIdx.x == 0 && blockIdx.x < 4096) pss_dbg_stamps[4 * blockIdx.x] = wall_clock64();
#endif
    if (tid >= 128 && tid < 192) ltab[tid - 128] = RCP14_AB[tid - 128];
    // ---- prologue: discriminator of times 0..87 into logical columns 8..95 (physical = logical at chunk 0).
    // Thread (frame tid / 4, part tid % 4) fills the 24 columns 8 + 24 part .. of its frame: its 26 samples come as thirteen 16-byte
    // buffer loads requested together, the discriminator values as three batches of eight interleaved branch-free chains, the
    // columns leave as six 16-byte LDS stores.  (One sample per thread and pass — load, wait, evaluate, store, 24 times — took 25 us
    // of the kernel's 520: every pass paid the full memory latency.)
    {
        typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
        const int fl = tid >> 2, part = tid & 3, t0 = 24 * part;
        const long tfr = (n_frames - tile * TILE) < TILE ? (n_frames - tile * TILE) : TILE;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(iq + (size_t)tile * TILE * n), 0,
                                                                             (int)(tfr * n * (long)sizeof(float2)), 0x00020000);
        const int voff = (int)((fl < tfr ? fl : tfr - 1) * (long)n * (long)sizeof(float2)) + t0 * (int)sizeof(float2);
        float2 x[26];
        v4u_t raw[13];
#pragma unroll
        for (int g = 0; g < 13; g++) raw[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 16 * g, 0);   // all thirteen requests first
        __syncthreads();      // the reciprocal table's LDS copy (a lookup in constant memory is a ~300-clock vector load in every chain)
#pragma unroll
        for (int g = 0; g < 13; g++) {
            x[2 * g] = make_float2(__uint_as_float(raw[g].x), __uint_as_float(raw[g].y));
            x[2 * g + 1] = make_float2(__uint_as_float(raw[g].z), __uint_as_float(raw[g].w));
        }
        float4 *dst = reinterpret_cast<float4 *>(win + fl * WSTR + 8 + t0);
        if (part == 0) *reinterpret_cast<float4 *>(win + fl * WSTR) = make_float4(0.0f, 0.0f, 0.0f, 0.0f),
                       *reinterpret_cast<float4 *>(win + fl * WSTR + 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // columns 0..7: times -8..-1
#pragma unroll
        for (int b = 0; b < 3; b++) {
            float d[8];
            bool ok[8], allok = true;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                d[e] = disc_sample_main(x[8 * b + e + 1], x[8 * b + e], kscale, SWAPPED, ok[e], ltab);
                allok = allok && ok[e];
            }
            if (__builtin_expect(!allok, 0)) {
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (!ok[e]) d[e] = disc_sample(x[8 * b + e + 1], x[8 * b + e], kscale, SWAPPED, ltab);
            }
            if (t0 + 8 * b < WCOLS - 8) {      // part 3 owns only 16 columns (times 72..87)
#pragma unroll
                for (int e = 0; e < 8; e++) d[e] = (t0 + 8 * b + e < M) ? d[e] : 0.0f;
                dst[2 * b] = make_float4(d[0], d[1], d[2], d[3]);
                dst[2 * b + 1] = make_float4(d[4], d[5], d[6], d[7]);
            }
        }
    }
    __syncthreads();
#ifdef PSS_UBENCH
    if (threadIdx.x == 0 && blockIdx.x < 4096) pss_dbg_stamps[4 * blockIdx.x + 2] = wall_clock64();
#endif
    // The first 64 FIR outputs have windows shorter than 65 samples, i.e. each its own ddot shape.  With lane = frame
    // the length is wave-uniform: the four waves take 16 outputs each (interleaved, so the dot lengths balance) and
    // park them in Uh (L2) for the IIR wave.
    {
        const float *row0 = win + lane * WSTR + 8;  // time t of this lane's frame at row0[t] (chunk-0 layout)
#pragma unroll 1
        for (int i = wave; i < HEAD; i += 1 + NFW) {
            Uht[(size_t)i * TILE] = ddot_head(row0, d_rev + (64 - i), i + 1);   // taps by scalar loads (wave-uniform address)
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
#ifdef PSS_UBENCH
    if (threadIdx.x == 0 && blockIdx.x < 4096) pss_dbg_stamps[4 * blockIdx.x + 3] = wall_clock64();
#endif
    if (wave == 0) {
        // =============================== IIR wave ===============================
        Iir4 st;
        long p = 0;  // stream position of the next input; outputs lag by 3
#ifdef PSS_EXP_NOSTORE   // timing experiment only: y_fwd rows all land on the tile's first rows (L2-resident)
        auto emit = [&](double v) { YAT((p - 3) & 15) = v; };
#else
        // (non-temporal stores here, so that the 614 MB of y_fwd rows do not push IQ lines out of the L2 between the two chunks that touch them:
        // FETCH_SIZE 747 -> 693 MB, launch time unchanged at 0.459 / 0.460 ms in a paired A/B — NOTEBOOK R5-07; not kept)
        auto emit = [&](double v) { YAT(p - 3) = v; };
#endif
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
#ifndef PSS_EXP_NOPRIO
                // progress balancing: the workgroup publishes its chunk index and ranks it among the (up to) four workgroups of its CU —
                // the one furthest behind issues first
                {
                    unsigned *pc = prog + 4 * cu;
                    // a word = launch epoch (upper 16 bits) | chunk index + 1: what an earlier launch left behind never counts as "ahead"
                    const unsigned mine = (epoch << 16) | (((unsigned)ch + 1u) & 0xffffu);
                    if (lane == 0) __hip_atomic_store(pc + slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    int rank = 0;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const unsigned v = __hip_atomic_load(pc + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        rank += ((v >> 16) == (epoch & 0xffffu) && v > mine) ? 1 : 0;
                    }
                    rank = __builtin_amdgcn_readfirstlane(rank);
                    if (lane == 0) *lprio = rank;
                    prio_rotate(rank);
                }
#endif
                lds_barrier();  // A: workers finished FIR(ch) -> ubuf
#pragma unroll
                for (int t = 0; t < FC; t++) reg[t] = ubuf[t * TILE + lane];
                lds_barrier_b();  // B
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
        // the tile's frames as a buffer resource; masked lanes (frames past the batch) replay the tile's last frame
        const long tfr = (n_frames - tile * TILE) < TILE ? (n_frames - tile * TILE) : TILE;
        const __amdgpu_buffer_rsrc_t rs_iq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(iq + (size_t)tile * TILE * n), 0,
                                                                                (int)(tfr * n * (long)sizeof(float2)), 0x00020000);
        const int voff_iq = (int)((lane < tfr ? lane : tfr - 1) * (long)n * (long)sizeof(float2));
        int rot = 0;  // physical block of logical block 0
        for (int ch = 0; ch <= NC; ch++) {
            if (ch < NC) {
                const int ibase = HEAD + ch * FC + OPT * J;  // first output index of this thread's batch
                const int tn = HEAD + (ch + 1) * FC + OPT * J;  // time of this thread's first new sample of the next chunk
                // ring position of this thread's first window column (logical 8 + OPT J) in the chunk's rotation: wave-uniform, a multiple of 4
                int q = 8 + OPT * J + FC * rot;
                q = q >= WCOLS ? q - WCOLS : q;
                const unsigned row_addr = (unsigned)(uintptr_t)row;
                const bool tail = ibase + OPT > M - 1 - EDGE;      // (wave-uniform) some of these outputs also go to the tail scratch
                // The next chunk's nine IQ samples are REQUESTED here, ahead of the FIR, and consumed behind it (18 VGPRs beside the FIR
                // statement's 94): a CU's wavefronts run the chunk in lockstep, so a load waited for right behind its issue is a stall every
                // SIMD takes at the same moment.  Buffer loads (the tile's frames as one resource in SGPRs, one 32-bit byte offset per lane,
                // the sample index in the scalar offset): no 64-bit per-lane pointer lives across the FIR statement, and a read past the
                // tile's last sample returns zeros instead of needing a branch.
                typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
                v2u_t raw[OPT + 1];
#pragma unroll
                for (int e = 0; e <= OPT; e++) {
#ifdef PSS_EXP_L2IQ   // timing experiment only: every chunk re-reads the frame's first samples (always cache hits)
                    raw[e] = __builtin_amdgcn_raw_buffer_load_b64(rs_iq, voff_iq, ((tn & 31) + e) * 8, 0);
#else
                    raw[e] = __builtin_amdgcn_raw_buffer_load_b64(rs_iq, voff_iq, (tn + e) * 8, 0);
#endif
                }
#pragma unroll 1
                for (int h = 0; h < OPT / FBH; h++) {
                    int qh = q + FBH * h;
                    qh = qh >= WCOLS ? qh - WCOLS : qh;
                    double *us = ubuf + (OPT * J + FBH * h) * TILE + lane;
                    // the hand-scheduled FIR (pss_fir_ring_asm.h): four outputs straight into the u chunk buffer
                    fir_ring_asm(row_addr, (unsigned)(uintptr_t)us, qh, d_rev);
                    if (tail) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the statement's own LDS writes (the compiler does not count them)
#pragma unroll
                        for (int o = 0; o < FBH; o++) {
                            const int i = ibase + FBH * h + o;
                            if (i < M && i >= M - 1 - EDGE) Utt[(size_t)(i - (M - 1 - EDGE)) * TILE] = us[o * TILE];
                        }
                    }
                }
                // discriminator of the next chunk's 8 samples (9 complex, requested above); kept in registers until the window block is free
                float dn[OPT];
                {
                    // The eight discriminator values are evaluated as eight interleaved branch-free chains (the routine's main path; `ok` =
                    // operands in its range) with ONE almost-never-taken fix-up block behind them — per-sample `e < left` branches put every
                    // sample into a basic block of its own, i.e. eight serial ~45-instruction dependent chains (208 clocks per sample in
                    // the role benchmark).
                    float2 pre[OPT + 1];
                    const int left = M - tn;  // discriminator samples of this batch inside the frame (wave-uniform)
#pragma unroll
                    for (int e = 0; e <= OPT; e++) {
                        // (the empty statement uses the loaded value unconditionally: without it the compiler sinks every load into the
                        // `e <= left` arm — a branch and a full vmcnt(0) wait per sample)
                        asm volatile("" : "+v"(raw[e]));
                        pre[e] = (e <= left) ? make_float2(__uint_as_float(raw[e].x), __uint_as_float(raw[e].y)) : make_float2(0.0f, 0.0f);
                    }
                    bool ok[OPT], allok = true;
#pragma unroll
                    for (int e = 0; e < OPT; e++) {
                        dn[e] = disc_sample_main(pre[e + 1], pre[e], kscale, SWAPPED, ok[e], ltab);
                        allok = allok && (ok[e] || e >= left);
                    }
                    if (__builtin_expect(!allok, 0)) {   // zeros, denormals, infinities, NaNs: the routine's special-value path
#pragma unroll
                        for (int e = 0; e < OPT; e++)
                            if (!ok[e] && e < left) dn[e] = disc_sample(pre[e + 1], pre[e], kscale, SWAPPED, ltab);
                    }
#pragma unroll
                    for (int e = 0; e < OPT; e++) dn[e] = (e < left) ? dn[e] : 0.0f;
                }
                lds_barrier();  // A
#ifndef PSS_EXP_NOPRIO
                prio_rotate(__builtin_amdgcn_readfirstlane(*lprio));
#endif
                // the oldest block (logical 0) becomes the newest (logical 3 of the next chunk)
                {
                    float4 *nb = reinterpret_cast<float4 *>(row + rot * FC + OPT * J);   // 16-byte stores (rows are 16-byte aligned, 25 x 16 bytes apart)
#pragma unroll
                    for (int e = 0; e < OPT / 4; e++) nb[e] = make_float4(dn[4 * e], dn[4 * e + 1], dn[4 * e + 2], dn[4 * e + 3]);
                }
                rot = (rot + 1) & (NB - 1);
                lds_barrier_b();  // B
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();  // A
                __syncthreads();  // B
            }
        }
    }
#ifdef PSS_UBENCH
    if (lane == 0 && blockIdx.x < 4096) atomicMax(&pss_dbg_stamps[4 * blockIdx.x + 1], wall_clock64());
#endif
#undef YAT
}


// Backward pass of sosfiltfilt over y_fwd (read in reverse from Y[tile][p][lane]), decimation [::q], peak
// normalisation, stereo int16 / float64 audio.  One wavefront per tile, lane = frame.
// WFM: a workgroup is the TWO rows of a tile of frames — wavefront 0 the left channel (row tile 2 t), wavefront 1 the right one (2 t + 1) —
// so that demodulate_wfm's joint peak normalisation of the two decimated channels (:157-163: audio / max(max|l|, max|r|), column_stack, no
// 0.95) happens here: the two peaks meet in LDS and each wavefront writes its channel of the stereo frames (until R5-11 a separate
// k_wfm_finalize launch read both peaks and both rows back: 0.047 ms of the 1.55 ms WFM step).
template <bool B121, bool WFM = false>
__global__ __launch_bounds__(WFM ? 2 * TILE : TILE) void k_nfm_bwd(const double *__restrict__ Y, double *__restrict__ A, int n, int q,
                                                  int n_out, long n_frames, NfmCoef c, int16_t *__restrict__ pcm,
                                                  double *__restrict__ audio)
{
#ifdef PSS_EXP_BWD_PRIO     // timing experiment: the backward pass's user priority while it runs beside the display chain (reset at the end)
    __builtin_amdgcn_s_setprio(PSS_EXP_BWD_PRIO);
#endif
    const int lane = threadIdx.x & (TILE - 1);
    const int chan = WFM ? (int)(threadIdx.x >> 6) : 0;
    const long tile = WFM ? 2 * (long)blockIdx.x + chan : (long)blockIdx.x;
    const long f = (long)blockIdx.x * TILE + lane;
    const int M = n - 1;
    const long L = (long)M + 2 * EDGE;
    const double *Yt = Y + (size_t)tile * L * TILE + lane;
    double *At = A + (size_t)tile * n_out * TILE + lane;
#ifdef PSS_EXP_BWD_NOLOAD   // timing experiment only (results wrong): the backward pass without its y_fwd reads
#define YAT(p) ((double)(p) * 1e-3 + (double)lane)
#else
#define YAT(p) Yt[(size_t)(p) * TILE]
#endif
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
    if constexpr (WFM) {
        __shared__ double mxs[2][TILE];
        mxs[chan][lane] = mx;
        __syncthreads();
        const double ml = mxs[0][lane], mr = mxs[1][lane];
        const double mxj = (mr > ml) ? mr : ml;          // python max(a, b): b only if b > a
        if (f < n_frames) {
            for (int k = 0; k < n_out; k++) {
                const double a = __ddiv_rn(At[(size_t)k * TILE], mxj);
                if (audio) audio[2 * ((size_t)f * n_out + k) + chan] = a;
                if (pcm) pcm[2 * ((size_t)f * n_out + k) + chan] = pcm16(a);
            }
        }
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
#ifdef PSS_EXP_BWD_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#undef YAT
}

}  // namespace fused
