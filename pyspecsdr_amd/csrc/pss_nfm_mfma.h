// pss_nfm_mfma.h — OPT-IN variant of the fused NFM forward kernel (option "fir_mfma"): the 65-tap FIR on the matrix pipe.
// Included by pss_demod.hip after pss_nfm_fused.h.
//
// Why it is not the default.  v_mfma_f64_16x16x4_f64 is, bit for bit, a k-ascending chain of IEEE FMAs per output element
// (probed on gfx950 against 51 200 random elements, tools/ubench/mfma_f64_probe.hip), so WHAT it computes is exactly defined —
// but the reference's lfilter is np.convolve -> OpenBLAS ddot, whose order is 32 two-product partial sums, a 31-addition
// tree and a tail FMA.  Mapped onto the instruction that order wastes half the K slots and 62 % of the output band (5 x
// slower than the VALU).  The efficient mapping is the Toeplitz form below — u[n] = sum over m ascending of
// taps[n - m] x[m], one 65-term FMA chain per output — which differs from OpenBLAS's sums in the last bits (~1e-16
// relative): float64 audio is then NOT bit-identical; the int16 samples are, unless a value falls within ~3e-11 of an
// integer boundary (measured mismatch counts: DESIGN.md §4).  The matrix pipe peaks at the VALU's own float64 rate
// (78.6 TFLOP/s), so the gain is concurrency: the 96 float64 + 23 conversion instructions per sample of the exact FIR
// leave the VALU to the discriminator and the IIR.
//
// One 192-thread workgroup per tile of 64 frames:
//   wave 0      IIR wave, lane = frame, exactly as in pss_nfm_fused.h (consumes u in chunks of 16 time steps from LDS)
//   waves 1, 2  FIR workers: worker w owns the frame blocks 2w, 2w+1 (16 frames each).  Per chunk of 16 outputs and block:
//               D[16 outputs][16 frames] = T[16][80] x X[80][16] as 20 MFMAs, A = the Toeplitz band of the taps (20
//               per-lane constants), B = the float32 discriminator window in LDS converted on the way in; then the
//               discriminator of the next 16 time steps of its 32 frames (8 samples per lane).
// The window is a ring of 96 columns (times t-80 .. t+15); two LDS-only barriers per chunk, as in the exact kernel.
#pragma once

namespace fusedm {

using namespace pss;
using fused::Iir4;
using fused::lds_barrier;
using fused::pipe_step;
using fused::sec_step;

constexpr int FC = 16;         // time steps per chunk = rows of one MFMA result
constexpr int WCOLS = 96;      // window ring: column of time m is (m + 96) % 96 (m >= -96)
constexpr int WSTR = 97;       // float row stride (odd: conflict-free for lane = frame at a fixed column)
constexpr int HEAD = fused::HEAD;   // outputs the IIR wave takes from the global head scratch (the odd extension needs u[0..27] first)
constexpr int NW = 2;          // worker waves
constexpr int WG = 64 * (1 + NW);
constexpr int KB = 20;         // MFMAs per result block: 80 window times / 4
constexpr size_t LDS_BYTES = (size_t)TILE * WSTR * sizeof(float) + (size_t)TILE * FC * sizeof(double);

typedef double v4d __attribute__((ext_vector_type(4)));

template <bool B121>
__global__ __launch_bounds__(WG) void k_nfm_fwd_mfma(const float2 *__restrict__ iq, double *__restrict__ Y,
                                                     double *__restrict__ Uh, double *__restrict__ Utl, int n, long n_frames,
                                                     NfmCoef c, float kscale, int swapped, TapsArg taps)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float *win = reinterpret_cast<float *>(smem);                                            // [TILE][WSTR]
    double *ubuf = reinterpret_cast<double *>(smem + (size_t)TILE * WSTR * sizeof(float));   // [FC][TILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long tile = blockIdx.x;
    const int M = n - 1;                      // FIR outputs per frame (M >= 128 guaranteed by the caller)
    const long L = (long)M + 2 * EDGE;
    const int NCH = (M + FC - 1) / FC;        // chunks of FIR outputs
    double *Yt = Y + (size_t)tile * L * TILE + lane;
    double *Uht = Uh + (size_t)tile * HEAD * TILE;
    double *Utt = Utl + (size_t)tile * (EDGE + 1) * TILE;
#define YAT(p) Yt[(size_t)(p) * TILE]
    // ---- prologue: window times -80 .. -1 are zero, times 0 .. 15 the first discriminator samples
    for (int idx = tid; idx < TILE * WCOLS; idx += WG) {
        const int fl = idx / WCOLS, col = idx % WCOLS;
        const int t = col < FC ? col : col - WCOLS;   // column of time t is (t + 96) % 96
        const long ff = tile * TILE + fl;
        const float2 *x = iq + (size_t)(ff < n_frames ? ff : n_frames - 1) * n;
        float d = 0.0f;
        if (t >= 0 && t < M) d = disc_sample(x[t + 1], x[t], kscale, swapped != 0);
        win[fl * WSTR + col] = d;
    }
    __syncthreads();
    if (wave == 0) {
        // =============================== IIR wave ===============================
        // The recurrence is a dependent chain (~350 clocks per time step even alone on a SIMD, as k_nfm_bwd shows): with the FIR
        // off the VALU this wavefront is the workgroup's critical path, so it takes precedence over the workers it shares
        // its SIMD with.
        __builtin_amdgcn_s_setprio(3);
        // chunks 0 .. 3 (outputs 0 .. 63) go to the global head scratch as well; the recurrence starts once they are there
        for (int ch = 0; ch < HEAD / FC; ch++) { lds_barrier(); lds_barrier(); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        __syncthreads();
        const double *Uhl = Uht + lane;
        Iir4 st;
        long p = 0;  // stream position of the next input; outputs lag by 3
        auto emit = [&](double v) { YAT(p - 3) = v; };
        const double u0 = Uhl[0];
        const double two_u0 = __dmul_rn(2.0, u0);
        auto head = [&](int pp) { return pp < EDGE ? __dsub_rn(two_u0, Uhl[(size_t)(EDGE - pp) * TILE]) : Uhl[(size_t)(pp - EDGE) * TILE]; };
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
        double reg[FC];
        for (int ch = HEAD / FC; ch <= NCH; ch++) {
            if (ch > HEAD / FC) {
                const int cnt = (M - (ch - 1) * FC) < FC ? (M - (ch - 1) * FC) : FC;
                if (cnt == FC) {
#pragma unroll
                    for (int t = 0; t < FC; t++) { double v = pipe_step<B121>(c, st, reg[t]); emit(v); p++; }
                } else {
#pragma unroll
                    for (int t = 0; t < FC; t++)
                        if (t < cnt) { double v = pipe_step<B121>(c, st, reg[t]); emit(v); p++; }
                }
            }
            if (ch < NCH) {
                lds_barrier();  // A: workers finished FIR(ch) -> ubuf
#pragma unroll
                for (int t = 0; t < FC; t++) reg[t] = ubuf[t * TILE + lane];
                lds_barrier();  // B
            } else {
                __syncthreads();  // final A / B: full fences, the workers' global tail rows must be visible
                __syncthreads();
            }
        }
        // odd extension tail: ext[27+M+k] = 2u[M-1] - u[M-2-k]; Utt[r] = u[M-28+r]
        {
            const double *Utl2 = Utt + lane;
            const double two_uL = __dmul_rn(2.0, Utl2[(size_t)EDGE * TILE]);
#pragma unroll 3
            for (int k = 0; k < EDGE; k++) {
                double v = pipe_step<B121>(c, st, __dsub_rn(two_uL, Utl2[(size_t)(EDGE - 1 - k) * TILE]));
                emit(v); p++;
            }
        }
        { double v = sec_step<B121>(c, st, 3, st.p2); emit(v); p++; st.p2 = sec_step<B121>(c, st, 2, st.p1); st.p1 = sec_step<B121>(c, st, 1, st.p0); }
        { double v = sec_step<B121>(c, st, 3, st.p2); emit(v); p++; st.p2 = sec_step<B121>(c, st, 2, st.p1); }
        { double v = sec_step<B121>(c, st, 3, st.p2); emit(v); p++; }
    } else {
        // =============================== FIR workers ===============================
        const int w = wave - 1;                       // owns frame blocks 2w, 2w + 1
        const int li = lane & 15, lk = lane >> 4;     // MFMA roles: A[i = li][k = lk], B[k = lk][frame = li], D[i = 4 r + lk][frame = li]
        // Toeplitz band: MFMA j of a block multiplies window times 16 ch - 64 + 4 j + k; output i = li needs tap (i + 64 - 4 j - k)
        double A[KB];
#pragma unroll
        for (int j = 0; j < KB; j++) {
            const int t = li + 64 - 4 * j - lk;
            A[j] = (t >= 0 && t <= 64) ? taps.fwd[t] : 0.0;
        }
        // discriminator role: frame = 32 w + (lane & 31), eight time steps (lane >> 5) * 8 .. + 7 of the next chunk
        const int dfl = 32 * w + (lane & 31), dt0 = (lane >> 5) * 8;
        const long dff = tile * TILE + dfl;
        const float2 *xq = iq + (size_t)(dff < n_frames ? dff : n_frames - 1) * n;
        float2 nxt[9];
#pragma unroll
        for (int e = 0; e <= 8; e++) nxt[e] = (FC + dt0 + e < n) ? xq[FC + dt0 + e] : make_float2(0.0f, 0.0f);
        for (int ch = 0; ch <= NCH; ch++) {
            if (ch < NCH) {
                const int n0 = ch * FC;
                // discriminator inputs of the next chunk's 16 time steps (8 per lane): times n0 + 16 + dt0 + e; the loads are
                // issued first, the arithmetic is interleaved with the MFMA chain below — a wavefront issues in order, and a
                // dependent MFMA (same accumulator, 64 clocks apart) would otherwise stall everything queued behind it
                const int tn = n0 + FC + dt0;
                float2 pre[9];
#pragma unroll
                for (int e = 0; e <= 8; e++) pre[e] = nxt[e];
                // ... and the chunk after that is requested now: a chunk is only a few thousand clocks long, a load issued and
                // consumed inside one chunk exposes the whole memory latency on every chunk
#pragma unroll
                for (int e = 0; e <= 8; e++) nxt[e] = (tn + FC + e < n) ? xq[tn + FC + e] : make_float2(0.0f, 0.0f);
                float dn[8];
                // window column of time n0 - 64 + 4 j + lk, j = 0: (n0 - 64 + lk + 96) % 96, then + 4 per j
                int col = (n0 + 32 + lk) % WCOLS;
                {
                    // the worker's two frame blocks side by side: two independent accumulator chains, so consecutive MFMAs
                    // never depend on each other, and one discriminator sample after every five pairs
                    const float *row0 = win + (16 * (2 * w) + li) * WSTR, *row1 = row0 + 16 * WSTR;
                    v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                    int cj = col;
#pragma unroll
                    for (int j = 0; j < KB; j++) {
                        const double b0 = (double)row0[cj], b1 = (double)row1[cj];
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[j], b0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(A[j], b1, acc1, 0, 0, 0);
                        cj += 4;
                        cj = cj >= WCOLS ? cj - WCOLS : cj;
                        if (j % 5 == 4) {
#pragma unroll
                            for (int e = 2 * (j / 5); e < 2 * (j / 5) + 2; e++)
                                dn[e] = tn + e < M ? disc_sample(pre[e + 1], pre[e], kscale, swapped != 0) : pre[e].x;
                        }
                    }
                    // scheduling: after every MFMA a slice of the discriminator's VALU work (the matrix pipe needs 64 clocks per
                    // instruction; an in-order wavefront that queues MFMAs back to back just waits)
#pragma unroll
                    for (int g = 0; g < 2 * KB; g++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 18, 0);   // eighteen VALU
                    }
#pragma unroll
                    for (int blk = 0; blk < 2; blk++) {
                        const int fb = 2 * w + blk;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int i = 4 * r + lk, o = n0 + i;   // output index
                            const double v = blk ? acc1[r] : acc0[r];
                            ubuf[i * TILE + 16 * fb + li] = v;
                            if (o < HEAD) Uht[(size_t)o * TILE + 16 * fb + li] = v;
                            if (o < M && o >= M - 1 - EDGE) Utt[(size_t)(o - (M - 1 - EDGE)) * TILE + 16 * fb + li] = v;
                        }
                    }
                }
                if (ch < HEAD / FC) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // head rows go through global memory
                lds_barrier();  // A
                {
                    // times n0 + 16 .. n0 + 31 replace times n0 - 80 .. n0 - 65 (no longer needed by any later chunk)
                    float *nb = win + dfl * WSTR;
                    const int c0 = (n0 + FC + dt0) % WCOLS;
#pragma unroll
                    for (int e = 0; e < 8; e++) { const int cc = c0 + e; nb[cc >= WCOLS ? cc - WCOLS : cc] = dn[e]; }
                }
                lds_barrier();  // B
                if (ch == HEAD / FC - 1) {   // the IIR wave starts: make the head rows visible
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __syncthreads();
                }
            } else {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();  // A
                __syncthreads();  // B
            }
        }
    }
#undef YAT
}

}  // namespace fusedm
