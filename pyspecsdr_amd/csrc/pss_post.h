// pss_post.h — the caller's spectrum post-process (pyspecsdr.py:2278-2283) and the batched display accumulators
// (pyspecsdr.py:1342-1406 waterfall, :1512-1564 persistence) for whole batches of dB rows.  Included by pss_fft.hip.
//
// k_post_sel<EPL, W>: 5-tap moving average ('valid', float64 products and sums as np.convolve forms them), the median of
// the smoothed row, clamp below median - 10.  A row is owned by W wavefronts (W = 1: four independent rows per
// 256-thread workgroup and not a single workgroup barrier); thread t of T = 64 W keeps
// the smoothed row in registers — EPL CONSECUTIVE elements, transposed through LDS on the way in and out, so that global
// accesses are 16 bytes per lane and contiguous per wavefront and each sample is converted to float64 once.  The median is a
// bit-by-bit binary search over the order-preserving integer image of the float32 values, below the common prefix of the
// row's extremes and until a single candidate is left (about a dozen steps for a dB row): per step every register slot is
// compared with the trial value, the per-lane counts are summed over the wavefront on the DPP crossbar — no LDS, no sort.
// (The previous kernel bitonic-sorted the row in LDS: 55 barrier stages to find two order statistics.)
// Optionally the finite minimum / maximum of each clamped row is written out: the display accumulators normalise with the
// extremes of the last 30 (10) rows, which then costs 8 bytes per row to look up instead of a pass over 30 rows.
#pragma once
#include <hip/hip_runtime.h>
// From here to the end of the including unit (pss_fft.hip: this header, the display quantisers, the Bluestein functors) float
// expressions are evaluated as written — no fused multiply-adds the reference's NumPy statements do not contain (np.convolve's
// products and sums, the normalisation / interpolation arithmetic of the display code).  (pss_device.h already switches
// contraction off for every unit that includes it — the transform kernels in pss_fft_r16.h / pss_fft_xl.h write the fused
// multiply-adds they want as explicit fma(); the pragma here keeps this header correct on its own.)
#pragma clang fp contract(off)

namespace pss_post {

// order-preserving integer image of a float32 (unsigned compare = float compare; -0 < +0; NaNs at the two ends), branch-free
__device__ __forceinline__ unsigned f2ord(float v)
{
    const unsigned u = __float_as_uint(v);
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float(o ^ ((unsigned)((int)~o >> 31) | 0x80000000u)); }
constexpr unsigned ORD_NEG_INF = 0x007fffffu, ORD_POS_INF = 0xff800000u;  // images of -inf / +inf: finite values lie strictly between

// The same image for float64 (the reference's own row type: compute_fft returns float64 and the caller smooths, clamps and draws in
// float64 — pss_frame_pipeline_nfm_f64), and the per-type constants the select works with.
__device__ __forceinline__ unsigned long long d2ord(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return u ^ ((unsigned long long)((long long)u >> 63) | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord2d(unsigned long long o)
{
    return __longlong_as_double((long long)(o ^ ((unsigned long long)((long long)~o >> 63) | 0x8000000000000000ull)));
}
template <class T> struct Ord;
template <> struct Ord<float> {
    using K = unsigned;
    static constexpr K PAD = 0xffffffffu, NEG_INF = ORD_NEG_INF, POS_INF = ORD_POS_INF;
    static constexpr int BITS = 32;
    static __device__ __forceinline__ K enc(float v) { return f2ord(v); }
    static __device__ __forceinline__ float dec(K k) { return ord2f(k); }
    static __device__ __forceinline__ int top_bit(K x) { return 31 - __builtin_clz(x); }
};
template <> struct Ord<double> {
    using K = unsigned long long;
    static constexpr K PAD = 0xffffffffffffffffull, NEG_INF = 0x000fffffffffffffull, POS_INF = 0xfff0000000000000ull;
    static constexpr int BITS = 64;
    static __device__ __forceinline__ K enc(double v) { return d2ord(v); }
    static __device__ __forceinline__ double dec(K k) { return ord2d(k); }
    static __device__ __forceinline__ int top_bit(K x) { return 63 - __builtin_clzll(x); }
};

// ---- wavefront reductions on the DPP crossbar (no LDS, no SALU chains) ---------------------------------------------------
// quad_perm / row_half_mirror / row_mirror leave every lane of a 16-lane row with its row's result; the four row results are
// then read with v_readlane and combined on the scalar unit.  K = unsigned or unsigned long long (two DPP moves per step).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_k(unsigned v) { return dpp_u32<CTRL>(v); }
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_k(unsigned long long v)
{
    const unsigned lo = dpp_u32<CTRL>((unsigned)v), hi = dpp_u32<CTRL>((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned readlane_k(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ unsigned long long readlane_k(unsigned long long v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
// v_min / v_max as they are: -0 < +0, a NaN operand loses (fminf / fmin cost a canonicalising v_max x, x per operand in front of each; the
// callers' operands are results of float additions / conversions)
__device__ __forceinline__ float vmin_f32(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax_f32(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
struct OpAdd { template <class K> static __device__ __forceinline__ K f(K a, K b) { return a + b; } };
struct OpFAdd { static __device__ __forceinline__ unsigned f(unsigned a, unsigned b) { return __float_as_uint(__uint_as_float(a) + __uint_as_float(b)); } };
struct OpFMin { static __device__ __forceinline__ unsigned f(unsigned a, unsigned b) { return __float_as_uint(vmin_f32(__uint_as_float(a), __uint_as_float(b))); } };
struct OpFMax { static __device__ __forceinline__ unsigned f(unsigned a, unsigned b) { return __float_as_uint(vmax_f32(__uint_as_float(a), __uint_as_float(b))); } };
struct OpMin { template <class K> static __device__ __forceinline__ K f(K a, K b) { return a < b ? a : b; } };
struct OpMax { template <class K> static __device__ __forceinline__ K f(K a, K b) { return a > b ? a : b; } };

template <class Op, class K>
__device__ __forceinline__ K wave_reduce(K v)
{
    v = Op::f(v, dpp_k<0xB1>(v));    // quad_perm [1,0,3,2]
    v = Op::f(v, dpp_k<0x4E>(v));    // quad_perm [2,3,0,1]
    v = Op::f(v, dpp_k<0x141>(v));   // row_half_mirror
    v = Op::f(v, dpp_k<0x140>(v));   // row_mirror
    const K a = readlane_k(v, 0), b = readlane_k(v, 16), c = readlane_k(v, 32), d = readlane_k(v, 48);
    return Op::f(Op::f(a, b), Op::f(c, d));
}

// the same over the W wavefronts of a row (W = 1: nothing more to do); `red` (2 W 64-bit slots) is double-buffered: one barrier per reduction
template <class Op, int W, class K>
__device__ __forceinline__ K row_reduce(K v, unsigned long long *red, int wave, int lane, int &phase)
{
    v = wave_reduce<Op>(v);
    if constexpr (W == 1) return v;
    K *slot = reinterpret_cast<K *>(red + (phase & 1) * W);
    phase++;
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    K s = slot[0];
#pragma unroll
    for (int w = 1; w < W; w++) s = Op::f(s, slot[w]);
    return s;
}

// number of keys of the row below `trial`.  The per-lane count is accumulated on the vector unit (v_cmp + v_addc per
// element) and reduced once: counting each register slot with ballot + s_bcnt1 + s_add instead put 2 scalar instructions
// per element on the CU's single scalar unit and made the kernel scalar-issue bound (measured: 0.23 ms at 65536 x 1024).
template <int EPL, int W, class K>
__device__ __forceinline__ unsigned count_below(const K (&key)[EPL], K trial, unsigned long long *red, int wave, int lane, int &phase)
{
    unsigned c = 0;
#pragma unroll
    for (int r = 0; r < EPL; r++) c += key[r] < trial ? 1u : 0u;
    return row_reduce<OpAdd, W>(c, red, wave, lane, phase);
}

// k-th smallest (0-based) of the row's n_valid keys (padding slots hold the all-ones key) and, in `next`, the (k+1)-th: binary
// search on the key bits below the common prefix of the row's minimum and maximum (returned in mn / mx); stops as soon as
// one candidate is left in the bracket, and that last pass also finds the smallest key above the bracket (= rank k + 1).
// below / upto: the numbers of keys < and <= the returned value (the ranks below .. upto - 1 hold that value).
// PAD_FROM: first register slot that may hold padding (EPL: none anywhere).
// guess: enc(d) = the key of (guess + d) for a value near which the order statistic is expected (the row's mean: most of a dB row is
// noise floor); have_guess = false: none.  The brackets [guess - 0.5, guess + 0.5] and [guess - 2, guess + 2] are tried first — two
// counts tell whether rank k lies inside — and the search then halves such a bracket instead of spending its first steps on the empty
// stretch between the noise floor and the row's extremes (measured on FM / noise / weak-tone rows: 10.4-10.7 counting passes instead of
// 15-19).  A wrong guess costs four counts and changes nothing else: every result is decided by exact counts.
template <int EPL, int W, int PAD_FROM, class K, class Enc>
__device__ __forceinline__ K select_kth(const K (&key)[EPL], unsigned k, unsigned n_valid, K &next, K &mn, K &mx, unsigned &below, unsigned &upto,
                                        unsigned long long *red, int wave, int lane, int &phase, bool have_guess, Enc enc)
{
    constexpr K PAD = (K)~(K)0;
    K a = PAD, b0 = 0, b2 = 0;
#pragma unroll
    for (int r = 0; r < EPL; r++) {
        a = key[r] < a ? key[r] : a;                          // padding = all ones never lowers the minimum
        if (r < PAD_FROM) b0 = key[r] > b0 ? key[r] : b0;     // never padding: taken as it is
        else { const K kp = key[r] + 1u; b2 = kp > b2 ? kp : b2; }  // padding wraps to 0, the neutral element
    }
    if (PAD_FROM < EPL) { b2 = b2 ? b2 - 1u : 0u; b0 = b2 > b0 ? b2 : b0; }
    mn = row_reduce<OpMin, W>(a, red, wave, lane, phase);
    mx = row_reduce<OpMax, W>(b0, red, wave, lane, phase);
    // smallest key above v among the row's keys (all ones if there is none but padding)
    auto above = [&](K v) {
        K c = PAD;
#pragma unroll
        for (int r = 0; r < EPL; r++) c = (key[r] > v && key[r] < c) ? key[r] : c;
        return row_reduce<OpMin, W>(c, red, wave, lane, phase);
    };
    if (mn == mx) { next = mn; below = 0; upto = n_valid; return mn; }   // a constant row (n_valid >= 2 wherever next is used)
    if (have_guess && mx != PAD) {
#pragma unroll 1
        for (int attempt = 0; attempt < 2; attempt++) {
            const float d = attempt ? 2.0f : 0.5f;
            K lo = enc(-d), hi = enc(d);
            lo = lo < mn ? mn : lo;
            hi = hi > mx ? mx : hi;
            if (lo > hi) continue;
            unsigned n_lo = count_below<EPL, W>(key, lo, red, wave, lane, phase);             // #keys < lo
            unsigned n_hi = count_below<EPL, W>(key, (K)(hi + 1u), red, wave, lane, phase);   // #keys <= hi
            if (!(n_lo <= k && k < n_hi)) continue;
            // rank k lies in [lo, hi]: halve the bracket until one key (or one value) is left
#pragma unroll 1
            while (n_hi - n_lo > 1u && lo < hi) {
                const K trial = lo + ((hi - lo + 1u) >> 1);
                const unsigned c = count_below<EPL, W>(key, trial, red, wave, lane, phase);
                if (c <= k) { lo = trial; n_lo = c; } else { hi = trial - 1u; n_hi = c; }
            }
            below = n_lo; upto = n_hi;
            if (n_hi - n_lo == 1u) {
                // the one key in [lo, hi] has rank k; exactly k + 1 keys are <= hi, so rank k + 1 is the smallest key above hi
                K cand = PAD, nx = PAD;
                const K span = hi - lo;
#pragma unroll
                for (int r = 0; r < EPL; r++) {
                    const bool in = key[r] - lo <= span;
                    cand = (in && key[r] < cand) ? key[r] : cand;     // (the minimum: padding keys may lie inside the bracket as well)
                    nx = (key[r] > hi && key[r] < nx) ? key[r] : nx;
                }
                cand = row_reduce<OpMin, W>(cand, red, wave, lane, phase);
                next = row_reduce<OpMin, W>(nx, red, wave, lane, phase);
                return cand;
            }
            next = (k + 1u < n_hi) ? lo : above(lo);          // lo == hi: the value lo holds the ranks n_lo .. n_hi - 1
            return lo;
        }
    }
    int b = (int)(8 * sizeof(K)) - 1 - (sizeof(K) == 8 ? __builtin_clzll((unsigned long long)(mn ^ mx)) : __builtin_clz((unsigned)(mn ^ mx)));   // highest bit in which two keys of the row differ
    K lo = mn & ~((((K)2) << b) - 1u);                        // all keys lie in [lo, lo + 2^(b+1))
    unsigned n_lo = 0, n_hi = n_valid;                        // #keys < lo, #keys < lo + 2^(b+1)
#pragma unroll 1
    for (; b >= 0; b--) {
        const K trial = lo | (((K)1) << b);
        const unsigned c = count_below<EPL, W>(key, trial, red, wave, lane, phase);
        if (c <= k) { lo = trial; n_lo = c; } else n_hi = c;
        if (n_hi - n_lo == 1) {
            // one key left in [lo, lo + 2^b): rank k.  Exactly k keys lie below it, so rank k + 1 is the smallest key at or
            // above lo + 2^b: both come out of one sweep (key - lo wraps to huge values for keys below lo).
            K cand = PAD, nx = PAD;
            const K span = ((K)1) << b;
#pragma unroll
            for (int r = 0; r < EPL; r++) {
                const K d = key[r] - lo;
                const bool in = d < span;
                cand = (in && key[r] < cand) ? key[r] : cand;         // (the minimum: padding keys may lie inside the bracket as well)
                nx = (!in && key[r] >= lo && key[r] < nx) ? key[r] : nx;
            }
            cand = row_reduce<OpMin, W>(cand, red, wave, lane, phase);
            next = row_reduce<OpMin, W>(nx, red, wave, lane, phase);
            below = n_lo; upto = n_hi;
            return cand;
        }
    }
    // every bit decided with several equal keys left: lo is repeated n_hi - n_lo times, ranks n_lo .. n_hi - 1
    below = n_lo; upto = n_hi;
    next = (k + 1u < n_hi) ? lo : above(lo);
    return lo;
}

// The same order statistics of 64-bit keys (float64 rows) held as two 32-bit words per element: the search runs on the HIGH words alone
// — sign, exponent and the top 20 mantissa bits: 3e-5 dB apart for a dB value, while a row's neighbours near its median lie ~5e-3 dB apart —
// at the cost of the 32-bit search (v_cmp_u32 + v_addc per element and pass; a 64-bit compare issues at half that rate), and ends with
// ONE key holding rank k's high word in all but ~1 % of the rows; the low words then come out of masked 32-bit reductions.  Rows in which
// several keys share that high word run a second 32-bit search over their low words.  Every result is decided by exact counts.
// v1 / v2: ranks k and k + 1; mn / mx: the row's extreme keys (mx: undefined when a key's high word is all ones — a NaN payload —, in which
// case the caller's `mx < POS_INF` test fails and it takes its per-element path).
template <int EPL, int W, int PAD_FROM>
__device__ __forceinline__ void select_kth64(const unsigned (&kh)[EPL], const unsigned (&kl)[EPL], unsigned k, unsigned n_valid,
                                             unsigned long long &v1, unsigned long long &v2, unsigned long long &mn, unsigned long long &mx,
                                             unsigned long long *red, int wave, int lane, int &phase, float guess)
{
    unsigned nexth, mnh, mxh, below, upto;
    const unsigned vh1 = select_kth<EPL, W, PAD_FROM, unsigned>(kh, k, n_valid, nexth, mnh, mxh, below, upto, red, wave, lane, phase, guess == guess,
                                                                [&](float d) { return (unsigned)(d2ord((double)(guess + d)) >> 32); });
    // smallest / largest low word among the keys whose high word is h
    auto lo_min = [&](unsigned h) {
        unsigned c = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < EPL; r++) c = (kh[r] == h && kl[r] < c) ? kl[r] : c;
        return row_reduce<OpMin, W>(c, red, wave, lane, phase);
    };
    auto lo_max = [&](unsigned h) {
        unsigned c = 0u;
#pragma unroll
        for (int r = 0; r < EPL; r++) c = (kh[r] == h && kl[r] > c) ? kl[r] : c;
        return row_reduce<OpMax, W>(c, red, wave, lane, phase);
    };
    auto join = [](unsigned h, unsigned l) { return ((unsigned long long)h << 32) | l; };
    if (upto - below == 1u) {
        v1 = join(vh1, lo_min(vh1));               // the only key with this high word
        v2 = join(nexth, lo_min(nexth));           // rank k + 1: the smallest key of the next occupied high word
    } else {
        // ranks below .. upto - 1 share the high word: the (k - below)-th smallest low word among them (32-bit search, no guess); the other
        // elements take the padding key.  A group member whose low word is all ones (every value whose float64 image ends in 32 zero bits,
        // e.g. -50.0) looks like padding to that search: those members are counted apart — they are the group's largest keys — and the search
        // runs over the rest.
        unsigned t[EPL], ones = 0;
#pragma unroll
        for (int r = 0; r < EPL; r++) {
            const bool member = kh[r] == vh1;
            t[r] = member ? kl[r] : 0xffffffffu;
            ones += (member && kl[r] == 0xffffffffu) ? 1u : 0u;
        }
        ones = row_reduce<OpAdd, W>(ones, red, wave, lane, phase);
        const unsigned j = k - below, c = upto - below - ones;       // rank inside the group; members below the all-ones keys
        unsigned l1 = 0xffffffffu, nextl = 0xffffffffu;
        if (j < c) {
            unsigned mnl, mxl, b2, u2;
            l1 = select_kth<EPL, W, 0, unsigned>(t, j, c, nextl, mnl, mxl, b2, u2, red, wave, lane, phase, false, [](float) { return 0u; });
            if (j + 1u >= c) nextl = 0xffffffffu;                    // (rank j + 1 of the group is an all-ones member, or lies outside: decided below)
        }
        v1 = join(vh1, l1);
        v2 = (k + 1u < upto) ? join(vh1, nextl) : join(nexth, lo_min(nexth));
    }
    mn = join(mnh, lo_min(mnh));
    mx = join(mxh, lo_max(mxh));
}

// ---- the same order statistics for a row owned by ONE wavefront, by COMPACTION (round 6) -----------------------------------------------
// Every counting pass of select_kth sweeps all EPL register slots of every lane (v_cmp + v_addc per element, then a wave reduction) to
// learn ONE bit of rank k's position, although only the keys inside the guess bracket [lo, hi] (a tenth of a dB row for the +-0.5 dB bracket
// around its mean) can still be rank k.  Here: ONE counting sweep (keys below the bracket), then a COMPACTION sweep that is the bracket's second
// count as well: every key inside goes to LDS (positions from v_mbcnt over the ballots of the slots + a running scalar base: no scan, no
// atomics; the row's idle staging buffer holds a whole row).  More than 64 CPL of them: sweeps over all elements halve the bracket first, then a
// second compaction.  CPL whole 64-bit keys per lane are read back and the search continues on them alone: one v_cmp per lane-slot and pass,
// counted with s_bcnt1 on the ballot — no wave reduction, the bracket arithmetic on the scalar unit — down to the key itself (low words
// included: no masked low-word sweeps afterwards); rank k + 1 is the smallest candidate above it (or, when rank k is the bracket's last key,
// the smallest key above the bracket: one sweep over the high words, one over the low words).
// The row's extreme keys come from v_min_f64 / v_max_f64 over the smoothed values (the caller reduces them before the search) — valid
// because this path is only taken for rows without a NaN (the row's mean, `guess`, is a sum over every element).
// Every result is decided by exact counts, as in select_kth; returns false — v1 / v2 untouched, only the idle staging buffer written — when the
// row does not fit the scheme (NaN or infinite mean, rank k outside both brackets, more than 64 CPL keys sharing one high word): the caller then
// runs select_kth64.
#ifndef PSS_POST_CPL
#define PSS_POST_CPL 2       // candidates per lane after the compaction (3 / 4: measured no faster, NOTEBOOK R6-14)
#endif
#ifndef PSS_POST_CBAR
#define PSS_POST_CBAR 4      // a scheduling barrier after every PSS_POST_CBAR-th slot of the compaction sweep
#endif
// the lanes of one wavefront execute their LDS instructions in order: a fence, no workgroup barrier
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// v_min_f64 / v_max_f64 as they are: -0 < +0, a NaN operand loses (__builtin_fmin costs a canonicalising v_max_f64 x, x per operand in
// front of each of them; the callers' operands are results of float64 additions)
__device__ __forceinline__ double vmin_f64(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax_f64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const unsigned lo = dpp_u32<CTRL>((unsigned)__double2loint(v)), hi = dpp_u32<CTRL>((unsigned)__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ double vmin_f64_s(double a, double b_uniform)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b_uniform));
    return r;
}
__device__ __forceinline__ double vmax_f64_s(double a, double b_uniform)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b_uniform));
    return r;
}
// minimum / maximum over the wavefront's lanes, as the ordered image, wave-uniform (scalar registers)
template <bool MIN>
__device__ __forceinline__ unsigned long long wave_extreme_key_f64(double v)
{
    auto op = [](double a, double b) { return MIN ? vmin_f64(a, b) : vmax_f64(a, b); };
    auto ops = [](double a, double b) { return MIN ? vmin_f64_s(a, b) : vmax_f64_s(a, b); };
    v = op(v, dpp_f64<0xB1>(v));
    v = op(v, dpp_f64<0x4E>(v));
    v = op(v, dpp_f64<0x141>(v));
    v = op(v, dpp_f64<0x140>(v));
    auto rl = [&](int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
    };
    const double r = ops(ops(ops(v, rl(16)), rl(32)), rl(48));    // (lane 0 holds row 0's result: every lane of row 0 ends with the wavefront's)
    const unsigned long long kk = d2ord(r);
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kk >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kk);
}

template <int EPL, int CPL>
__device__ __forceinline__ bool select_kth64_compact(const unsigned (&kh)[EPL], const unsigned (&kl)[EPL], unsigned k, unsigned long long &v1,
                                                     unsigned long long &v2, unsigned long long *cand, int lane, float guess)
{
    using K = unsigned long long;
    constexpr K PADK = ~(K)0;
    constexpr unsigned CAP = 64u * CPL;
    if (!(guess - guess == 0.0f)) return false;               // NaN or infinite mean
    int phase = 0;
    auto join = [](unsigned h, unsigned l) { return ((K)h << 32) | l; };
#pragma unroll 1
    for (int attempt = 0; attempt < 2; attempt++) {
        const float d = attempt ? 2.0f : 0.5f;
        // (finite values: high words below 0xfff00000, so hi + 1 cannot wrap and the padding keys — all ones — are never counted)
        // wave-uniform by construction; said so (v_readfirstlane), the search's 64-bit bracket arithmetic below runs on the scalar unit instead of
        // as a dependent chain of a dozen vector instructions per pass
        unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(d2ord((double)(guess - d)) >> 32));
        unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(d2ord((double)(guess + d)) >> 32));
        unsigned n_lo = count_below<EPL, 1>(kh, lo, nullptr, 0, lane, phase);            // #keys below the bracket
        // The compaction sweep is the bracket's second count as well: every key inside [lo, hi] goes to LDS (slot r's keys take the positions
        // after those of the slots before it, in lane order; the staging buffer holds a whole row) and their number M tells whether rank k
        // lies inside.  More than CAP of them (the wide bracket): sweeps over all elements halve the bracket first, then a second compaction.
        unsigned M = 0;
        bool inside = true;
        wave_lds_sync();                                       // (cand is the row's staging buffer: every lane has read its window)
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            const unsigned span = hi - lo;
            unsigned base = 0;
#pragma unroll
            for (int r = 0; r < EPL; r++) {
                const bool in = kh[r] - lo <= span;
                const K mask = __builtin_amdgcn_ballot_w64(in);
                const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, base));
                if (in) cand[pos] = join(kh[r], kl[r]);
                base += (unsigned)__builtin_popcountll(mask);
                if (r % PSS_POST_CBAR == PSS_POST_CBAR - 1) __builtin_amdgcn_sched_barrier(0);   // (sixteen positions and addresses in flight cost 48 registers)
            }
            M = base;
            if (pass == 0 && !(n_lo <= k && k < n_lo + M)) { inside = false; break; }
            if (M <= CAP) break;
            if (pass == 1) return false;
            unsigned n_hi = n_lo + M;
#pragma unroll 1
            while (n_hi - n_lo > CAP && lo < hi) {
                const unsigned trial = lo + ((hi - lo + 1u) >> 1);
                const unsigned c = count_below<EPL, 1>(kh, trial, nullptr, 0, lane, phase);
                if (c <= k) { lo = trial; n_lo = c; } else { hi = trial - 1u; n_hi = c; }
            }
            if (n_hi - n_lo > CAP) return false;               // (more than CAP keys share one high word)
            wave_lds_sync();
        }
        if (!inside) continue;
        const unsigned j = k - n_lo;                           // rank among the candidates
        wave_lds_sync();
        K c[CPL];
#pragma unroll
        for (int i = 0; i < CPL; i++) c[i] = (unsigned)(lane + 64 * i) < M ? cand[lane + 64 * i] : PADK;
        // the search continued on whole keys of the candidates: a = #candidates below lo64, b = #candidates up to hi64
        K lo64 = join(lo, 0u), hi64 = join(hi, 0xffffffffu);
        unsigned a = 0, b = M;
#pragma unroll 1
        while (b - a > 1u && lo64 != hi64) {          // (lo64 <= hi64 throughout; s_cmp_lg_u64 is a scalar instruction, an ordered 64-bit compare is not)
            const K trial = lo64 + ((hi64 - lo64 + 1u) >> 1);
            unsigned cnt = 0;
#pragma unroll
            for (int i = 0; i < CPL; i++) cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c[i] < trial));
            if (cnt <= j) { lo64 = trial; a = cnt; } else { hi64 = trial - 1u; b = cnt; }
        }
        if (b - a == 1u) {
            // one candidate in [lo64, hi64] (the padding lanes hold all ones, above every bracket): rank k
            K got = 0;
            const K w64 = hi64 - lo64;
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                const K mask = __builtin_amdgcn_ballot_w64(c[i] - lo64 <= w64);
                if (mask) got = readlane_k(c[i], (int)__builtin_ctzll(mask));
            }
            v1 = got;
        } else {
            v1 = lo64;                                         // lo64 == hi64: the value holds the ranks a .. b - 1
        }
        if (j + 1u < b) {
            v2 = v1;                                           // several candidates share the value and rank k + 1 is one of them
        } else if (j + 1u < M) {
            K nx = PADK;                                       // the smallest candidate above it
#pragma unroll
            for (int i = 0; i < CPL; i++) nx = (c[i] > v1 && c[i] < nx) ? c[i] : nx;
            v2 = wave_reduce<OpMin>(nx);
        } else {
            // rank k is the bracket's last key: rank k + 1 is the smallest key above the bracket
            unsigned nh = 0xffffffffu;
#pragma unroll
            for (int r = 0; r < EPL; r++) nh = (kh[r] > hi && kh[r] < nh) ? kh[r] : nh;
            nh = wave_reduce<OpMin>(nh);
            unsigned nl = 0xffffffffu;
#pragma unroll
            for (int r = 0; r < EPL; r++) nl = (kh[r] == nh && kl[r] < nl) ? kl[r] : nl;
            v2 = join(nh, wave_reduce<OpMin>(nl));
        }
        return true;
    }
    return false;
}

// The same for float32 rows (32-bit keys; CPL candidates per lane).  mn / mx are not produced here: the caller takes them from v_min_f32 / v_max_f32.
template <int EPL, int CPL>
__device__ __forceinline__ bool select_kth32_compact(const unsigned (&key)[EPL], unsigned k, unsigned &v1, unsigned &v2, unsigned *cand, int lane, float guess)
{
    constexpr unsigned PADK = 0xffffffffu, CAP = 64u * CPL;
    if (!(guess - guess == 0.0f)) return false;               // NaN or infinite mean
    int phase = 0;
#pragma unroll 1
    for (int attempt = 0; attempt < 2; attempt++) {
        const float d = attempt ? 2.0f : 0.5f;
        // (finite values: keys below 0xff800000, so hi + 1 cannot wrap and the padding keys are never inside a bracket)
        unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)f2ord(guess - d));
        unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)f2ord(guess + d));
        unsigned n_lo = count_below<EPL, 1>(key, lo, nullptr, 0, lane, phase);
        unsigned M = 0;
        bool inside = true;
        wave_lds_sync();
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            const unsigned span = hi - lo;
            unsigned base = 0;
#pragma unroll
            for (int r = 0; r < EPL; r++) {
                const bool in = key[r] - lo <= span;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(in);
                const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, base));
                if (in) cand[pos] = key[r];
                base += (unsigned)__builtin_popcountll(mask);
                if (r % PSS_POST_CBAR == PSS_POST_CBAR - 1) __builtin_amdgcn_sched_barrier(0);
            }
            M = base;
            if (pass == 0 && !(n_lo <= k && k < n_lo + M)) { inside = false; break; }
            if (M <= CAP) break;
            if (pass == 1) return false;
            unsigned n_hi = n_lo + M;
#pragma unroll 1
            while (n_hi - n_lo > CAP && lo < hi) {
                const unsigned trial = lo + ((hi - lo + 1u) >> 1);
                const unsigned c = count_below<EPL, 1>(key, trial, nullptr, 0, lane, phase);
                if (c <= k) { lo = trial; n_lo = c; } else { hi = trial - 1u; n_hi = c; }
            }
            if (n_hi - n_lo > CAP) return false;               // (more than CAP equal keys)
            wave_lds_sync();
        }
        if (!inside) continue;
        const unsigned j = k - n_lo;
        wave_lds_sync();
        unsigned c[CPL];
#pragma unroll
        for (int i = 0; i < CPL; i++) c[i] = (unsigned)(lane + 64 * i) < M ? cand[lane + 64 * i] : PADK;
        unsigned a = 0, b = M;
#pragma unroll 1
        while (b - a > 1u && lo != hi) {
            const unsigned trial = lo + ((hi - lo + 1u) >> 1);
            unsigned cnt = 0;
#pragma unroll
            for (int i = 0; i < CPL; i++) cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(c[i] < trial));
            if (cnt <= j) { lo = trial; a = cnt; } else { hi = trial - 1u; b = cnt; }
        }
        if (b - a == 1u) {
            unsigned got = 0;
            const unsigned w = hi - lo;
#pragma unroll
            for (int i = 0; i < CPL; i++) {
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(c[i] - lo <= w);
                if (mask) got = readlane_k(c[i], (int)__builtin_ctzll(mask));
            }
            v1 = got;
        } else {
            v1 = lo;                                           // lo == hi: the value holds the ranks a .. b - 1
        }
        if (j + 1u < b) {
            v2 = v1;
        } else if (j + 1u < M) {
            unsigned nx = PADK;
#pragma unroll
            for (int i = 0; i < CPL; i++) nx = (c[i] > v1 && c[i] < nx) ? c[i] : nx;
            v2 = wave_reduce<OpMin>(nx);
        } else {
            // rank k is the bracket's last key: rank k + 1 is the smallest key above the ORIGINAL candidates' upper end — above v1 will do
            unsigned nx = PADK;
#pragma unroll
            for (int r = 0; r < EPL; r++) nx = (key[r] > v1 && key[r] < nx) ? key[r] : nx;
            v2 = wave_reduce<OpMin>(nx);
        }
        return true;
    }
    return false;
}

// LDS row stride (elements) of a thread's EPL consecutive elements: the stride in bytes is = 16 mod 32, so that the 16 lanes
// a ds_read_b128 / ds_write_b128 services together start 16 bytes apart modulo the 64 banks (conflict-free).
// float: EPL + 4 (or + 8); double: EPL + 2.
template <int EPL, class T = float>
struct PostCfg {
    static constexpr int S = sizeof(T) == 8 ? EPL + 2 : (((EPL + 4) % 8 == 4) ? EPL + 4 : EPL + 8);
    static constexpr int CH = 16 / sizeof(T);     // elements per 16-byte chunk (the unit of the global accesses)
    static constexpr int Q = EPL / CH;            // chunks per thread
    static constexpr int TAILQ = 4 / CH;          // chunks of the next thread's first four elements
};

template <bool WAVE_LOCAL>
__device__ __forceinline__ void row_sync()
{
    if constexpr (WAVE_LOCAL) {  // the row's threads are the lanes of one wavefront, whose LDS instructions execute in order
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// np.interp(np.linspace(0, len-1, W), np.arange(len), row)[x] in float64 (pyspecsdr.py:1379-1383 / :1550-1554)
template <class Row>
__device__ __forceinline__ double interp_at(const Row &row, int len, int W, int x)
{
    const double stop = (double)(len - 1);
    double xp;
    if (W == 1) xp = 0.0;
    else {
        const double step = stop / (double)(W - 1);
        xp = (x == W - 1) ? stop : (double)x * step;
    }
    if (xp >= stop) return (double)row[len - 1];
    const int j = (int)xp;
    // np.interp divides the difference by xp[j + 1] - xp[j]: np.arange's step, exactly 1.0, and x / 1.0 is x for every x — no division here
    // (35 float64-class instructions per value otherwise)
    const double slope = (double)row[j + 1] - (double)row[j];
    return slope * (xp - (double)j) + (double)row[j];
}

// a row in the staging layout (element e at buf[(e / EPL) * S + e % EPL])
template <int EPL, class T>
struct StagedRow {
    const T *buf;
    __device__ __forceinline__ T operator[](int e) const { return buf[(e / EPL) * PostCfg<EPL, T>::S + (e % EPL)]; }
};

// One row, staged in LDS in the per-thread-consecutive layout (element e at buf[(e / EPL) * S + e % EPL]) and already
// synchronised: 5-tap smoothing, median, clamp, the clamped row out through the same staging buffer (coalesced 16-byte
// stores to out4), the row's finite extremes.  slot[j]: LDS slot of the 16-byte chunk j * T + t.  Ends with a row_sync
// (the buffer may be overwritten afterwards).  FULL: the row has exactly T * EPL points (padding = the last thread's last
// four slots only).
// out4 == nullptr: the clamped row is NOT written (the display kernels can rebuild any element of it from the dB row and the clamp
// threshold: k_disp_rows<..., FROM_DB>); row_thr (nullable) receives the row's clamp threshold (float32(median - 10); float64 rows: median - 10).
// vals (nullable, with disp_w): the row resampled to the display width — np.interp(np.linspace(0, m - 1, disp_w), np.arange(m), row), float64,
// what draw_waterfall / draw_persistence normalise and quantise (pyspecsdr.py:1379-1383, :1550-1554) — so that the display kernel of a batch
// reads disp_w values per row instead of the row (k_disp_vals_win).
// TR = float: the float32 rows of pss_spectrum_db (sums in float64, rounded once); TR = double: the reference's own row type throughout
// (np.median of a row with a NaN is NaN and clamps nothing).
template <int EPL, int W, bool FULL, class TR>
__device__ __forceinline__ void post_row_staged(TR *buf, const int (&slot)[PostCfg<EPL, TR>::Q], int t, int m, unsigned long long *red, int wave,
                                                int lane, int &phase, float4 *out4, TR *row_lo, TR *row_hi, long f, TR *row_thr = nullptr,
                                                double *vals = nullptr, int disp_w = 0)
{
    using O = Ord<TR>;
    using K = typename O::K;
    using PC = PostCfg<EPL, TR>;
    constexpr int T = 64 * W, S = PC::S, Q = PC::Q, CH = PC::CH;
    constexpr int PAD_FROM = FULL ? EPL - 4 : 0;
    constexpr bool F64 = sizeof(TR) == 8;
    const int mq = m / CH;                       // whole 16-byte chunks of the output row (m % CH == 0: m = N - 4, N % 4 == 0)
    const int nv = m - t * EPL;                  // this thread's slots r < nv hold elements of the smoothed row
    // EPL consecutive elements + the next thread's first four (the 5-tap window of the last four outputs)
    K key[EPL];                                  // (float64 rows: the two words of a key live in kh / kl below; `key` is then unused)
    unsigned kh[F64 ? EPL : 1], kl[F64 ? EPL : 1];
    float lsum = 0.0f;
    double dmin = INFINITY, dmax = -INFINITY;    // (float64 rows of one wavefront: this lane's smoothed extremes, for select_kth64_compact)
    float fmin32 = INFINITY, fmax32 = -INFINITY; // (float32 rows: the same)
    {
        double xd[EPL + 4];
#pragma unroll
        for (int i = 0; i < Q + PC::TAILQ; i++) {
            const float4 v = *reinterpret_cast<const float4 *>(buf + (i < Q ? t * S + CH * i : (t + 1) * S + CH * (i - Q)));
            if constexpr (F64) {
                xd[2 * i] = __hiloint2double(__float_as_int(v.y), __float_as_int(v.x));
                xd[2 * i + 1] = __hiloint2double(__float_as_int(v.w), __float_as_int(v.z));
            } else {
                xd[4 * i] = (double)v.x; xd[4 * i + 1] = (double)v.y; xd[4 * i + 2] = (double)v.z; xd[4 * i + 3] = (double)v.w;
            }
        }
        // np.convolve(fd, ones(5)/5, 'valid') in float64 (pyspecsdr.py:2279); the row is kept as the order-preserving
        // integer image of its value (float32 rows: of the float32 rounding of the sum — 4 bytes per element instead of 12)
#pragma unroll
        for (int r = 0; r < EPL; r++) {
            double acc = 0.0;
#if defined(PSS_EXP_POST_KNOCK) && (PSS_EXP_POST_KNOCK & 4)
            acc = xd[r];
#else
#pragma unroll
            for (int k = 0; k < 5; k++) acc += xd[r + k] * 0.2;
#endif
            const TR sm = (TR)acc;
            // padding sorts above everything
            const bool pad = FULL ? (r >= PAD_FROM && t == T - 1) : r >= nv;
            const K kk0 = pad ? O::PAD : O::enc(sm);
            if constexpr (F64) { kh[r] = (unsigned)(kk0 >> 32); kl[r] = (unsigned)kk0; }
            else key[r] = kk0;
            lsum += pad ? 0.0f : (float)sm;
            if constexpr (!F64 && W == 1) {
                if (FULL && r < PAD_FROM) { fmin32 = vmin_f32(fmin32, (float)sm); fmax32 = vmax_f32(fmax32, (float)sm); }
                else { fmin32 = vmin_f32(fmin32, pad ? INFINITY : (float)sm); fmax32 = vmax_f32(fmax32, pad ? -INFINITY : (float)sm); }
            }
            if constexpr (F64 && W == 1) {
                if (FULL && r < PAD_FROM) { dmin = vmin_f64(dmin, (double)sm); dmax = vmax_f64(dmax, (double)sm); }
                else { dmin = vmin_f64(dmin, pad ? (double)INFINITY : (double)sm); dmax = vmax_f64(dmax, pad ? -(double)INFINITY : (double)sm); }
            }
        }
    }
    // the row's mean: where the median search starts looking (select_kth; a hint only, never part of a result)
    const float guess = __uint_as_float(row_reduce<OpFAdd, W>(__float_as_uint(lsum), red, wave, lane, phase)) / (float)m;
    // np.median: the middle order statistic, or the mean of the two middle ones
    const unsigned k1 = (unsigned)((m - 1) >> 1);
    K mn, mx, v1, v2;
    // PSS_EXP_POST_KNOCK: timing experiments (results wrong) — bit 0: no order statistics, bit 1: no resampled row, bit 2: no smoothing (R6-11)
#if defined(PSS_EXP_POST_KNOCK) && (PSS_EXP_POST_KNOCK & 1)
    if constexpr (F64) {
        v1 = v2 = O::enc((TR)guess); mn = O::enc((TR)(guess - 20.0f)); mx = O::enc((TR)(guess + 20.0f));
#pragma unroll
        for (int r = 0; r < EPL; r++) key[r] = ((K)kh[r] << 32) | kl[r];
    } else
#endif
    if constexpr (F64) {
        bool done = false;
#ifndef PSS_POST_NO_COMPACT
        if constexpr (W == 1) {
            // (the row's extreme keys first: two wave-uniform results instead of two per-lane doubles kept across the search)
            mn = wave_extreme_key_f64<true>(dmin);
            mx = wave_extreme_key_f64<false>(dmax);
            done = select_kth64_compact<EPL, PSS_POST_CPL>(kh, kl, k1, v1, v2, reinterpret_cast<K *>(buf), lane, guess);
        }
#endif
        if (!done) select_kth64<EPL, W, PAD_FROM>(kh, kl, k1, (unsigned)m, v1, v2, mn, mx, red, wave, lane, phase, guess);
#pragma unroll
        for (int r = 0; r < EPL; r++) key[r] = ((K)kh[r] << 32) | kl[r];     // (register pairs: no instruction)
    } else {
        bool done = false;
#ifndef PSS_POST_NO_COMPACT
        if constexpr (W == 1) {
            mn = (K)__builtin_amdgcn_readfirstlane((int)f2ord(__uint_as_float(wave_reduce<OpFMin>(__float_as_uint(fmin32)))));
            mx = (K)__builtin_amdgcn_readfirstlane((int)f2ord(__uint_as_float(wave_reduce<OpFMax>(__float_as_uint(fmax32)))));
            done = select_kth32_compact<EPL, PSS_POST_CPL>(key, k1, v1, v2, reinterpret_cast<unsigned *>(buf), lane, guess);
        }
#endif
        if (!done) {
            unsigned below, upto;
            v1 = select_kth<EPL, W, PAD_FROM, K>(key, k1, (unsigned)m, v2, mn, mx, below, upto, red, wave, lane, phase, guess == guess,
                                                 [&](float d) { return f2ord(guess + d); });
        }
    }
    const double med = (m & 1) ? (double)O::dec(v1) : 0.5 * ((double)O::dec(v1) + (double)O::dec(v2));
    // fd[fd < thr] = thr (:2282-2283).  float32(max(s, thr)) = max(float32(s), float32(thr)) (rounding is monotonic), and
    // the maximum of two floats is the maximum of their ordered images
    K thr = O::enc((TR)(med - 10.0));
    if constexpr (F64) {
        // np.median of a row that holds a NaN is NaN, and `fd < NaN` selects nothing: the lowest image clamps nothing (and survives the
        // round trip through row_thr: enc(dec(0)) = 0).  A row holds a NaN iff its largest key lies above +inf's image (a NaN with the sign
        // bit clear) or its smallest below -inf's (sign bit set): the row's extreme keys are known to every thread (no per-element test).
        const bool has_nan = mx > O::POS_INF || mn < O::NEG_INF;
        if (has_nan || !(med == med)) thr = 0;
    }
    if (row_thr && t == 0) row_thr[f] = O::dec(thr);
    row_sync<W == 1>();                      // every thread has read its input window
    if (out4 || vals) {
#pragma unroll
        for (int i = 0; i < Q; i++) {
            TR o[CH];
#pragma unroll
            for (int k = 0; k < CH; k++) {
                const K kk = key[CH * i + k];
                o[k] = O::dec(kk > thr ? kk : thr);
            }
            if constexpr (F64) *reinterpret_cast<double2 *>(buf + t * S + CH * i) = make_double2(o[0], o[1]);
            else *reinterpret_cast<float4 *>(buf + t * S + CH * i) = make_float4(o[0], o[1], o[2], o[3]);
        }
        row_sync<W == 1>();
        if (out4) {
#pragma unroll
            for (int j = 0; j < Q; j++) {
                const int c = j * T + t;
                if (c < mq) out4[c] = *reinterpret_cast<const float4 *>(buf + slot[j]);
            }
        }
#if defined(PSS_EXP_POST_KNOCK) && (PSS_EXP_POST_KNOCK & 2)
        if (false) {
#else
        if (vals) {
#endif
            const StagedRow<EPL, TR> row{buf};
            for (int x = t; x < disp_w; x += T) vals[(size_t)f * disp_w + x] = interp_at(row, m, disp_w, x);
        }
    }
    if (row_lo) {
        // finite extremes of the clamped row, as the accumulators' np.isfinite masks see them.  A row whose smoothed
        // values are all finite (any real dB row): min / max commute with the clamp.
        K a, b;
        if (mn > O::NEG_INF && mx < O::POS_INF) {
            a = mn > thr ? mn : thr;
            b = mx > thr ? mx : thr;
        } else {
            a = O::PAD; b = 0;
#pragma unroll
            for (int r = 0; r < EPL; r++) {
                const K kk = key[r] > thr ? key[r] : thr;
                const bool fin = key[r] != O::PAD && kk > O::NEG_INF && kk < O::POS_INF;
                a = (fin && kk < a) ? kk : a;
                b = (fin && kk > b) ? kk : b;
            }
            a = row_reduce<OpMin, W>(a, red, wave, lane, phase);
            b = row_reduce<OpMax, W>(b, red, wave, lane, phase);
            if (a > b) { a = O::enc((TR)INFINITY); b = O::enc((TR)-INFINITY); }  // no finite value: the neutral pair
        }
        if (t == 0) { row_lo[f] = O::dec(a); row_hi[f] = O::dec(b); }
    }
    row_sync<W == 1>();                      // the staging buffer may be overwritten now
}

// Requires N % 4 == 0 (16-byte aligned rows) and N - 4 <= 64 * W * EPL.  FULL: N == 64 * W * EPL exactly (the power-of-two
// read buffers), where the only padding is the last thread's last four slots.
template <int EPL, int W, bool FULL, class TR = float>
__global__ __launch_bounds__(W == 1 ? 256 : 64 * W) void k_post_sel(const TR *__restrict__ db, TR *__restrict__ post, int N,
                                                                     long n_frames, TR *__restrict__ row_lo,
                                                                     TR *__restrict__ row_hi, TR *__restrict__ row_thr,
                                                                     double *__restrict__ vals, int disp_w)
{
    using PC = PostCfg<EPL, TR>;
    constexpr int T = 64 * W;                    // threads per row
    constexpr int RPW = W == 1 ? 4 : 1;          // rows per workgroup
    constexpr int S = PC::S, Q = PC::Q, CH = PC::CH;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ unsigned long long red[2 * (W > 1 ? W : 1)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = W == 1 ? lane : tid;           // thread inside the row
    TR *buf = reinterpret_cast<TR *>(smem) + (W == 1 ? wave : 0) * (T + 1) * S;
    const int m = N - 4, nq = N / CH;
    int phase = 0;
    // LDS slot of the 16-byte chunk c = j * T + t (the unit of the coalesced global accesses)
    int slot[Q];
#pragma unroll
    for (int j = 0; j < Q; j++) {
        const int e0 = CH * (j * T + t);
        slot[j] = (e0 / EPL) * S + (e0 % EPL);
    }
    const long groups = (n_frames + RPW - 1) / RPW;
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long f = g * RPW + (W == 1 ? wave : 0);
        if (W == 1 && f >= n_frames) continue;   // whole wavefront (rows are wavefront-private when W = 1)
        const float4 *row4 = reinterpret_cast<const float4 *>(db + (size_t)f * N);
        float4 q[Q];
#pragma unroll
        for (int j = 0; j < Q; j++) {
            const int c = j * T + t;
            q[j] = row4[(FULL || c < nq) ? c : 0];   // chunks past the row re-read chunk 0 (their elements are never used)
        }
#pragma unroll
        for (int j = 0; j < Q; j++) *reinterpret_cast<float4 *>(buf + slot[j]) = q[j];
        // rows of more than T * EPL points (N - 4 <= T * EPL < N): the last thread's window reaches into four more elements
        if (!FULL && t < PC::TAILQ && T * Q + t < nq) *reinterpret_cast<float4 *>(buf + T * S + CH * t) = row4[T * Q + t];
        row_sync<W == 1>();
        post_row_staged<EPL, W, FULL, TR>(buf, slot, t, m, red, wave, lane, phase,
                                          post ? reinterpret_cast<float4 *>(post + (size_t)f * m) : nullptr, row_lo, row_hi, f, row_thr, vals,
                                          disp_w);
    }
}

// ---- batched display accumulators -----------------------------------------------------------------------------------
// The reference keeps the last 30 (waterfall, pyspecsdr.py:130-131,1351-1353) / 10 (persistence, :151-152,1521-1523)
// post-processed rows and, for every new frame, normalises with the finite minimum / maximum over that history
// (:1356-1358, :1525-1530).  For a batch of frames that is a sliding-window extreme over per-row extremes:
// win_lo[i] = min(row_lo[i - window + 1 .. i]); rows before the batch are supplied as a halo of n_halo (lo, hi) pairs in
// front of the arrays (from the previous batch, or from the left neighbour when the batch is sharded over GPUs: 8 bytes
// per row instead of the rows themselves).

// finite minimum / maximum of every row (np.min / np.max over all_data[np.isfinite(all_data)], one row's share);
// a row without a finite value yields (+inf, -inf), the neutral pair.  One wavefront per row.
template <class T>
__global__ __launch_bounds__(256) void k_row_extremes(const T *__restrict__ rows, long n_rows, int len, T *__restrict__ row_lo,
                                                      T *__restrict__ row_hi)
{
    const int lane = threadIdx.x & 63;
    const long wpb = blockDim.x >> 6;
    for (long f = (long)blockIdx.x * wpb + (threadIdx.x >> 6); f < n_rows; f += (long)gridDim.x * wpb) {
        const T *row = rows + (size_t)f * len;
        T lo = (T)INFINITY, hi = (T)-INFINITY;
        for (int i = lane; i < len; i += 64) {
            const T v = row[i];
            if (isfinite(v)) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const T a = __shfl_xor(lo, off), b = __shfl_xor(hi, off);
            lo = a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
        if (lane == 0) { row_lo[f] = lo; row_hi[f] = hi; }
    }
}

template <class T>
__global__ __launch_bounds__(256) void k_slide_extremes(const T *__restrict__ row_lo, const T *__restrict__ row_hi, long n_frames,
                                                        int n_halo, int window, double *__restrict__ win_lo,
                                                        double *__restrict__ win_hi)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_frames; i += (long)gridDim.x * blockDim.x) {
        const long p = i + n_halo;               // position in the halo-prefixed arrays
        long first = p - (window - 1);
        if (first < 0) first = 0;
        double lo = INFINITY, hi = -INFINITY;
        for (long q = first; q <= p; q++) {
            const double a = (double)row_lo[q], b = (double)row_hi[q];
            lo = a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
        win_lo[i] = lo;
        win_hi[i] = hi;
    }
}

// element j of the post-processed row rebuilt from the dB row and the row's clamp threshold, bit for bit what k_post_sel
// stores: (float32 rows: the float32 rounding of) sum_k (double)db[j + k] * 0.2 in np.convolve's order, then the maximum with the
// threshold on the ordered images
template <class T>
struct PostFromDb {
    const T *db;                  // the frame's dB row (n_fft = len + 4 points)
    typename Ord<T>::K thr;       // ordered image of the clamp threshold
    __device__ __forceinline__ T operator[](int j) const
    {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 5; k++) acc += (double)db[j + k] * 0.2;
        const typename Ord<T>::K kk = Ord<T>::enc((T)acc);
        return Ord<T>::dec(kk > thr ? kk : thr);
    }
};

// one display cell from the resampled value v and the window extremes:
//   MODE 0  waterfall (pyspecsdr.py:1386-1403): glyph 0 '.', 1 '-', 2 '=', 3 '#'; colour int(norm * 5); -1: not finite.
//           No zero-range guard (:1389), as in the reference.
//   MODE 1  persistence (:1556-1563): y = int((1 - norm) * (disp_h - 1)) of the newest trace, -1 if not drawn (not finite or
//           outside the grid); range 0 -> 1 (:1528-1530).
template <int MODE>
__device__ __forceinline__ void quantise_cell(double v, double lo, double hi, int disp_h, int8_t &a, int8_t &b)
{
    if (MODE == 0) {
        int8_t g = -1, ci = -1;
        if (isfinite(v)) {
            const double nv = (v - lo) / (hi - lo);
            ci = (int8_t)(int)(nv * 5);
            g = nv > 0.75 ? 3 : nv > 0.5 ? 2 : nv > 0.25 ? 1 : 0;
        }
        a = g;
        b = ci;
    } else {
        int8_t y8 = -1;
        if (isfinite(v)) {
            double range = hi - lo;
            if (range == 0) range = 1;
            const int y = (int)((1 - (v - lo) / range) * (disp_h - 1));
            if (y >= 0 && y < disp_h) y8 = (int8_t)y;
        }
        a = y8;
    }
}

// One display line per frame: the NEWEST row of the history as the reference draws it at frame i (waterfall line y = 0).
// FROM_DB: `post` holds the dB rows (len + 4 points each) and row_thr the clamp thresholds: the post-processed rows were
// never written (the line needs two neighbouring elements of it per cell: 10 dB values).
template <class T, int MODE, bool FROM_DB = false>
__global__ __launch_bounds__(256) void k_disp_rows(const T *__restrict__ post, const double *__restrict__ win_lo,
                                                   const double *__restrict__ win_hi, long n_frames, int len, int disp_w,
                                                   int disp_h, int8_t *__restrict__ out_a, int8_t *__restrict__ out_b,
                                                   const T *__restrict__ row_thr = nullptr)
{
    const long total = n_frames * disp_w;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (long)gridDim.x * blockDim.x) {
        const long f = c / disp_w;
        const int x = (int)(c - f * disp_w);
        const double lo = win_lo[f], hi = win_hi[f];
        double v;
        if constexpr (FROM_DB) {
            const PostFromDb<T> row{post + (size_t)f * (len + 4), Ord<T>::enc(row_thr[f])};
            v = interp_at(row, len, disp_w, x);
        } else {
            v = interp_at(post + (size_t)f * len, len, disp_w, x);
        }
        int8_t a, b = 0;
        quantise_cell<MODE>(v, lo, hi, disp_h, a, b);
        out_a[c] = a;
        if (MODE == 0) out_b[c] = b;
    }
}

// The display lines of a batch from the rows already resampled to the display width (k_post_sel's `vals`: disp_w float64 values per frame instead
// of the row) AND the sliding-window extremes they are normalised with, in one launch (one kernel and one launch gap less at the end of the chain): a
// workgroup takes RPB consecutive frames, its first RPB threads form their frames' window extremes (the same loop as k_slide_extremes: <= window
// (lo, hi) pairs each, L2-resident), then all threads quantise the RPB x disp_w cells; the window extremes never reach memory.
template <int MODE, class T, int RPB>
__global__ __launch_bounds__(256) void k_disp_vals_win(const double *__restrict__ vals, const T *__restrict__ row_lo, const T *__restrict__ row_hi,
                                                       long n_frames, int n_halo, int window, int disp_w, int disp_h, int8_t *__restrict__ out_a,
                                                       int8_t *__restrict__ out_b)
{
    __shared__ double s_lo[RPB], s_hi[RPB];
    const long groups = (n_frames + RPB - 1) / RPB;
    for (long g = blockIdx.x; g < groups; g += gridDim.x) {
        const long f0 = g * RPB;
        const int rows = (int)(n_frames - f0 < RPB ? n_frames - f0 : RPB);
        if ((int)threadIdx.x < rows) {
            const long i = f0 + threadIdx.x, p = i + n_halo;
            long first = p - (window - 1);
            if (first < 0) first = 0;
            double lo = INFINITY, hi = -INFINITY;
            for (long q = first; q <= p; q++) {
                const double a = (double)row_lo[q], b = (double)row_hi[q];
                lo = a < lo ? a : lo;
                hi = b > hi ? b : hi;
            }
            s_lo[threadIdx.x] = lo;
            s_hi[threadIdx.x] = hi;
        }
        __syncthreads();
        const int cells = rows * disp_w;
        const size_t c0 = (size_t)f0 * disp_w;
        for (int c = threadIdx.x; c < cells; c += blockDim.x) {
            const int fr = c / disp_w;
            int8_t a, b = 0;
            quantise_cell<MODE>(vals[c0 + c], s_lo[fr], s_hi[fr], disp_h, a, b);
            out_a[c0 + c] = a;
            if (MODE == 0) out_b[c0 + c] = b;
        }
        __syncthreads();
    }
}

}  // namespace pss_post
