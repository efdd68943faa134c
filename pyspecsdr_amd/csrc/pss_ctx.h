// pss_ctx.h — the context object behind the C ABI (include/pss.h): device, stream, cached tables, scratch.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/pss.h"

struct PssNfmFilt {
    double taps[65];
    double sos[24];
    double zi[8];
    double *d_rev = nullptr;   // device copy of the reversed taps (rev[j] = taps[64 - j], 65 doubles + 15 zeros): the fused NFM kernel's
                               // hand-scheduled FIR fetches its taps from here with scalar loads; made on first use, owned by the context
};

struct PssWfmFilt {
    double lp[18], pilot[30], lmr[30];  // butter(5) SOS rows: low-pass 15 kHz, pilot band-pass, L-R band-pass
    double alpha;                       // de-emphasis pole exp(-1/(75e-6 fs))
};

struct PssPairwisePlan {  // numpy pairwise-sum tree for one frame length (see pss_demod.hip)
    int n_leaves = 0, n_nodes = 0, n_levels = 0, n_roots = 0;
    int wave_tree = 0;  // 64 (leaf, accumulator) pairs and a perfectly balanced tree over equal leaves: one wavefront can fold it with xor shuffles
    int *d_leaf_off = nullptr, *d_leaf_len = nullptr, *d_node_l = nullptr, *d_node_r = nullptr, *d_level_start = nullptr;
    int *d_roots = nullptr;
};

struct pss_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream2 = nullptr;  // side stream: the spectrum kernel of pss_spectrum_nfm runs beside the demodulator
    hipStream_t cur = nullptr;      // stream the next launches go to (nullptr = `stream`)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;   // pss_order_after / pss_order_before (created on first use)
    int n_cus = 0;                                  // hipDeviceProp_t::multiProcessorCount
    bool iq_c128 = false;           // set by pss_demod_ssb_c128 around pss_demod: d_iq points at complex128 frames (the SSB kernels' float64 loader)
    int fuse_post = 1;              // option "fuse_post" (float64-row pipelines, 1024-point frames): 1 = transform + post-process in ONE kernel (pss_spec_post.h), 0 = two kernels (A/B reference)
    double target_rate = 22050.0;   // demodulate_nfm / _wfm's target_rate (pss_set_target_rate): decimation factor int(fs / target_rate)
    bool wfm_correct = false;       // pss_demod(WFM) is handed the frames AS READ and applies iq_correction itself (consumed there; set by pss_demod_signal / pss_frame_pipeline)
    const float *wfm_scal = nullptr;   // ... the per-frame correction scalars for the fused forward kernel
    bool wfm_corr_copy = false;     // option "wfm_corr_copy": always materialise the corrected copy of the batch (the round-4 path; A/B reference)
    float *power_out = nullptr;     // pss_demod_power -> pss_demod(AM): the power dB of the same frames from the mean pass (consumed there)
    bool fork_after_fwd = false;
    bool did_fork = false;
    bool defer_bwd = false;              // the fused NFM path launches only its forward kernel and parks the backward launch here:
    std::function<int()> pending_bwd;    // pss_frame_pipeline_nfm places it beside the post-process  // set by pss_demod when it recorded ev_fork (fused NFM path only)
    std::string err;
    std::map<int, double2 *> tw;       // exp(-2 pi i k / N), k < N
    std::map<int, double *> win;       // np.hamming(N)
    std::map<int, double2 *> tw_pf;    // exp(+2 pi i k / N) as pocketfft tabulates it (pss_hilbert_pf.h), k < N
    struct Bluestein { double2 *d_chirp; double2 *d_B; double *d_win; int M; };
    std::map<int, Bluestein> bs;       // per frame length that is not a power of two (pss_fft.hip)
    std::map<double, PssNfmFilt> nfm;  // per sample rate
    std::map<double, std::array<double, 65>> ssb;
    std::map<double, PssWfmFilt> wfm;
    std::map<int, PssPairwisePlan> plans;
    void *scratch = nullptr;       // demodulator scratch (main stream)
    size_t scratch_bytes = 0;
    void *scratch_iqc = nullptr;   // IQ-corrected frames for the WFM dispatcher path
    void *scratch_scan = nullptr;  // scanner dB rows when the caller asks only for the per-slice numbers
    size_t scratch_scan_bytes = 0;
    void *scratch_pk = nullptr;
    size_t scratch_pk_bytes = 0;
    void *prog = nullptr;          // k_nfm_fwd: per-CU progress words of its workgroups (priority balancing)
    unsigned prog_epoch = 0;       // launch counter stamped into those words (16 bits, never 0)
    void *scratch_post = nullptr;  // pss_frame_pipeline_nfm without materialised post-processed rows: the rows' clamp thresholds
    size_t scratch_post_bytes = 0;
    void *scratch_db64 = nullptr;  // pss_frame_pipeline_cells / pss_spectrum_cells with float32 rows only, on lengths the fused kernel does not serve: the float64 rows
    size_t scratch_db64_bytes = 0;
    void *scratch_win = nullptr;   // sliding-window extremes of the batched display accumulators
    size_t scratch_win_bytes = 0;
    float *d_hann = nullptr;       // pss_classify: scipy's periodic Hann window (1024, float32) and sum(win * win)
    float hann_sum = 0.0f;
    // streaming entry points (pss_h_stream_*): copy streams, events and the two device buffer sets, kept across calls
    // (creating and freeing them per capture cost ~2 ms of a 17 ms capture)
    hipStream_t st_up = nullptr, st_dn = nullptr;
    hipEvent_t st_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // up_done[2], cmp_done[2], dn_done[2]
    void *st_buf[18] = {};         // [0..11]: the chunk buffer sets; [12..15]: per-chunk display grids (two sets x two planes); [16..17]: the last `window` post-processed rows (ping-pong)
    size_t st_cap[18] = {};
    float *d_hann_short = nullptr;  // the same for reads shorter than 1024 samples (window length = read length hann_short_n)
    float hann_short_sum = 0.0f;
    int hann_short_n = 0;
    size_t scratch_iqc_bytes = 0;
    void *scratch_fft = nullptr;   // spectrum scratch: separate, the spectrum kernel may run on the side stream
    size_t scratch_fft_bytes = 0;
    void *scratch_hil = nullptr;   // hilbert() of rows longer than 16384 samples (spectrum Z + pre-pass scratch): the demodulator's, never shared with the side stream's spectrum
    size_t scratch_hil_bytes = 0;
    void *stage = nullptr;         // device staging of the host-buffer convenience calls (grow-only)
    size_t stage_bytes = 0;
    bool ssb_unfused = false;      // (-DPSS_VARIANTS builds) demodulate_ssb at 8192 / 16384 samples as k_ssb_fir + k_hilbert_xl (the round-2 shape)
    bool fft_two_per_wg = false;   // (-DPSS_VARIANTS builds) N = 2048 spectra with two frames per 256-thread workgroup (the round-2 shape)
    bool no_wfm_fused = false;  // option "wfm_fused" = 0: k_wfm_front + k_nfm_iir path (A/B testing)
    long small_batch_max = 8192;  // option "small_batch_max": largest frame count that takes the small-batch path (measured crossover ~12000)
    long wfm_small_batch_max = 6000;  // option "wfm_small_batch_max" (crossover with the fused kernels, 1024-sample frames: ~12000 frames in round 2, ~6000 since the small-batch path is one array: 0.58 / 1.13 ms at 4096 / 8192 frames against 0.83)
    bool hilbert_exact = false;   // option "hilbert_exact": scipy.signal.hilbert bit for bit (pocketfft's butterfly order, pss_hilbert_pf.h) for rows of 256..1048576 samples
    bool ssb_rfft = true;      // option "ssb_rfft": frames of 8192 / 16384 samples evaluate hilbert() as a real transform pair (k_ssb_rfft); 0: two full-length complex transforms (k_ssb_hilbert_xl)
    bool ssb_hilbert = true;   // option "ssb_hilbert": run the reference's hilbert() round trip inside demodulate_ssb where a register transform exists for the frame length
    bool db_exact = false;         // true: compute_fft's dB rows evaluated to float64 accuracy and rounded once (= float32 of the reference's float64 rows); false: float32 evaluation, 1-2 ulp off, 15-25 % faster kernels
    bool f64_plain = false;        // option "f64_plain": the float64-row entry points (pss_spectrum_db_f64, pss_spectrum_post_f64, pss_frame_pipeline_nfm_f64) on the plain round-3 kernels (generic LDS transform with hypot / log10, radix select) instead of the register kernels: A/B reference
    bool scan_exact = true;        // scanner slices: NumPy's float32 chain bit for bit (scan_db_np); false: the float64 / hardware-log2 dB of compute_fft
    int fft_xl4096 = -1;           // N = 4096 on the component-wise-exchange kernel (pss_fft_xl.h, R4 = 1) instead of k_spectrum_r16<4>; -1 = scanner slices only
    bool post_legacy = false;  // option "post_legacy": LDS bitonic sort / LDS-histogram radix select instead of the register select
    int post_sort_max = 8192;  // option "post_sort_max": longest row that takes the LDS bitonic sort, else radix select (measured crossover 8192..16384)
    bool fft_big_scratch = false;  // option "fft_big_scratch": N = 8192 / 16384 on the scratch-based radix-R pre-pass kernel (A/B reference)
    int fft_prefetch = -1;  // option "fft_prefetch": request the next frame's samples before transforming the current one; -1 = automatic
    int fft_split = -1;  // option "fft_split": component-wise LDS exchanges in k_spectrum_r16; -1 = automatic (N = 256 only)
    bool no_small_batch = false;  // option "small_batch" = 0: never take the systolic small-batch NFM path (A/B testing)
    bool no_fused = false;  // PSS_NO_FUSED=1: use the three-kernel NFM path (A/B and fallback testing)
    void *comm = nullptr;          // pss_comm_init: the RCCL communicator (ncclComm_t) of this context's rank; collectives run on the context's stream
    int comm_rank = 0, comm_n = 1;
    bool timing = false;
    std::string tfilter;  // pss_timing_filter: only launches of this kernel get events (and no per-call events); empty = all
    bool kskip = false;
    int tdepth = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float last_ms = -1.0f;
    // per-kernel timing of the most recent outermost call (timing mode only)
    struct KRec { const char *name; hipEvent_t e0, e1; };
    std::vector<KRec> krecs;   // event pool (grown on demand, reused)
    int kused = 0;
};

// Every entry point works on ITS context's device, whatever the calling thread's current device is, and leaves the
// caller's current device as it found it (HIP's current device is per thread and defaults to 0).
struct PssDevGuard {
    int prev = -1;
    bool switched = false;
    explicit PssDevGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~PssDevGuard()
    {
        if (switched) hipSetDevice(prev);
    }
    PssDevGuard(const PssDevGuard &) = delete;
    PssDevGuard &operator=(const PssDevGuard &) = delete;
};
#define PSS_GUARD(ctx) PssDevGuard _pss_dev_guard((ctx)->device)

// Temporarily override a context switch (restored on scope exit, also on early returns)
template <class T>
struct PssScoped {
    T &ref;
    T keep;
    PssScoped(T &r, T v) : ref(r), keep(r) { r = v; }
    ~PssScoped() { ref = keep; }
    PssScoped(const PssScoped &) = delete;
    PssScoped &operator=(const PssScoped &) = delete;
};
using PssFlagScope = PssScoped<bool>;
using PssStreamScope = PssScoped<hipStream_t>;

int pss_fail(pss_ctx *ctx, int code, const std::string &msg);
int pss_hip_check(pss_ctx *ctx, hipError_t e, const char *what);
int pss_ensure_scratch(pss_ctx *ctx, size_t bytes);
int pss_ensure_buffer(pss_ctx *ctx, void **buf, size_t *cap, size_t bytes, const char *what);
void pss_time_begin(pss_ctx *ctx);
void pss_time_end(pss_ctx *ctx);
// bracket ONE kernel launch with events on the context's stream (no-ops unless timing is enabled)
void pss_kernel_begin(pss_ctx *ctx, const char *name);
void pss_kernel_end(pss_ctx *ctx);

#define PSS_STREAM(ctx) ((ctx)->cur ? (ctx)->cur : (ctx)->stream)

#define PSS_HIP(ctx, call)                                         \
    do {                                                           \
        int _r = pss_hip_check((ctx), (call), #call);              \
        if (_r) return _r;                                         \
    } while (0)

// implemented in pss_fft.hip
bool pss_hilbert_supported(int n);
int pss_hilbert_rows(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_out, int out_mode, unsigned long long *d_maxbits,
                     int16_t *d_pcm);
int pss_fft_tables(pss_ctx *ctx, int n, const double2 **tw, const double **win);
// pss_frame_pipeline's display chain behind the dB rows (rows of either type): thresholds + extremes + the rows resampled to the display
// width in one pass (no post-processed rows in memory), sliding extremes, the line of every frame.  d_vals: n_frames x disp_w doubles.
bool pss_post_sel_serves(const pss_ctx *ctx, int n_fft, bool f64);
bool pss_spec_post_serves(const pss_ctx *ctx, int n_fft);
int pss_spec_post_chain(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, float *d_db32, double *d_db64, double *d_lo, double *d_hi, int n_halo,
                        int window, int display, int disp_h, int disp_w, int8_t *d_a, int8_t *d_b, double *d_vals);
int pss_chain_vals_f32(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_lo, float *d_hi, int n_halo, int window, int display,
                       int disp_h, int disp_w, int8_t *d_a, int8_t *d_b, double *d_vals);
int pss_chain_vals_f64(pss_ctx *ctx, const double *d_db, long n_frames, int n_fft, double *d_lo, double *d_hi, int n_halo, int window, int display,
                       int disp_h, int disp_w, int8_t *d_a, int8_t *d_b, double *d_vals);
bool pss_ssb_fused_supported(int n);
// NumPy's float64 tan / exp under its AVX512_SKX dispatch (SVML's _ha routines restated, pss_design.cpp)
double pss_np_tan(double x);
double pss_np_exp(double x);
int pss_ssb_hilbert_fused(pss_ctx *ctx, const float *d_iq, long n_rows, int n, const double *taps65, double *d_audio, int16_t *d_pcm);
