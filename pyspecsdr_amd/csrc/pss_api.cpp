// pss_api.cpp — context management and the host-buffer convenience calls of the C ABI (include/pss.h).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pss_ctx.h"

static std::string g_create_err;

int pss_fail(pss_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    else g_create_err = msg;
    return code;
}

int pss_hip_check(pss_ctx *ctx, hipError_t e, const char *what)
{
    if (e == hipSuccess) return PSS_OK;
    return pss_fail(ctx, PSS_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

int pss_ensure_buffer(pss_ctx *ctx, void **buf, size_t *cap, size_t bytes, const char *what)
{
    if (bytes <= *cap) return PSS_OK;
    if (*buf) {
        // the old buffer may still be in use by work queued on either stream
        hipStreamSynchronize(ctx->stream);
        if (ctx->stream2) hipStreamSynchronize(ctx->stream2);
        hipFree(*buf);
        *buf = nullptr;
        *cap = 0;
    }
    hipError_t e = hipMalloc(buf, bytes);
    if (e != hipSuccess) return pss_fail(ctx, PSS_E_NOMEM, std::string(what) + " hipMalloc: " + hipGetErrorString(e));
    *cap = bytes;
    return PSS_OK;
}

int pss_ensure_scratch(pss_ctx *ctx, size_t bytes)
{
    return pss_ensure_buffer(ctx, &ctx->scratch, &ctx->scratch_bytes, bytes, "scratch");
}

void pss_time_begin(pss_ctx *ctx)
{
    if (ctx->timing && ctx->tdepth++ == 0 && ctx->tfilter.empty()) hipEventRecord(ctx->ev0, ctx->stream);
}
void pss_kernel_begin(pss_ctx *ctx, const char *name)
{
    if (!ctx->timing) return;
    ctx->kskip = !ctx->tfilter.empty() && ctx->tfilter != name;
    if (ctx->kskip) return;
    if (ctx->kused == (int)ctx->krecs.size()) {
        pss_ctx::KRec r{name, nullptr, nullptr};
        hipEventCreate(&r.e0);
        hipEventCreate(&r.e1);
        ctx->krecs.push_back(r);
    }
    ctx->krecs[ctx->kused].name = name;
    hipEventRecord(ctx->krecs[ctx->kused].e0, PSS_STREAM(ctx));
}
void pss_kernel_end(pss_ctx *ctx)
{
    if (!ctx->timing || ctx->kskip) return;
    hipEventRecord(ctx->krecs[ctx->kused].e1, PSS_STREAM(ctx));
    ctx->kused++;
}
void pss_time_end(pss_ctx *ctx)
{
    if (ctx->timing && --ctx->tdepth == 0) {
        if (ctx->tfilter.empty()) hipEventRecord(ctx->ev1, ctx->stream);
        ctx->last_ms = 0.0f;
    }
}

extern "C" int pss_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int pss_create(int device, pss_ctx **out)
{
    if (!out) return PSS_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return pss_fail(nullptr, PSS_E_HIP, std::string("no HIP device available (libpss has no CPU fallback): ") +
                                                (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
    if (device < 0 || device >= n) return pss_fail(nullptr, PSS_E_ARG, "device index out of range");
    int prev = -1;
    hipGetDevice(&prev);
    e = hipSetDevice(device);
    if (e != hipSuccess) return pss_fail(nullptr, PSS_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    struct Restore {  // the caller's current device is left as it was
        int prev, dev;
        ~Restore() { if (prev >= 0 && prev != dev) hipSetDevice(prev); }
    } restore{prev, device};
    pss_ctx *ctx = new pss_ctx();
    ctx->device = device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) ctx->n_cus = cus; }
    { const char *e = getenv("PSS_NO_FUSED"); ctx->no_fused = e && e[0] == '1'; }
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) ctx->own_stream = true;
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) {
        const std::string msg = std::string("context stream / event creation: ") + hipGetErrorString(e);
        if (ctx->ev0) hipEventDestroy(ctx->ev0);
        if (ctx->ev1) hipEventDestroy(ctx->ev1);
        if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
        if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
        if (ctx->stream2) hipStreamDestroy(ctx->stream2);
        if (ctx->own_stream) hipStreamDestroy(ctx->stream);
        delete ctx;
        return pss_fail(nullptr, PSS_E_HIP, msg);
    }
    *out = ctx;
    return PSS_OK;
}

extern "C" void pss_destroy(pss_ctx *ctx)
{
    if (!ctx) return;
    PssDevGuard guard(ctx->device);
    hipStreamSynchronize(ctx->stream);
    if (ctx->comm) pss_comm_free(ctx);
    for (auto &kv : ctx->tw) hipFree(kv.second);
    for (auto &kv : ctx->tw_pf) hipFree(kv.second);
    for (auto &kv : ctx->win) hipFree(kv.second);
    for (auto &kv : ctx->bs) { hipFree(kv.second.d_chirp); hipFree(kv.second.d_B); hipFree(kv.second.d_win); }
    for (auto &kv : ctx->plans) {
        hipFree(kv.second.d_leaf_off); hipFree(kv.second.d_leaf_len); hipFree(kv.second.d_node_l);
        hipFree(kv.second.d_node_r); hipFree(kv.second.d_level_start); hipFree(kv.second.d_roots);
    }
    for (auto &kv : ctx->nfm) if (kv.second.d_rev) hipFree(kv.second.d_rev);
    if (ctx->scratch) hipFree(ctx->scratch);
    if (ctx->scratch_fft) hipFree(ctx->scratch_fft);
    if (ctx->scratch_hil) hipFree(ctx->scratch_hil);
    if (ctx->scratch_iqc) hipFree(ctx->scratch_iqc);
    if (ctx->d_hann) hipFree(ctx->d_hann);
    if (ctx->d_hann_short) hipFree(ctx->d_hann_short);
    if (ctx->st_up) hipStreamDestroy(ctx->st_up);
    if (ctx->st_dn) hipStreamDestroy(ctx->st_dn);
    for (auto &e : ctx->st_ev) if (e) hipEventDestroy(e);
    for (auto &b : ctx->st_buf) if (b) hipFree(b);
    if (ctx->scratch_scan) hipFree(ctx->scratch_scan);
    if (ctx->scratch_pk) hipFree(ctx->scratch_pk);
    if (ctx->scratch_win) hipFree(ctx->scratch_win);
    if (ctx->scratch_db64) hipFree(ctx->scratch_db64);
    if (ctx->scratch_post) hipFree(ctx->scratch_post);
    if (ctx->prog) hipFree(ctx->prog);
    if (ctx->stage) hipFree(ctx->stage);
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    hipStreamSynchronize(ctx->stream2);
    hipStreamDestroy(ctx->stream2);
    hipEventDestroy(ctx->ev_fork);
    hipEventDestroy(ctx->ev_join);
    if (ctx->ev_in) hipEventDestroy(ctx->ev_in);
    if (ctx->ev_out) hipEventDestroy(ctx->ev_out);
    for (auto &k : ctx->krecs) { hipEventDestroy(k.e0); hipEventDestroy(k.e1); }
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int pss_set_stream(pss_ctx *ctx, void *hip_stream)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return PSS_OK;
}

// Ordering against a caller's stream without giving up the context's own (non-blocking) stream: one event record + one stream wait each,
// nothing blocks the host.
extern "C" int pss_order_after(pss_ctx *ctx, void *hip_stream)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    if (s == ctx->stream) return PSS_OK;
    if (!ctx->ev_in) PSS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming));
    PSS_HIP(ctx, hipEventRecord(ctx->ev_in, s));
    PSS_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_in, 0));
    return PSS_OK;
}

extern "C" int pss_order_before(pss_ctx *ctx, void *hip_stream)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    hipStream_t s = reinterpret_cast<hipStream_t>(hip_stream);
    if (s == ctx->stream) return PSS_OK;
    if (!ctx->ev_out) PSS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_out, hipEventDisableTiming));
    PSS_HIP(ctx, hipEventRecord(ctx->ev_out, ctx->stream));
    PSS_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_out, 0));
    return PSS_OK;
}

extern "C" void *pss_get_stream(pss_ctx *ctx) { return ctx ? reinterpret_cast<void *>(ctx->stream) : nullptr; }

extern "C" int pss_set_option(pss_ctx *ctx, const char *key, int value)
{
    if (!ctx || !key) return PSS_E_ARG;
    if (!strcmp(key, "nfm_fused")) { ctx->no_fused = (value == 0); return PSS_OK; }
    if (!strcmp(key, "wfm_fused")) { ctx->no_wfm_fused = (value == 0); return PSS_OK; }
    if (!strcmp(key, "small_batch")) { ctx->no_small_batch = (value == 0); return PSS_OK; }
    if (!strcmp(key, "small_batch_max")) { ctx->small_batch_max = value; return PSS_OK; }
    if (!strcmp(key, "wfm_small_batch_max")) { ctx->wfm_small_batch_max = value; return PSS_OK; }
    if (!strcmp(key, "fuse_post")) { ctx->fuse_post = value != 0; return PSS_OK; }
    if (!strcmp(key, "ssb_rfft")) { ctx->ssb_rfft = value != 0; return PSS_OK; }
    if (!strcmp(key, "ssb_hilbert")) { ctx->ssb_hilbert = value != 0; return PSS_OK; }
    if (!strcmp(key, "hilbert_exact")) { ctx->hilbert_exact = value != 0; return PSS_OK; }
    if (!strcmp(key, "scan_exact")) { ctx->scan_exact = value != 0; return PSS_OK; }
    if (!strcmp(key, "db_exact")) { ctx->db_exact = value != 0; return PSS_OK; }
    if (!strcmp(key, "f64_plain")) { ctx->f64_plain = value != 0; return PSS_OK; }
    if (!strcmp(key, "wfm_corr_copy")) { ctx->wfm_corr_copy = value != 0; return PSS_OK; }
#ifdef PSS_VARIANTS   // kernel-selection knobs for A/B measurements: variant builds only (tools/build_variant.py <name> -DPSS_VARIANTS)
    if (!strcmp(key, "ssb_unfused")) { ctx->ssb_unfused = value != 0; return PSS_OK; }
    if (!strcmp(key, "fft_two_per_wg")) { ctx->fft_two_per_wg = value != 0; return PSS_OK; }
    if (!strcmp(key, "fft_split")) { ctx->fft_split = value; return PSS_OK; }
    if (!strcmp(key, "fft_big_scratch")) { ctx->fft_big_scratch = value != 0; return PSS_OK; }
    if (!strcmp(key, "fft_prefetch")) { ctx->fft_prefetch = value; return PSS_OK; }
    if (!strcmp(key, "fft_xl4096")) { ctx->fft_xl4096 = value; return PSS_OK; }
    if (!strcmp(key, "post_sort_max")) { ctx->post_sort_max = value; return PSS_OK; }
    if (!strcmp(key, "post_legacy")) { ctx->post_legacy = value != 0; return PSS_OK; }
#endif
    return pss_fail(ctx, PSS_E_ARG, std::string("unknown option: ") + key);
}

extern "C" int pss_sync(pss_ctx *ctx)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    return pss_hip_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

extern "C" const char *pss_last_error(pss_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int pss_enable_timing(pss_ctx *ctx, int on)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    ctx->timing = on != 0;
    ctx->tdepth = 0;
    ctx->kused = 0;
    ctx->last_ms = -1.0f;
    return PSS_OK;
}

extern "C" int pss_timing_filter(pss_ctx *ctx, const char *kernel)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    ctx->tfilter = kernel ? kernel : "";
    ctx->kused = 0;
    ctx->last_ms = -1.0f;
    return PSS_OK;
}

extern "C" float pss_last_kernel_ms(pss_ctx *ctx)
{
    if (!ctx || !ctx->timing || ctx->last_ms < 0.0f || !ctx->tfilter.empty()) return -1.0f;
    PSS_GUARD(ctx);
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
    return ms;
}

extern "C" int pss_kernel_times(pss_ctx *ctx, char *buf, int buf_len)
{
    // "name=ms;name=ms;..." for every kernel launched since timing was enabled / since the previous read
    if (!ctx || !buf || buf_len < 1) return PSS_E_ARG;
    PSS_GUARD(ctx);
    std::string out;
    if (ctx->timing && ctx->last_ms >= 0.0f) {
        if (ctx->tfilter.empty()) hipEventSynchronize(ctx->ev1);
        else hipStreamSynchronize(ctx->stream);
        for (int i = 0; i < ctx->kused; i++) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, ctx->krecs[i].e0, ctx->krecs[i].e1) != hipSuccess) continue;
            out += ctx->krecs[i].name;
            out += "=" + std::to_string(ms) + ";";
        }
    }
    ctx->kused = 0;
    if ((int)out.size() + 1 > buf_len) return pss_fail(ctx, PSS_E_ARG, "buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return PSS_OK;
}

// ---- host-buffer convenience: one frame, synchronous ------------------------------------------------
// Device staging comes from one grow-only buffer owned by the context (no hipMalloc per call: this is the path the
// drop-in module takes on every loop iteration of the host application).
namespace {
size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
}  // namespace

extern "C" int pss_h_compute_fft(pss_ctx *ctx, const float *h_iq, int n, double *h_db)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || !h_db || n < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_db = up256(sizeof(float) * 2 * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_db + up256(sizeof(float) * n), "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_spectrum_db(ctx, reinterpret_cast<const float *>(base), 1, n, reinterpret_cast<float *>(base + o_db));
    if (r) return r;
    std::vector<float> tmp(n);
    PSS_HIP(ctx, hipMemcpyAsync(tmp.data(), base + o_db, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; i++) h_db[i] = (double)tmp[i];
    return PSS_OK;
}

// complex128 read buffers (the reference computes these two functions in float64 then; tests/golden/c128.npz)
extern "C" int pss_h_compute_fft_c128(pss_ctx *ctx, const double *h_iq, int n, double *h_db)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || !h_db || n < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_db = up256(sizeof(double) * 2 * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_db + up256(sizeof(double) * n), "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(double) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_spectrum_db_c128(ctx, reinterpret_cast<const double *>(base), 1, n, reinterpret_cast<double *>(base + o_db));
    if (r) return r;
    PSS_HIP(ctx, hipMemcpyAsync(h_db, base + o_db, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSS_OK;
}

extern "C" int pss_h_demodulate_am_c128(pss_ctx *ctx, const double *h_iq, int n, double *h_audio_stereo, int16_t *h_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || n < 1 || (!h_audio_stereo && !h_pcm)) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_pcm = up256(sizeof(double) * 2 * n), o_au = o_pcm + up256(sizeof(int16_t) * 2 * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_au + up256(sizeof(double) * n), "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(double) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_demod_am_c128(ctx, reinterpret_cast<const double *>(base), 1, n, reinterpret_cast<int16_t *>(base + o_pcm), reinterpret_cast<double *>(base + o_au));
    if (r) return r;
    std::vector<double> au(n);
    PSS_HIP(ctx, hipMemcpyAsync(au.data(), base + o_au, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (h_pcm) PSS_HIP(ctx, hipMemcpyAsync(h_pcm, base + o_pcm, sizeof(int16_t) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_audio_stereo)
        for (int i = 0; i < n; i++) h_audio_stereo[2 * i] = h_audio_stereo[2 * i + 1] = au[i];   // mono_to_stereo (:83-88)
    return PSS_OK;
}

extern "C" int pss_h_demodulate_ssb_c128(pss_ctx *ctx, int lower, const double *h_iq, int n, double fs, double *h_audio_stereo, int16_t *h_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || n < 1 || (!h_audio_stereo && !h_pcm)) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_pcm = up256(sizeof(double) * 2 * n), o_au = o_pcm + up256(sizeof(int16_t) * 2 * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_au + up256(sizeof(double) * n), "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(double) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_demod_ssb_c128(ctx, lower, reinterpret_cast<const double *>(base), 1, n, fs, reinterpret_cast<int16_t *>(base + o_pcm), reinterpret_cast<double *>(base + o_au));
    if (r) return r;
    std::vector<double> au(n);
    PSS_HIP(ctx, hipMemcpyAsync(au.data(), base + o_au, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    if (h_pcm) PSS_HIP(ctx, hipMemcpyAsync(h_pcm, base + o_pcm, sizeof(int16_t) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_audio_stereo)
        for (int i = 0; i < n; i++) h_audio_stereo[2 * i] = h_audio_stereo[2 * i + 1] = au[i];   // mono_to_stereo (:83-88)
    return PSS_OK;
}

extern "C" int pss_h_mean_power_c128(pss_ctx *ctx, const double *h_iq, int n, double *h_power)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || !h_power || n < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_p = up256(sizeof(double) * 2 * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_p + 256, "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(double) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_mean_power_c128(ctx, reinterpret_cast<const double *>(base), 1, n, reinterpret_cast<double *>(base + o_p));
    if (r) return r;
    PSS_HIP(ctx, hipMemcpyAsync(h_power, base + o_p, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSS_OK;
}

static int h_demod_impl(pss_ctx *ctx, int mode, const float *h_iq, int n, double fs, double *h_audio_stereo,
                        int16_t *h_pcm, bool dispatcher)
{
    if (!ctx) return PSS_E_ARG;
    if (!h_iq || n < 1 || (!h_audio_stereo && !h_pcm)) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    int n_out = pss_demod_out_len_ctx(ctx, mode, n, fs);
    if (n_out < 0) return pss_fail(ctx, PSS_E_ARG, "unknown mode or sample rate below the target rate");
    const bool stereo = mode == PSS_MODE_WFM;
    const size_t o_pcm = up256(sizeof(float) * 2 * n), o_au = o_pcm + up256(sizeof(int16_t) * 2 * n_out);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_au + up256(sizeof(double) * 2 * n_out), "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = (dispatcher ? pss_demod_signal : pss_demod)(ctx, mode, reinterpret_cast<const float *>(base), 1, n, fs,
                                                    reinterpret_cast<int16_t *>(base + o_pcm),
                                                    reinterpret_cast<double *>(base + o_au));
    if (r) return r;
    std::vector<double> au((size_t)n_out * (stereo ? 2 : 1));
    PSS_HIP(ctx, hipMemcpyAsync(au.data(), base + o_au, sizeof(double) * au.size(), hipMemcpyDeviceToHost, ctx->stream));
    if (h_pcm) PSS_HIP(ctx, hipMemcpyAsync(h_pcm, base + o_pcm, sizeof(int16_t) * 2 * n_out, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_audio_stereo) {
        if (stereo) memcpy(h_audio_stereo, au.data(), sizeof(double) * au.size());  // np.column_stack((left, right))
        else
            for (int i = 0; i < n_out; i++) h_audio_stereo[2 * i] = h_audio_stereo[2 * i + 1] = au[i];  // mono_to_stereo
    }
    return PSS_OK;
}

extern "C" int pss_h_demodulate(pss_ctx *ctx, int mode, const float *h_iq, int n, double fs, double *h_audio_stereo,
                                int16_t *h_pcm)
{
    return h_demod_impl(ctx, mode, h_iq, n, fs, h_audio_stereo, h_pcm, false);
}

extern "C" int pss_h_demodulate_signal(pss_ctx *ctx, int mode, const float *h_iq, int n, double fs,
                                       double *h_audio_stereo, int16_t *h_pcm)
{
    return h_demod_impl(ctx, mode, h_iq, n, fs, h_audio_stereo, h_pcm, true);
}

// demodulate_signal over a batch of frames in HOST memory (a recording cut into read buffers, pyspecsdr.py:814-824 /
// :2236): chunks of frames go up, through pss_demod_signal, and the int16 PCM comes back — the byte stream
// audio_processing.write_audio_samples / io_manager.write_to_pipe would have produced buffer by buffer.
extern "C" int pss_h_demodulate_batch(pss_ctx *ctx, int mode, const float *h_iq, long n_frames, int n, double fs,
                                      long chunk_frames, int16_t *h_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || !h_pcm || n_frames < 0 || n < 1 || chunk_frames < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const int n_out = pss_demod_out_len_ctx(ctx, mode, n, fs);
    if (n_out < 0) return pss_fail(ctx, PSS_E_ARG, "unknown mode or sample rate below the target rate");
    if (chunk_frames > n_frames) chunk_frames = n_frames;
    if (n_frames == 0) return PSS_OK;
    const size_t iq_b = up256((size_t)chunk_frames * n * 2 * sizeof(float));
    const size_t pcm_b = up256((size_t)chunk_frames * n_out * 2 * sizeof(int16_t));
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, iq_b + pcm_b, "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    for (long f0 = 0; f0 < n_frames; f0 += chunk_frames) {
        const long cnt = (n_frames - f0) < chunk_frames ? (n_frames - f0) : chunk_frames;
        PSS_HIP(ctx, hipMemcpyAsync(base, h_iq + (size_t)f0 * n * 2, (size_t)cnt * n * 2 * sizeof(float), hipMemcpyHostToDevice,
                                    ctx->stream));
        r = pss_demod_signal(ctx, mode, reinterpret_cast<const float *>(base), cnt, n, fs, reinterpret_cast<int16_t *>(base + iq_b),
                             nullptr);
        if (r) return r;
        PSS_HIP(ctx, hipMemcpyAsync(h_pcm + (size_t)f0 * n_out * 2, base + iq_b, (size_t)cnt * n_out * 2 * sizeof(int16_t),
                                    hipMemcpyDeviceToHost, ctx->stream));
        PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return PSS_OK;
}

extern "C" int pss_h_measure_power(pss_ctx *ctx, const float *h_iq, int n, float *h_power)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || !h_power || n < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_p = up256(sizeof(float) * 2 * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_p + 256, "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_power_db(ctx, reinterpret_cast<const float *>(base), 1, n, reinterpret_cast<float *>(base + o_p));
    if (r) return r;
    PSS_HIP(ctx, hipMemcpyAsync(h_power, base + o_p, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSS_OK;
}

extern "C" int pss_h_morse_edges(pss_ctx *ctx, const float *h_iq, int n, double threshold_db, int cap, int32_t *h_rise,
                                 int32_t *h_fall, int *n_rise, int *n_fall)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || n < 1 || cap < 0 || (cap > 0 && (!h_rise || !h_fall))) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_r = up256(sizeof(float) * 2 * n), o_f = o_r + up256(sizeof(int32_t) * (size_t)cap), o_c = o_f + up256(sizeof(int32_t) * (size_t)cap);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_c + 256, "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_morse_edges(ctx, reinterpret_cast<const float *>(base), 1, n, threshold_db, cap, reinterpret_cast<int32_t *>(base + o_r),
                        reinterpret_cast<int32_t *>(base + o_f), reinterpret_cast<int32_t *>(base + o_c));
    if (r) return r;
    int32_t cnt[2];
    PSS_HIP(ctx, hipMemcpyAsync(cnt, base + o_c, sizeof cnt, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int kr = cnt[0] < cap ? cnt[0] : cap, kf = cnt[1] < cap ? cnt[1] : cap;
    if (kr) PSS_HIP(ctx, hipMemcpyAsync(h_rise, base + o_r, sizeof(int32_t) * kr, hipMemcpyDeviceToHost, ctx->stream));
    if (kf) PSS_HIP(ctx, hipMemcpyAsync(h_fall, base + o_f, sizeof(int32_t) * kf, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n_rise) *n_rise = cnt[0];
    if (n_fall) *n_fall = cnt[1];
    return PSS_OK;
}

extern "C" int pss_h_classify_signal(pss_ctx *ctx, const float *h_iq, int n, double fs, int *label, double *bw, float *mi, float *flat)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || n < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_out = up256(sizeof(float) * 2 * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_out + 256, "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    // results: [bw f64][label i32][mi f32][flat f32]
    double *d_bw = reinterpret_cast<double *>(base + o_out);
    int32_t *d_lab = reinterpret_cast<int32_t *>(base + o_out + 8);
    float *d_mi = reinterpret_cast<float *>(base + o_out + 12), *d_fl = reinterpret_cast<float *>(base + o_out + 16);
    r = pss_classify(ctx, reinterpret_cast<const float *>(base), 1, n, fs, d_lab, d_bw, d_mi, d_fl, nullptr);
    if (r) return r;
    unsigned char res[24];
    PSS_HIP(ctx, hipMemcpyAsync(res, base + o_out, sizeof res, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (bw) memcpy(bw, res, 8);
    int32_t lab;
    memcpy(&lab, res + 8, 4);
    if (label) *label = lab;
    if (mi) memcpy(mi, res + 12, 4);
    if (flat) memcpy(flat, res + 16, 4);
    return PSS_OK;
}

extern "C" int pss_h_iq_correction(pss_ctx *ctx, const float *h_iq, int n, float *h_out_iq, float *h_raw)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || (!h_out_iq && !h_raw) || n < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    const size_t o_c = up256(sizeof(float) * 2 * n), o_r = o_c + up256(sizeof(float) * 2 * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, o_r + up256(sizeof(float) * n), "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_iq_correction(ctx, reinterpret_cast<const float *>(base), 1, n,
                          h_out_iq ? reinterpret_cast<float *>(base + o_c) : nullptr,
                          h_raw ? reinterpret_cast<float *>(base + o_r) : nullptr);
    if (r) return r;
    if (h_out_iq) PSS_HIP(ctx, hipMemcpyAsync(h_out_iq, base + o_c, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, ctx->stream));
    if (h_raw) PSS_HIP(ctx, hipMemcpyAsync(h_raw, base + o_r, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSS_OK;
}

// bandpass_filter(data, lowcut, highcut, sample_rate) (signal_processing.py:34-42) on one host row of float64 samples.
// sos: caller-supplied table (nsec rows) or NULL to design butter(5, ...) natively.
extern "C" int pss_h_bandpass_filter(pss_ctx *ctx, const double *h_x, int n, double lowcut, double highcut, double fs,
                                     const double *sos, int nsec, double *h_y)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_x || !h_y || n < 0) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    double tbl[48];
    if (!sos) {
        int r = pss_design_butter_sos(5, lowcut <= 0 ? 0.0 : lowcut / (fs / 2.0), highcut / (fs / 2.0), tbl, &nsec);
        if (r) return pss_fail(ctx, r, "butter: digital filter critical frequencies must be 0 < Wn < 1");
        sos = tbl;
    }
    if (n == 0) return PSS_OK;
    const size_t o_y = up256(sizeof(double) * n);
    int r = pss_ensure_buffer(ctx, &ctx->stage, &ctx->stage_bytes, 2 * o_y, "staging");
    if (r) return r;
    char *base = reinterpret_cast<char *>(ctx->stage);
    PSS_HIP(ctx, hipMemcpyAsync(base, h_x, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_sosfilt(ctx, reinterpret_cast<const double *>(base), 1, n, sos, nsec, reinterpret_cast<double *>(base + o_y));
    if (r) return r;
    PSS_HIP(ctx, hipMemcpyAsync(h_y, base + o_y, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSS_OK;
}

// ---- streamed capture --------------------------------------------------------------------------------
extern "C" void *pss_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) return nullptr;  // pinned for every device
    return p;
}

extern "C" void pss_host_free(void *p)
{
    if (p) hipHostFree(p);
}

// Streaming resources of a context: created on first use, grown when a capture needs more, released by pss_destroy.
static int stream_res(pss_ctx *ctx)
{
    if (!ctx->st_up) PSS_HIP(ctx, hipStreamCreateWithFlags(&ctx->st_up, hipStreamNonBlocking));
    if (!ctx->st_dn) PSS_HIP(ctx, hipStreamCreateWithFlags(&ctx->st_dn, hipStreamNonBlocking));
    for (auto &e : ctx->st_ev)
        if (!e) PSS_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return PSS_OK;
}
static int stream_buf(pss_ctx *ctx, int slot, size_t bytes, void **out)
{
    if (ctx->st_cap[slot] < bytes) {
        if (ctx->st_buf[slot]) { PSS_HIP(ctx, hipFree(ctx->st_buf[slot])); ctx->st_buf[slot] = nullptr; ctx->st_cap[slot] = 0; }
        PSS_HIP(ctx, hipMalloc(&ctx->st_buf[slot], bytes));
        ctx->st_cap[slot] = bytes;
    }
    *out = ctx->st_buf[slot];
    return PSS_OK;
}
// every queued copy / kernel of a capture has finished (also on the error paths: the caller's host buffers must not be in use)
static void stream_drain(pss_ctx *ctx)
{
    hipStreamSynchronize(ctx->stream);
    if (ctx->st_up) hipStreamSynchronize(ctx->st_up);
    if (ctx->st_dn) hipStreamSynchronize(ctx->st_dn);
}

extern "C" int pss_h_stream_spectrum_nfm(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames,
                                         float *h_db, int16_t *h_pcm)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (!h_iq || !h_pcm || n_frames < 0 || n < 1 || chunk_frames < 1) return pss_fail(ctx, PSS_E_ARG, "bad stream arguments");
    const int n_out = pss_demod_out_len_ctx(ctx, PSS_MODE_NFM, n, fs);
    if (n_out < 0) return pss_fail(ctx, PSS_E_ARG, "sample rate below the target rate");
    if (n_frames == 0) return PSS_OK;
    if (chunk_frames > n_frames) chunk_frames = n_frames;
    const size_t iq_b = (size_t)chunk_frames * n * 2 * sizeof(float), db_b = (size_t)chunk_frames * n * sizeof(float),
                 pcm_b = (size_t)chunk_frames * n_out * 2 * sizeof(int16_t);
    int rc = stream_res(ctx);
    if (rc) return rc;
    hipStream_t s_up = ctx->st_up, s_dn = ctx->st_dn;
    hipEvent_t *up_done = ctx->st_ev, *cmp_done = ctx->st_ev + 2, *dn_done = ctx->st_ev + 4;
    void *d_iq[2], *d_db[2], *d_pcm[2];
    for (int i = 0; i < 2; i++) {
        if (!rc) rc = stream_buf(ctx, 0 + i, iq_b, &d_iq[i]);
        if (!rc) rc = stream_buf(ctx, 2 + i, db_b, &d_db[i]);
        if (!rc) rc = stream_buf(ctx, 4 + i, pcm_b, &d_pcm[i]);
    }
    if (rc) return rc;
    auto cleanup = [&]() { stream_drain(ctx); };
#define STREAM_HIP(call)                                          \
    do {                                                          \
        rc = pss_hip_check(ctx, (call), #call);                   \
        if (rc) { cleanup(); return rc; }                         \
    } while (0)
    const long n_chunks = (n_frames + chunk_frames - 1) / chunk_frames;
    for (long k = 0; k < n_chunks; k++) {
        const int b = (int)(k & 1);
        const long f0 = k * chunk_frames;
        const long cnt = (n_frames - f0) < chunk_frames ? (n_frames - f0) : chunk_frames;
        // the buffer set is free once chunk k-2's download has finished
        if (k >= 2) STREAM_HIP(hipEventSynchronize(dn_done[b]));   // host-side wait: see pss_h_stream_display_nfm
        STREAM_HIP(hipMemcpyAsync(d_iq[b], h_iq + (size_t)f0 * n * 2, (size_t)cnt * n * 2 * sizeof(float),
                                  hipMemcpyHostToDevice, s_up));
        STREAM_HIP(hipEventRecord(up_done[b], s_up));
        STREAM_HIP(hipStreamWaitEvent(ctx->stream, up_done[b], 0));
        if (k >= 2) STREAM_HIP(hipStreamWaitEvent(ctx->stream, dn_done[b], 0));
        {
            // chunks of a stream are throughput work: keep them on the fused large-batch kernels, which also overlap the
            // spectrum kernel with the backward pass (the small-batch path measured 1.5x slower here)
            PssFlagScope keep(ctx->no_small_batch, true);
            rc = pss_spectrum_nfm(ctx, (const float *)d_iq[b], cnt, n, fs, (float *)d_db[b], (int16_t *)d_pcm[b]);
        }
        if (rc) { cleanup(); return rc; }
        STREAM_HIP(hipEventRecord(cmp_done[b], ctx->stream));
        STREAM_HIP(hipStreamWaitEvent(s_dn, cmp_done[b], 0));
        if (h_db)
            STREAM_HIP(hipMemcpyAsync(h_db + (size_t)f0 * n, d_db[b], (size_t)cnt * n * sizeof(float), hipMemcpyDeviceToHost, s_dn));
        STREAM_HIP(hipMemcpyAsync(h_pcm + (size_t)f0 * n_out * 2, d_pcm[b], (size_t)cnt * n_out * 2 * sizeof(int16_t),
                                  hipMemcpyDeviceToHost, s_dn));
        STREAM_HIP(hipEventRecord(dn_done[b], s_dn));
    }
#undef STREAM_HIP
    cleanup();
    return PSS_OK;
}

// BASELINE.json configs[4] as the host sees it: a long capture streamed through the GPU, and what comes back per frame is
// what the application consumes — the display accumulator's line (waterfall: glyph + colour, persistence: the newest trace's
// row indices) and the int16 PCM — not the dB rows (8 KB per frame of PCIe traffic nobody reads; still available: h_db).
// Per chunk: upload (copy stream) | spectrum + NFM, post-process + row extremes, display lines (compute stream) | download
// (copy stream), two buffer sets.  The row extremes of the whole capture stay in one device array, so a frame's history
// window reaches back across chunk boundaries; h_halo_lo / h_halo_hi (n_halo values each, may be NULL) are the extremes of
// the rows that precede this capture (previous capture, or the left neighbour's tail when the capture is sharded).
// TR = float: float32 dB / post-processed rows (the float32 pipeline's cells); TR = double: compute_fft's own float64 rows from the transform to the
// cells (pss_h_stream_display_nfm_f64: the reference's cells; the capture is PCIe-bound either way).
namespace {
inline int sd_spectrum_nfm(pss_ctx *ctx, const float *d_iq, long cnt, int n, double fs, float *d_db, int16_t *d_pcm) { return pss_spectrum_nfm(ctx, d_iq, cnt, n, fs, d_db, d_pcm); }
inline int sd_spectrum_nfm(pss_ctx *ctx, const float *d_iq, long cnt, int n, double fs, double *d_db, int16_t *d_pcm)
{
    int r = pss_demod(ctx, PSS_MODE_NFM, d_iq, cnt, n, fs, d_pcm, nullptr);
    return r ? r : pss_spectrum_db_f64(ctx, d_iq, cnt, n, d_db);
}
inline int sd_post(pss_ctx *ctx, const float *d_db, long cnt, int n, float *d_post, float *lo, float *hi) { return pss_spectrum_post_extremes(ctx, d_db, cnt, n, d_post, lo, hi); }
inline int sd_post(pss_ctx *ctx, const double *d_db, long cnt, int n, double *d_post, double *lo, double *hi) { return pss_spectrum_post_f64(ctx, d_db, cnt, n, d_post, lo, hi); }
inline int sd_wf_rows(pss_ctx *c, const float *p, long n, int m, const float *lo, const float *hi, int h, int w, int dw, int8_t *a, int8_t *b) { return pss_waterfall_rows(c, p, n, m, lo, hi, h, w, dw, a, b); }
inline int sd_wf_rows(pss_ctx *c, const double *p, long n, int m, const double *lo, const double *hi, int h, int w, int dw, int8_t *a, int8_t *b) { return pss_waterfall_rows_f64(c, p, n, m, lo, hi, h, w, dw, a, b); }
inline int sd_ps_rows(pss_ctx *c, const float *p, long n, int m, const float *lo, const float *hi, int h, int w, int dh, int dw, int8_t *a) { return pss_persistence_rows(c, p, n, m, lo, hi, h, w, dh, dw, a); }
inline int sd_ps_rows(pss_ctx *c, const double *p, long n, int m, const double *lo, const double *hi, int h, int w, int dh, int dw, int8_t *a) { return pss_persistence_rows_f64(c, p, n, m, lo, hi, h, w, dh, dw, a); }
inline int sd_wf_cells(pss_ctx *c, const float *r, int nr, int m, int dh, int dw, int8_t *a, int8_t *b) { return pss_waterfall_cells(c, r, nr, m, dh, dw, a, b); }
inline int sd_wf_cells(pss_ctx *c, const double *r, int nr, int m, int dh, int dw, int8_t *a, int8_t *b) { return pss_waterfall_cells_f64(c, r, nr, m, dh, dw, a, b); }
inline int sd_ps_cells(pss_ctx *c, const float *r, int nr, int m, int dh, int dw, int8_t *a) { return pss_persistence_cells(c, r, nr, m, dh, dw, a); }
inline int sd_ps_cells(pss_ctx *c, const double *r, int nr, int m, int dh, int dw, int8_t *a) { return pss_persistence_cells_f64(c, r, nr, m, dh, dw, a); }
}  // namespace
template <class TR>
static int stream_display(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames,
                          int mode, int window, int disp_h, int disp_w, const TR *h_halo_lo,
                          const TR *h_halo_hi, int n_halo, int8_t *h_line_a, int8_t *h_line_b, int16_t *h_pcm,
                          TR *h_db, TR *h_row_lo, TR *h_row_hi, int8_t *h_grid_a, int8_t *h_grid_b)
{
    if (!ctx) return PSS_E_ARG;
    PSS_GUARD(ctx);
    if (h_grid_a && (window > 64 || (mode == 0 && !h_grid_b)))
        return pss_fail(ctx, PSS_E_ARG, "stream display grids: window <= 64, and both planes for the waterfall");
    if (!h_iq || !h_pcm || !h_line_a || n_frames < 0 || n < 8 || chunk_frames < 1 || window < 1 || disp_w < 1 || disp_h < 1 ||
        disp_h > 127 || n_halo < 0 || (mode != 0 && mode != 1) || (mode == 0 && !h_line_b) || (n_halo > 0 && (!h_halo_lo || !h_halo_hi)))
        return pss_fail(ctx, PSS_E_ARG, "bad stream-display arguments");
    const int n_out = pss_demod_out_len_ctx(ctx, PSS_MODE_NFM, n, fs);
    if (n_out < 0) return pss_fail(ctx, PSS_E_ARG, "sample rate below the target rate");
    if (n_frames == 0) return PSS_OK;
    if (chunk_frames > n_frames) chunk_frames = n_frames;
    const int m = n - 4;
    const size_t iq_b = (size_t)chunk_frames * n * 2 * sizeof(float), db_b = (size_t)chunk_frames * n * sizeof(TR),
                 post_b = (size_t)chunk_frames * m * sizeof(TR), pcm_b = (size_t)chunk_frames * n_out * 2 * sizeof(int16_t),
                 line_b = (size_t)chunk_frames * disp_w;
    int rc = stream_res(ctx);
    if (rc) return rc;
    hipStream_t s_up = ctx->st_up, s_dn = ctx->st_dn;
    hipEvent_t *up_done = ctx->st_ev, *cmp_done = ctx->st_ev + 2, *dn_done = ctx->st_ev + 4;
    void *d_iq[2], *d_db[2], *d_pcm[2], *d_la[2], *d_lb[2] = {nullptr, nullptr};
    void *d_post = nullptr, *d_ext = nullptr;   // post rows of the chunk in flight; [lo | hi] x (n_halo + n_frames)
    const size_t n_ext = (size_t)n_halo + (size_t)n_frames;
    for (int i = 0; i < 2; i++) {
        if (!rc) rc = stream_buf(ctx, 0 + i, iq_b, &d_iq[i]);
        if (!rc) rc = stream_buf(ctx, 2 + i, db_b, &d_db[i]);
        if (!rc) rc = stream_buf(ctx, 4 + i, pcm_b, &d_pcm[i]);
        if (!rc) rc = stream_buf(ctx, 6 + i, line_b, &d_la[i]);
        if (!rc && mode == 0) rc = stream_buf(ctx, 8 + i, line_b, &d_lb[i]);
    }
    if (!rc) rc = stream_buf(ctx, 10, post_b, &d_post);
    if (!rc) rc = stream_buf(ctx, 11, 2 * n_ext * sizeof(TR), &d_ext);
    // full screens (h_grid_a): after the LAST frame of every chunk the whole grid as the reference redraws it — all lines / traces of the
    // history normalised with the history's current extremes (pyspecsdr.py:1342-1406, :1512-1564) — from the last `window` post-processed
    // rows, which are carried from chunk to chunk in a small device buffer
    void *d_ga[2] = {nullptr, nullptr}, *d_gb[2] = {nullptr, nullptr}, *d_tail[2] = {nullptr, nullptr};
    const size_t grid_b = (size_t)disp_h * disp_w, tail_b = (size_t)window * m * sizeof(TR);
    if (h_grid_a) {
        for (int i = 0; i < 2; i++) {
            if (!rc) rc = stream_buf(ctx, 12 + i, grid_b, &d_ga[i]);
            if (!rc && mode == 0) rc = stream_buf(ctx, 14 + i, grid_b, &d_gb[i]);
            if (!rc) rc = stream_buf(ctx, 16 + i, tail_b, &d_tail[i]);
        }
    }
    long tail_rows = 0;   // rows of the history held in d_tail[tail_cur], oldest first
    int tail_cur = 0;
    if (rc) return rc;
    auto cleanup = [&]() { stream_drain(ctx); };
#define STREAM_HIP(call)                                          \
    do {                                                          \
        rc = pss_hip_check(ctx, (call), #call);                   \
        if (rc) { cleanup(); return rc; }                         \
    } while (0)
    TR *d_lo = reinterpret_cast<TR *>(d_ext), *d_hi = d_lo + n_ext;
    if (n_halo) {
        STREAM_HIP(hipMemcpyAsync(d_lo, h_halo_lo, sizeof(TR) * n_halo, hipMemcpyHostToDevice, ctx->stream));
        STREAM_HIP(hipMemcpyAsync(d_hi, h_halo_hi, sizeof(TR) * n_halo, hipMemcpyHostToDevice, ctx->stream));
    }
    const long n_chunks = (n_frames + chunk_frames - 1) / chunk_frames;
    for (long k = 0; k < n_chunks; k++) {
        const int b = (int)(k & 1);
        const long f0 = k * chunk_frames;
        const long cnt = (n_frames - f0) < chunk_frames ? (n_frames - f0) : chunk_frames;
        // the buffer set is free once chunk k-2 has been downloaded.  The HOST waits for that (this is a synchronous call: the thread has nothing
        // else to do), so that the upload itself carries no dependency: an upload behind a hipStreamWaitEvent becomes a barrier packet in
        // whichever hardware queue the upload stream shares (the runtime multiplexes streams over a few queues), and sat behind chunk k-1's
        // kernels there — uploads and compute alternated instead of overlapping (21.4 ms per cfg 5 capture; 15 ms with the host-side wait)
        if (k >= 2) STREAM_HIP(hipEventSynchronize(dn_done[b]));
        STREAM_HIP(hipMemcpyAsync(d_iq[b], h_iq + (size_t)f0 * n * 2, (size_t)cnt * n * 2 * sizeof(float), hipMemcpyHostToDevice, s_up));
        STREAM_HIP(hipEventRecord(up_done[b], s_up));
        STREAM_HIP(hipStreamWaitEvent(ctx->stream, up_done[b], 0));
        if (k >= 2) STREAM_HIP(hipStreamWaitEvent(ctx->stream, dn_done[b], 0));
        {
            PssFlagScope keep(ctx->no_small_batch, true);   // chunks of a stream are throughput work: fused large-batch kernels
            rc = sd_spectrum_nfm(ctx, (const float *)d_iq[b], cnt, n, fs, (TR *)d_db[b], (int16_t *)d_pcm[b]);
        }
        // rows f0 .. f0+cnt-1 of the capture sit at positions n_halo + f0 .. of the extremes arrays; their history reaches
        // back over everything before them (previous chunks and the caller's halo)
        if (!rc) rc = sd_post(ctx, (const TR *)d_db[b], cnt, n, (TR *)d_post, d_lo + n_halo + f0, d_hi + n_halo + f0);
        const long before = (long)n_halo + f0;
        const int halo_k = (int)(before < (long)(window - 1) ? before : (long)(window - 1));
        const TR *lo_k = d_lo + n_halo + f0 - halo_k, *hi_k = d_hi + n_halo + f0 - halo_k;
        if (!rc) {
            if (mode == 0) rc = sd_wf_rows(ctx, (const TR *)d_post, cnt, m, lo_k, hi_k, halo_k, window, disp_w, (int8_t *)d_la[b], (int8_t *)d_lb[b]);
            else rc = sd_ps_rows(ctx, (const TR *)d_post, cnt, m, lo_k, hi_k, halo_k, window, disp_h, disp_w, (int8_t *)d_la[b]);
        }
        if (rc) { cleanup(); return rc; }
        if (h_grid_a) {
            // history after this chunk = the last `window` rows of (history before it ++ the chunk's rows)
            const long take = cnt < window ? cnt : window, keep = (tail_rows + take > window) ? window - take : tail_rows;
            TR *dst = (TR *)d_tail[tail_cur ^ 1];
            if (keep > 0)
                STREAM_HIP(hipMemcpyAsync(dst, (const TR *)d_tail[tail_cur] + (size_t)(tail_rows - keep) * m, (size_t)keep * m * sizeof(TR),
                                          hipMemcpyDeviceToDevice, ctx->stream));
            STREAM_HIP(hipMemcpyAsync(dst + (size_t)keep * m, (const TR *)d_post + (size_t)(cnt - take) * m, (size_t)take * m * sizeof(TR),
                                      hipMemcpyDeviceToDevice, ctx->stream));
            tail_cur ^= 1;
            tail_rows = keep + take;
            if (mode == 0) rc = sd_wf_cells(ctx, dst, (int)tail_rows, m, disp_h, disp_w, (int8_t *)d_ga[b], (int8_t *)d_gb[b]);
            else rc = sd_ps_cells(ctx, dst, (int)tail_rows, m, disp_h, disp_w, (int8_t *)d_ga[b]);
            if (rc) { cleanup(); return rc; }
        }
        STREAM_HIP(hipEventRecord(cmp_done[b], ctx->stream));
        STREAM_HIP(hipStreamWaitEvent(s_dn, cmp_done[b], 0));
        if (h_grid_a) {
            STREAM_HIP(hipMemcpyAsync(h_grid_a + (size_t)k * grid_b, d_ga[b], grid_b, hipMemcpyDeviceToHost, s_dn));
            if (mode == 0) STREAM_HIP(hipMemcpyAsync(h_grid_b + (size_t)k * grid_b, d_gb[b], grid_b, hipMemcpyDeviceToHost, s_dn));
        }
        if (h_db) STREAM_HIP(hipMemcpyAsync(h_db + (size_t)f0 * n, d_db[b], (size_t)cnt * n * sizeof(TR), hipMemcpyDeviceToHost, s_dn));
        STREAM_HIP(hipMemcpyAsync(h_line_a + (size_t)f0 * disp_w, d_la[b], (size_t)cnt * disp_w, hipMemcpyDeviceToHost, s_dn));
        if (mode == 0) STREAM_HIP(hipMemcpyAsync(h_line_b + (size_t)f0 * disp_w, d_lb[b], (size_t)cnt * disp_w, hipMemcpyDeviceToHost, s_dn));
        STREAM_HIP(hipMemcpyAsync(h_pcm + (size_t)f0 * n_out * 2, d_pcm[b], (size_t)cnt * n_out * 2 * sizeof(int16_t), hipMemcpyDeviceToHost, s_dn));
        STREAM_HIP(hipEventRecord(dn_done[b], s_dn));
    }
    STREAM_HIP(hipStreamSynchronize(ctx->stream));
    if (h_row_lo) STREAM_HIP(hipMemcpy(h_row_lo, d_lo + n_halo, sizeof(TR) * n_frames, hipMemcpyDeviceToHost));
    if (h_row_hi) STREAM_HIP(hipMemcpy(h_row_hi, d_hi + n_halo, sizeof(TR) * n_frames, hipMemcpyDeviceToHost));
#undef STREAM_HIP
    cleanup();
    return PSS_OK;
}

extern "C" int pss_h_stream_display_nfm(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames,
                                        int mode, int window, int disp_h, int disp_w, const float *h_halo_lo,
                                        const float *h_halo_hi, int n_halo, int8_t *h_line_a, int8_t *h_line_b, int16_t *h_pcm,
                                        float *h_db, float *h_row_lo, float *h_row_hi)
{
    return stream_display<float>(ctx, h_iq, n_frames, n, fs, chunk_frames, mode, window, disp_h, disp_w, h_halo_lo, h_halo_hi, n_halo, h_line_a,
                                 h_line_b, h_pcm, h_db, h_row_lo, h_row_hi, nullptr, nullptr);
}

// ... with compute_fft's own float64 rows from the transform to the cells: the lines (and grids) the reference draws from this capture
extern "C" int pss_h_stream_display_nfm_f64(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames,
                                            int mode, int window, int disp_h, int disp_w, const double *h_halo_lo,
                                            const double *h_halo_hi, int n_halo, int8_t *h_line_a, int8_t *h_line_b, int16_t *h_pcm,
                                            double *h_db, double *h_row_lo, double *h_row_hi, int8_t *h_grid_a, int8_t *h_grid_b)
{
    if (ctx && (n < 16 || n > 65536 || (n & (n - 1)))) return pss_fail(ctx, PSS_E_ARG, "pss_h_stream_display_nfm_f64: n must be a power of two in [16, 65536]");
    if (ctx && h_grid_a && n_halo > 0) return pss_fail(ctx, PSS_E_ARG, "pss_h_stream_display_nfm_f64: grids need a fresh history (no halo)");
    return stream_display<double>(ctx, h_iq, n_frames, n, fs, chunk_frames, mode, window, disp_h, disp_w, h_halo_lo, h_halo_hi, n_halo, h_line_a,
                                  h_line_b, h_pcm, h_db, h_row_lo, h_row_hi, h_grid_a, h_grid_b);
}

extern "C" int pss_h_stream_display_nfm_grids(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames,
                                              int mode, int window, int disp_h, int disp_w, int8_t *h_line_a, int8_t *h_line_b,
                                              int16_t *h_pcm, float *h_row_lo, float *h_row_hi, int8_t *h_grid_a, int8_t *h_grid_b)
{
    if (ctx && !h_grid_a) return pss_fail(ctx, PSS_E_ARG, "pss_h_stream_display_nfm_grids: h_grid_a is null");
    return stream_display<float>(ctx, h_iq, n_frames, n, fs, chunk_frames, mode, window, disp_h, disp_w, nullptr, nullptr, 0, h_line_a, h_line_b,
                                 h_pcm, nullptr, h_row_lo, h_row_hi, h_grid_a, h_grid_b);
}
