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

int pss_ensure_scratch(pss_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes) return PSS_OK;
    if (ctx->scratch) {
        hipStreamSynchronize(ctx->stream);
        hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    hipError_t e = hipMalloc(&ctx->scratch, bytes);
    if (e != hipSuccess) return pss_fail(ctx, PSS_E_NOMEM, std::string("scratch hipMalloc: ") + hipGetErrorString(e));
    ctx->scratch_bytes = bytes;
    return PSS_OK;
}

void pss_time_begin(pss_ctx *ctx)
{
    if (ctx->timing && ctx->tdepth++ == 0) hipEventRecord(ctx->ev0, ctx->stream);
}
void pss_kernel_begin(pss_ctx *ctx, const char *name)
{
    if (!ctx->timing) return;
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
    if (!ctx->timing) return;
    hipEventRecord(ctx->krecs[ctx->kused].e1, PSS_STREAM(ctx));
    ctx->kused++;
}
void pss_time_end(pss_ctx *ctx)
{
    if (ctx->timing && --ctx->tdepth == 0) { hipEventRecord(ctx->ev1, ctx->stream); ctx->last_ms = 0.0f; }
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
    e = hipSetDevice(device);
    if (e != hipSuccess) return pss_fail(nullptr, PSS_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    pss_ctx *ctx = new pss_ctx();
    ctx->device = device;
    { const char *e = getenv("PSS_NO_FUSED"); ctx->no_fused = e && e[0] == '1'; }
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return pss_fail(nullptr, PSS_E_HIP, "hipStreamCreate failed"); }
    ctx->own_stream = true;
    hipEventCreate(&ctx->ev0);
    hipEventCreate(&ctx->ev1);
    hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming);
    *out = ctx;
    return PSS_OK;
}

extern "C" void pss_destroy(pss_ctx *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->tw) hipFree(kv.second);
    for (auto &kv : ctx->win) hipFree(kv.second);
    for (auto &kv : ctx->plans) {
        hipFree(kv.second.d_leaf_off); hipFree(kv.second.d_leaf_len); hipFree(kv.second.d_node_l);
        hipFree(kv.second.d_node_r); hipFree(kv.second.d_level_start);
    }
    if (ctx->scratch) hipFree(ctx->scratch);
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    hipStreamSynchronize(ctx->stream2);
    hipStreamDestroy(ctx->stream2);
    hipEventDestroy(ctx->ev_fork);
    hipEventDestroy(ctx->ev_join);
    for (auto &k : ctx->krecs) { hipEventDestroy(k.e0); hipEventDestroy(k.e1); }
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int pss_set_stream(pss_ctx *ctx, void *hip_stream)
{
    if (!ctx) return PSS_E_ARG;
    hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    ctx->own_stream = false;
    return PSS_OK;
}

extern "C" int pss_set_option(pss_ctx *ctx, const char *key, int value)
{
    if (!ctx || !key) return PSS_E_ARG;
    if (!strcmp(key, "nfm_fused")) { ctx->no_fused = (value == 0); return PSS_OK; }
    return pss_fail(ctx, PSS_E_ARG, std::string("unknown option: ") + key);
}

extern "C" int pss_sync(pss_ctx *ctx)
{
    if (!ctx) return PSS_E_ARG;
    return pss_hip_check(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
}

extern "C" const char *pss_last_error(pss_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int pss_enable_timing(pss_ctx *ctx, int on)
{
    if (!ctx) return PSS_E_ARG;
    ctx->timing = on != 0;
    ctx->tdepth = 0;
    ctx->kused = 0;
    ctx->last_ms = -1.0f;
    return PSS_OK;
}

extern "C" float pss_last_kernel_ms(pss_ctx *ctx)
{
    if (!ctx || !ctx->timing || ctx->last_ms < 0.0f) return -1.0f;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
    return ms;
}

extern "C" int pss_kernel_times(pss_ctx *ctx, char *buf, int buf_len)
{
    // "name=ms;name=ms;..." for every kernel launched since timing was enabled / since the previous read
    if (!ctx || !buf || buf_len < 1) return PSS_E_ARG;
    std::string out;
    if (ctx->timing && ctx->last_ms >= 0.0f) {
        hipEventSynchronize(ctx->ev1);
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
namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(pss_ctx *ctx, size_t bytes) { return pss_hip_check(ctx, hipMalloc(&p, bytes ? bytes : 1), "hipMalloc"); }
};
}  // namespace

extern "C" int pss_h_compute_fft(pss_ctx *ctx, const float *h_iq, int n, double *h_db)
{
    if (!ctx) return PSS_E_ARG;
    if (!h_iq || !h_db || n < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    DevBuf iq, db;
    int r;
    if ((r = iq.alloc(ctx, sizeof(float) * 2 * n)) || (r = db.alloc(ctx, sizeof(float) * n))) return r;
    PSS_HIP(ctx, hipMemcpyAsync(iq.p, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_spectrum_db(ctx, (const float *)iq.p, 1, n, (float *)db.p);
    if (r) return r;
    std::vector<float> tmp(n);
    PSS_HIP(ctx, hipMemcpyAsync(tmp.data(), db.p, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n; i++) h_db[i] = (double)tmp[i];
    return PSS_OK;
}

extern "C" int pss_h_demodulate(pss_ctx *ctx, int mode, const float *h_iq, int n, double fs, double *h_audio_stereo,
                                int16_t *h_pcm)
{
    if (!ctx) return PSS_E_ARG;
    if (!h_iq || n < 1 || (!h_audio_stereo && !h_pcm)) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    int n_out = pss_demod_out_len(mode, n, fs);
    if (n_out < 0) return pss_fail(ctx, PSS_E_ARG, "unknown mode or sample rate below 22050 Hz");
    DevBuf iq, pcm, au;
    int r;
    if ((r = iq.alloc(ctx, sizeof(float) * 2 * n)) || (r = pcm.alloc(ctx, sizeof(int16_t) * 2 * n_out)) ||
        (r = au.alloc(ctx, sizeof(double) * n_out)))
        return r;
    PSS_HIP(ctx, hipMemcpyAsync(iq.p, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_demod(ctx, mode, (const float *)iq.p, 1, n, fs, (int16_t *)pcm.p, (double *)au.p);
    if (r) return r;
    std::vector<double> mono(n_out);
    PSS_HIP(ctx, hipMemcpyAsync(mono.data(), au.p, sizeof(double) * n_out, hipMemcpyDeviceToHost, ctx->stream));
    if (h_pcm) PSS_HIP(ctx, hipMemcpyAsync(h_pcm, pcm.p, sizeof(int16_t) * 2 * n_out, hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h_audio_stereo)
        for (int i = 0; i < n_out; i++) h_audio_stereo[2 * i] = h_audio_stereo[2 * i + 1] = mono[i];  // mono_to_stereo
    return PSS_OK;
}

extern "C" int pss_h_measure_power(pss_ctx *ctx, const float *h_iq, int n, float *h_power)
{
    if (!ctx) return PSS_E_ARG;
    if (!h_iq || !h_power || n < 1) return pss_fail(ctx, PSS_E_ARG, "bad arguments");
    DevBuf iq, p;
    int r;
    if ((r = iq.alloc(ctx, sizeof(float) * 2 * n)) || (r = p.alloc(ctx, sizeof(float)))) return r;
    PSS_HIP(ctx, hipMemcpyAsync(iq.p, h_iq, sizeof(float) * 2 * n, hipMemcpyHostToDevice, ctx->stream));
    r = pss_power_db(ctx, (const float *)iq.p, 1, n, (float *)p.p);
    if (r) return r;
    PSS_HIP(ctx, hipMemcpyAsync(h_power, p.p, sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PSS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PSS_OK;
}

// ---- streamed capture --------------------------------------------------------------------------------
extern "C" void *pss_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void pss_host_free(void *p)
{
    if (p) hipHostFree(p);
}

extern "C" int pss_h_stream_spectrum_nfm(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames,
                                         float *h_db, int16_t *h_pcm)
{
    if (!ctx) return PSS_E_ARG;
    if (!h_iq || !h_pcm || n_frames < 0 || n < 1 || chunk_frames < 1) return pss_fail(ctx, PSS_E_ARG, "bad stream arguments");
    const int n_out = pss_demod_out_len(PSS_MODE_NFM, n, fs);
    if (n_out < 0) return pss_fail(ctx, PSS_E_ARG, "sample rate below 22050 Hz");
    if (n_frames == 0) return PSS_OK;
    if (chunk_frames > n_frames) chunk_frames = n_frames;
    const size_t iq_b = (size_t)chunk_frames * n * 2 * sizeof(float), db_b = (size_t)chunk_frames * n * sizeof(float),
                 pcm_b = (size_t)chunk_frames * n_out * 2 * sizeof(int16_t);
    hipStream_t s_up = nullptr, s_dn = nullptr;
    hipEvent_t up_done[2] = {nullptr, nullptr}, cmp_done[2] = {nullptr, nullptr}, dn_done[2] = {nullptr, nullptr};
    void *d_iq[2] = {nullptr, nullptr}, *d_db[2] = {nullptr, nullptr}, *d_pcm[2] = {nullptr, nullptr};
    int rc = PSS_OK;
    auto cleanup = [&]() {
        hipStreamSynchronize(ctx->stream);
        if (s_up) { hipStreamSynchronize(s_up); hipStreamDestroy(s_up); }
        if (s_dn) { hipStreamSynchronize(s_dn); hipStreamDestroy(s_dn); }
        for (int i = 0; i < 2; i++) {
            if (up_done[i]) hipEventDestroy(up_done[i]);
            if (cmp_done[i]) hipEventDestroy(cmp_done[i]);
            if (dn_done[i]) hipEventDestroy(dn_done[i]);
            if (d_iq[i]) hipFree(d_iq[i]);
            if (d_db[i]) hipFree(d_db[i]);
            if (d_pcm[i]) hipFree(d_pcm[i]);
        }
    };
#define STREAM_HIP(call)                                          \
    do {                                                          \
        rc = pss_hip_check(ctx, (call), #call);                   \
        if (rc) { cleanup(); return rc; }                         \
    } while (0)
    STREAM_HIP(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
    STREAM_HIP(hipStreamCreateWithFlags(&s_dn, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        STREAM_HIP(hipEventCreateWithFlags(&up_done[i], hipEventDisableTiming));
        STREAM_HIP(hipEventCreateWithFlags(&cmp_done[i], hipEventDisableTiming));
        STREAM_HIP(hipEventCreateWithFlags(&dn_done[i], hipEventDisableTiming));
        STREAM_HIP(hipMalloc(&d_iq[i], iq_b));
        STREAM_HIP(hipMalloc(&d_db[i], db_b));
        STREAM_HIP(hipMalloc(&d_pcm[i], pcm_b));
    }
    const long n_chunks = (n_frames + chunk_frames - 1) / chunk_frames;
    for (long k = 0; k < n_chunks; k++) {
        const int b = (int)(k & 1);
        const long f0 = k * chunk_frames;
        const long cnt = (n_frames - f0) < chunk_frames ? (n_frames - f0) : chunk_frames;
        // the buffer set is free once chunk k-2's download has finished
        if (k >= 2) STREAM_HIP(hipStreamWaitEvent(s_up, dn_done[b], 0));
        STREAM_HIP(hipMemcpyAsync(d_iq[b], h_iq + (size_t)f0 * n * 2, (size_t)cnt * n * 2 * sizeof(float),
                                  hipMemcpyHostToDevice, s_up));
        STREAM_HIP(hipEventRecord(up_done[b], s_up));
        STREAM_HIP(hipStreamWaitEvent(ctx->stream, up_done[b], 0));
        if (k >= 2) STREAM_HIP(hipStreamWaitEvent(ctx->stream, dn_done[b], 0));
        rc = pss_spectrum_nfm(ctx, (const float *)d_iq[b], cnt, n, fs, (float *)d_db[b], (int16_t *)d_pcm[b]);
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
