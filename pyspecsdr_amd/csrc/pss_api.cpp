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
