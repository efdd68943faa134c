/*
 * pss.h — C ABI of libpss.so, the MI355X (gfx950) implementation of PySpecSDR's IQ -> spectrum + demod
 * hot path.  Plain C: pointers and sizes only, no torch / HIP types in any signature.
 *
 * What this replaces.  The reference (xqtr/PySpecSDR, pure Python) has no FFI; its hot path is the module
 * namespace `from signal_processing import *` (pyspecsdr.py:98).  Each entry point below names the
 * reference callable it replaces; pyspecsdr_amd/signal_processing.py is the ctypes binding that presents
 * those callables with their original Python signatures (INTEGRATION.md shows the two-line patch).
 *
 * Conventions
 *   - iq: interleaved complex64 (I0,Q0,I1,Q1,...) — the SoapySDR CF32 buffer of SDRDevice.read_samples
 *     (pyspecsdr.py:1885-1891).  Batched calls take n_frames contiguous frames of n samples each.
 *   - `d_` parameters are DEVICE pointers (hipMalloc / torch tensors' data_ptr); `h_` are host pointers.
 *   - All functions return 0 on success or a negative pss_status; pss_last_error() gives the text.
 *     Nothing aborts or throws.  A missing GPU is an error (PSS_E_HIP), never a CPU fallback.
 *   - A context is bound to one device and one stream; calls on a context are stream-ordered and must
 *     not be issued concurrently from several threads.  Batched calls are asynchronous w.r.t. the host
 *     unless stated; call pss_sync() (or sync the stream you supplied) before reading results.
 */
#ifndef PSS_H
#define PSS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct pss_ctx pss_ctx;

typedef enum {
    PSS_OK = 0,
    PSS_E_ARG = -1,      /* bad argument (size not supported, null pointer, ...) */
    PSS_E_HIP = -2,      /* HIP runtime error / no device */
    PSS_E_PADLEN = -3,   /* NFM: n-1 <= 27 -> the reference raises ValueError (scipy sosfiltfilt padlen)   */
    PSS_E_CUTOFF = -4,   /* NFM/SSB: cutoff >= Nyquist -> the reference raises ValueError (scipy firwin)  */
    PSS_E_NOMEM = -5,
    PSS_E_COMM = -6      /* multi-GPU exchange: librccl not found, no communicator, or an RCCL error (text in pss_last_error) */
} pss_status;

typedef enum { PSS_MODE_NFM = 0, PSS_MODE_AM = 1, PSS_MODE_USB = 2, PSS_MODE_LSB = 3, PSS_MODE_WFM = 4 } pss_mode;

/* ---- context ------------------------------------------------------------------------------------ */
int pss_create(int device, pss_ctx **out);
void pss_destroy(pss_ctx *ctx);
/* Use an existing hipStream_t (passed as void*) instead of the context's own stream; NULL = default stream. */
int pss_set_stream(pss_ctx *ctx, void *hip_stream);
/* The hipStream_t (as void*) the context's work is queued on: lets a caller order its own streams / events against it
 * (e.g. torch.cuda.ExternalStream(pss_get_stream(ctx)) to overlap an RCCL gather with the next batch). */
void *pss_get_stream(pss_ctx *ctx);
/* Order the context's stream against a caller's stream (hipStream_t as void*, NULL = the default stream) WITHOUT adopting it — the context
 * keeps its own non-blocking stream, which the legacy default stream does not synchronise with:
 *   pss_order_after:  work queued on the context from now on starts after everything queued on the caller's stream so far (inputs the
 *                     caller produced, outputs it cleared);
 *   pss_order_before: work queued on the caller's stream from now on starts after everything the context has queued so far (results).
 * One event record + one stream wait each (~2 us), nothing blocks the host.  A single-threaded caller that fills buffers on its stream,
 * calls an entry point and reads the results on its stream — the call order of the reference's loop, pyspecsdr.py:2250-2283 — brackets every
 * call with the pair; pyspecsdr_amd.engine.Engine does so by default with torch's current stream. */
int pss_order_after(pss_ctx *ctx, void *hip_stream);
int pss_order_before(pss_ctx *ctx, void *hip_stream);
int pss_sync(pss_ctx *ctx);
const char *pss_last_error(pss_ctx *ctx); /* ctx may be NULL: error of the last failed pss_create */
int pss_device_count(void);
/* Switches.  Returns PSS_E_ARG for unknown keys.  Every alternative path produces the same int16 samples and float64 audio.
 *   "nfm_fused" (1)            0: lane-per-frame three-kernel NFM path (front, edge, iir) instead of the fused kernels (test fallback)
 *   "wfm_fused" (1)            0: k_wfm_front + lane-per-frame decimator instead of the fused WFM forward kernel (test fallback)
 *   "small_batch" (1)          0: never take the latency-oriented small-batch kernels (one lane per filter section)
 *   "small_batch_max" (8192)   largest NFM frame count that takes them;  "wfm_small_batch_max" (6000) likewise for WFM
 *   "ssb_hilbert" (1)          0: demodulate_ssb skips the reference's hilbert() FFT round trip (the identity on the real part it
 *                              keeps, up to ~1e-16); 1: executed for power-of-two frames of 256..1048576 samples
 *   "hilbert_exact" (0)        1: pss_hilbert and demodulate_ssb's hilbert() run pocketfft's own butterfly order for rows of 256..1048576
 *                              samples (real radix-4 / radix-2 forward passes, complex radix-8 / 4 / 2 inverse passes, its twiddle products):
 *                              the analytic signal and the SSB float64 audio then equal SciPy's on every bit.  0 (default): the register
 *                              transforms, within 2e-14 of it (int16 PCM equal either way) and about twice as fast
 *   "db_exact" (0)             1: compute_fft's dB rows (pss_spectrum_db and everything built on it) are evaluated to float64 accuracy and
 *                              rounded once: the float32 row then IS the float32 rounding of the reference's float64 row (all 69 490
 *                              golden values; measured on FM frames: 5 of 8.4 million bins off by one ulp, where the value lies within
 *                              ~1e-15 of a rounding boundary).  0 (default): float32 evaluation of the logarithm, 1-2 ulp from that
 *                              (the contract is 1e-4 relative); the exact evaluation costs the spectrum kernels 15-25 % (0.17 -> 0.20 ms
 *                              at cfg 2, 0.75 -> 0.91 ms at 8192 x 16384) and the bench step 3 %
 *   "scan_exact" (1)           0: scanner slices (pss_scan, pss_scan_threshold) get their dB values from compute_fft's float64 / hardware-log2
 *                              evaluation (1e-4 relative; 30 % faster at 8192 x 4096) instead of NumPy's float32 chain bit for bit
 * Kernel-selection knobs of earlier rounds' A/B measurements ("fft_split", "fft_prefetch", "fft_xl4096", "fft_big_scratch", "post_legacy",
 * "post_sort_max") exist only in builds with -DPSS_VARIANTS (tools/build_variant.py); the experiments that lost — the FIR on the matrix
 * pipe, discriminator rows handed from the spectrum kernel, the fused spectrum + post-process kernel, the 112-VGPR spectrum kernel, the
 * alternative pipeline schedules, CU-mask partitioning — are documented with their measurements in DESIGN.md and no longer compiled in. */
int pss_set_option(pss_ctx *ctx, const char *key, int value);

/* ---- filter design (host side, pure C++; replaces the per-call SciPy design work) --------------
 * The designers restate SciPy's / NumPy's arithmetic operation by operation (DESIGN.md par. 2): firwin, cheby1 as scipy.signal.decimate
 * calls it, sosfilt_zi and butter return SciPy 1.15's tables bit for bit at every cutoff / decimation factor / sample rate tried (the pre-warp
 * tan is NumPy's SVML routine restated, pss_h_np_f64). */
/* scipy.signal.firwin(numtaps, cutoff) low-pass, Hamming window, cutoff normalised to Nyquist
 * (signal_processing.py:107 and :203/:208).  Returns PSS_E_CUTOFF unless 0 < cutoff < 1. */
int pss_design_firwin(int numtaps, double cutoff, double *taps);
/* scipy.signal.cheby1(order, rp_db, wn, output='sos') low-pass, order even (decimate(): order 8, rp 0.05,
 * wn 0.8/q — signal_processing.py:112 via scipy _signaltools.py:4831).  sos[order/2][6]. */
int pss_design_cheby1_sos(int order, double rp_db, double wn, double *sos);
/* scipy.signal.butter(order, Wn, output='sos'): low-pass with cutoff wn_high when wn_low <= 0 (as bandpass_filter does,
 * signal_processing.py:37-39), else band-pass [wn_low, wn_high] (:41).  sos[ceil(order/2)][6] (low) / sos[order][6] (band);
 * *nsec (nullable) receives the row count.  PSS_E_CUTOFF where SciPy raises ValueError (Wn outside 0 < Wn < 1). */
int pss_design_butter_sos(int order, double wn_low, double wn_high, double *sos, int *nsec);
/* scipy.signal.sosfilt_zi(sos).  zi[nsec][2]. */
int pss_design_sosfilt_zi(const double *sos, int nsec, double *zi);
/* NumPy's float64 tan (op 0) / exp (op 1) as the reference's SciPy calls evaluate them (NumPy's AVX512_SKX dispatch: SVML, an ulp from libm
 * on 0.5 % / 5 % of arguments) — the two primitives the designers need beyond libm; host arrays. */
int pss_h_np_f64(int op, const double *x, long n, double *out);
/* butter(5, [300, 3000]/(22050/2), 'band', output='sos') — the fixed AM filter (signal_processing.py:188-191,
 * :39-41; pyspecconst.py:3,5).  sos[5][6]. */
int pss_am_bandpass_sos(double *sos);

/* Override the designed coefficients for a sample rate (e.g. with tables captured from SciPy). Optional. */
int pss_set_nfm_filters(pss_ctx *ctx, double fs, const double *taps65, const double *sos4x6, const double *zi4x2);
int pss_set_ssb_taps(pss_ctx *ctx, double fs, const double *taps65);
/* Read back the coefficients the context uses for fs (designing them on first use). */
int pss_get_nfm_filters(pss_ctx *ctx, double fs, double *taps65, double *sos4x6, double *zi4x2);
int pss_get_ssb_taps(pss_ctx *ctx, double fs, double *taps65);

/* ---- batched device entry points --------------------------------------------------------------- */
/* compute_fft (signal_processing.py:243-264) for n_frames frames: Hamming window, n_fft-point FFT (float64
 * butterflies), fftshift, 10*log10(|X|^2 + 1e-10).  d_db: float32 [n_frames][n_fft].
 * n_fft: a power of two in [16, 1048576] (register / LDS radix-16 kernels), or any other length in [2, 524288]
 * (Bluestein's algorithm over the two-pass 2^17..2^20-point transform: NumPy's fft takes any length too). */
int pss_spectrum_db(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, float *d_db);
/* Caller-side post-process (pyspecsdr.py:2278-2283): 5-tap moving average ('valid') then clamp below
 * median-10.  d_post: float32 [n_frames][n_fft-4]. */
int pss_spectrum_post(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_post);
/* The same, and the finite minimum / maximum of every post-processed row (float32 [n_frames] each; (+inf, -inf) for a row
 * without a finite value): what the display accumulators below normalise with. */
int pss_spectrum_post_extremes(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_post, float *d_row_lo,
                               float *d_row_hi);
/* compute_fft and the post-process of the same frames in one call (d_db, d_post, and — both or neither — the row extremes): two launches. */
int pss_spectrum_db_post(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, float *d_db, float *d_post, float *d_row_lo,
                         float *d_row_hi);
/* Finite extremes of arbitrary rows (np.min / np.max over all_data[np.isfinite(all_data)], pyspecsdr.py:1356-1358, per row). */
int pss_row_extremes(pss_ctx *ctx, const float *d_rows, long n_rows, int len, float *d_row_lo, float *d_row_hi);
int pss_row_extremes_f64(pss_ctx *ctx, const double *d_rows, long n_rows, int len, double *d_row_lo, double *d_row_hi);
/* Inline scanner slice (pyspecsdr.py:2542-2552): unwindowed FFT, dB, peak, 20-dB-down bin count,
 * bandwidth = count * fs / n_fft.  d_db float32 [n][n_fft] (may be NULL), d_peak float32 [n],
 * d_bw float64 [n], d_count int32 [n] (may be NULL).  n_fft: power of two in [16, 16384] (one kernel), or any other
 * length in [2, 524288] (Bluestein + a reduction kernel).  The dB values are the reference's float32 values bit for bit (so are
 * peak, count and bandwidth): np.fft.fft on complex64 is a double transform rounded to complex64 (NumPy 2.2), everything behind
 * it float32 arithmetic that is modelled exactly; a bin can differ only if a float64 component lies within the transform's ~1e-16 * max|X|
 * of a float32 rounding boundary (weak bins beside a strong carrier: ~1e-6 of the bins) (option "scan_exact" = 0: the 1e-4-relative evaluation of pss_spectrum_db, 30 % faster). */
int pss_scan(pss_ctx *ctx, const float *d_iq, long n_slices, int n_fft, double fs, float *d_db, float *d_peak,
             double *d_bw, int32_t *d_count);
/* The sweep driver's per-read arithmetic (scan_frequencies, pyspecsdr.py:1049-1057): unwindowed fft of a read of
 * n = int(0.1 * fs) samples (240 000 at 2.4 MS/s: not a power of two), dB, max_power, bins above the ABSOLUTE threshold,
 * bandwidth = count * fs / n.  Buffers as for pss_scan (d_db, d_peak, d_bw, d_count may each be NULL). */
int pss_scan_threshold(pss_ctx *ctx, const float *d_iq, long n_slices, int n, double fs, double threshold_db, float *d_db,
                       float *d_peak, double *d_bw, int32_t *d_count);
/* iq_correction — signal_processing.py:46-80, per frame: DC removal, IQ amplitude/phase balance, power restore; float32
 * throughout, bit-identical to the reference on NumPy 2.2.  d_out_iq: complex64 [n_frames][n] (nullable);
 * d_raw: float32 [n_frames][n] = real part = demodulate_signal(..., 'RAW') (signal_processing.py:222-238) (nullable). */
int pss_iq_correction(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_out_iq, float *d_raw);

/* scipy.signal.hilbert along n_rows float64 rows of n samples (the analytic signal demodulate_ssb builds,
 * signal_processing.py:205, :210): fft, one-sided mask, ifft.  n: a power of two in [256, 1048576] (PSS_E_ARG otherwise) — up to
 * 16384 both transforms in one kernel; longer rows (the reference's read buffers: 32768 by default, up to 2^20) through a complex
 * float64 spectrum in HBM.  d_analytic: complex128 [n_rows][n] (interleaved re, im). */
int pss_hilbert(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_analytic);

/* measure_signal_power (signal_processing.py:325-328): float32 [n_frames]. */
int pss_power_db(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_power);
/* adjust_gain (pyspecsdr.py:898-919) run sequentially over a series of power readings:
 * d_idx_out[i] = gain index after reading i, starting from start_idx. */
int pss_agc_steps(pss_ctx *ctx, const float *d_power, long n, int start_idx, int n_gains, int32_t *d_idx_out);

/* demodulate_signal (signal_processing.py:220-240) for NFM / AM / USB / LSB, every frame independently
 * (filter state reset and peak normalisation per frame, exactly like the reference's per-buffer calls).
 *   n_out per frame = pss_demod_out_len(mode, n, fs): NFM ceil((n-1)/int(fs/22050)), AM/SSB n.
 *   d_pcm   int16  [n_frames][n_out][2]  — np.int16(audio*32767), L=R  (io_manager.py:25-26); may be NULL
 *   d_audio double [n_frames][n_out]     — the mono float64 audio before stereo duplication; may be NULL */
int pss_demod(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, int16_t *d_pcm,
              double *d_audio);
int pss_demod_out_len(int mode, int n, double fs);
/* demodulate_nfm / demodulate_wfm's `target_rate` argument (signal_processing.py:91, :119; default DEFAULT_SAMPLE_RATE = 22050): the decimation
 * factor is int(sample_rate / target_rate) (:111).  pss_set_target_rate changes it for the context (and drops the cached decimator designs);
 * pss_demod_out_len_ctx is pss_demod_out_len at the context's target rate, pss_demod_out_len_rate at an explicit one.
 * Every factor >= 1 is served: NFM runs decimate(x, 1) at factor 1 (as the reference does); WFM at factor 1 skips the decimate() stage as
 * the reference does (:152-155) and normalises the de-emphasised channels (n - 1 samples per channel; plain kernels). */
int pss_set_target_rate(pss_ctx *ctx, double target_rate);
int pss_demod_out_len_ctx(pss_ctx *ctx, int mode, int n, double fs);
int pss_demod_out_len_rate(int mode, int n, double fs, double target_rate);

/* PSS_MODE_WFM = demodulate_wfm (signal_processing.py:119-176): discriminator, L+R / pilot / L-R Butterworth branches,
 * 75 us de-emphasis, zero-phase decimation, joint peak normalisation.  n_out = ceil((n-1)/int(fs/22050)); here
 *   d_audio double [n_frames][n_out][2] holds np.column_stack((left, right)) and d_pcm its int16 image.
 * pss_demod() runs the demodulator on the samples as given; pss_demod_signal() is the reference's dispatcher
 * demodulate_signal (:220-240): identical for NFM/AM/USB/LSB, and for WFM it applies iq_correction first (:222-225). */
int pss_demod_signal(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, int16_t *d_pcm,
                     double *d_audio);
/* measure_signal_power (signal_processing.py:325-328) and demodulate of the same read buffers, as the main loop runs them back to back
 * (pyspecsdr.py:2251, :2262).  d_power float32 [n_frames]: pss_power_db's bits; d_pcm / d_audio: pss_demod's.  For AM the power and the
 * demodulator's mean of |x| are reduced in ONE pass over the IQ; for the other modes this is the two calls in order. */
int pss_demod_power(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, int16_t *d_pcm, double *d_audio,
                    float *d_power);
/* WFM filter set of one sample rate: lp = butter(5, 15000/(fs/2)) [3][6], pilot = butter(5, [18800,19200]/(fs/2), 'band')
 * [5][6], lmr = butter(5, [23000,53000]/(fs/2), 'band') [5][6], alpha = exp(-1/(75e-6 fs)).  Designed on first use
 * (pss_design_butter_sos and NumPy's exp restated: SciPy 1.15's / NumPy's bits); set_ lets a caller inject another SciPy build's tables. */
int pss_set_wfm_filters(pss_ctx *ctx, double fs, const double *lp3x6, const double *pilot5x6, const double *lmr5x6,
                        double alpha);
int pss_get_wfm_filters(pss_ctx *ctx, double fs, double *lp3x6, double *pilot5x6, double *lmr5x6, double *alpha);

/* Headline fused call: spectrum + NFM demod of the same frames (BASELINE.json metric). */
int pss_spectrum_nfm(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double fs, float *d_db,
                     int16_t *d_pcm);

/* One iteration of the reference's main loop (pyspecsdr.py:2262-2283 and the waterfall draw) for a batch of read buffers:
 * NFM demod -> d_pcm; compute_fft -> d_db [n_frames][n]; post-process -> d_post [n_frames][n-4] and the row extremes
 * (d_row_lo / d_row_hi: [n_halo + n_frames], the first n_halo entries supplied by the caller as for pss_waterfall_rows);
 * waterfall line per frame -> d_glyph / d_colour [n_frames][disp_w].  The same results as the separate calls; part of the
 * display chain runs on a side stream beside the demodulator's backward pass and is joined before the call returns.
 * d_post may be NULL: the post-processed rows are then not written to memory at all (a third of the chain's HBM traffic) — the
 * display lines are the same bytes, rebuilt from the dB rows and the 12 bytes per row that pss_spectrum_post_thresholds leaves. */
int pss_frame_pipeline_nfm(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double fs, float *d_db, float *d_post,
                           float *d_row_lo, float *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph,
                           int8_t *d_colour, int16_t *d_pcm);

/* The same iteration in ANY of the main loop's demodulation modes — demodulate_signal(samples, fs, CURRENT_DEMOD) (pyspecsdr.py:2262), whose
 * default is WFM (:2855) — and for either batched display accumulator.  mode: PSS_MODE_*, with demodulate_signal's dispatcher semantics (WFM
 * frames are IQ-corrected first, signal_processing.py:222-225; compute_fft sees the samples as read, pyspecsdr.py:2275).  d_pcm int16
 * [n_frames][pss_demod_out_len(mode, n, fs)][2].  display 0: waterfall line, d_line_a = glyph, d_line_b = colour (window: 30 in the reference);
 * display 1: persistence trace, d_line_a = row index per column, d_line_b unused (may be NULL), disp_h <= 127 (window: 10 in the reference).
 * The other arguments and the results are those of the separate calls (pss_demod_signal, pss_spectrum_db, pss_spectrum_post_extremes /
 * _thresholds, pss_waterfall_rows[_db] / pss_persistence_rows[_db]); NFM runs the schedule of pss_frame_pipeline_nfm, the other modes run the
 * display chain on the side stream beside the whole demodulator.  PSS_E_ARG for n < 8 (no post-processed row), n_halo < 0, window < 1. */
int pss_frame_pipeline(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, float *d_db, float *d_post,
                       float *d_row_lo, float *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w, int8_t *d_line_a,
                       int8_t *d_line_b, int16_t *d_pcm);

/* The reference's own row type.  compute_fft returns float64 rows (signal_processing.py:243-264) and the caller smooths, clamps and
 * draws them in float64 (pyspecsdr.py:2278-2283, :1342-1406); the float32 rows above agree with them to 1e-7 relative, but a display cell
 * can differ where a value sits on a quantisation edge (<= 2e-3 of the cells).  These entry points keep float64 throughout, so the
 * display lines are the reference's cells.  Since round 5 they run on the register kernels of the float32 calls (float64 dB values stored from
 * the transform's registers for 256 .. 4096 points, register select on 64-bit keys up to 16 388 points; option "f64_plain" = 1: the plain
 * round-3 kernels) and pss_frame_pipeline_nfm_f64 is the step bench.py times.  n_fft: a power of two in [16, 65536].
 *   pss_spectrum_db_f64:   d_db float64 [n_frames][n_fft] = 10 log10(|fftshift(fft(iq * hamming))|^2 + 1e-10)
 *   pss_spectrum_post_f64: d_post float64 [n_frames][n_fft - 4]; d_row_lo / d_row_hi [n_frames] (both or neither): finite extremes
 *   pss_frame_pipeline_nfm_f64: pss_frame_pipeline_nfm with float64 rows, extremes (and halo) */
int pss_spectrum_db_f64(pss_ctx *ctx, const float *d_iq, long n_frames, int n_fft, double *d_db);
int pss_spectrum_post_f64(pss_ctx *ctx, const double *d_db, long n_frames, int n_fft, double *d_post, double *d_row_lo,
                          double *d_row_hi);
int pss_frame_pipeline_nfm_f64(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double fs, double *d_db, double *d_post,
                               double *d_row_lo, double *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph,
                               int8_t *d_colour, int16_t *d_pcm);
/* pss_frame_pipeline (any demodulation mode, waterfall line or persistence trace) with float64 rows: the cells of either batched accumulator
 * as the reference draws them from this IQ.  d_post may be NULL (rows of up to 16 388 points are then never materialised). */
int pss_frame_pipeline_f64(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, double *d_db, double *d_post,
                           double *d_row_lo, double *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w, int8_t *d_line_a,
                           int8_t *d_line_b, int16_t *d_pcm);

/* The cell-exact iteration with the dB ROW materialised as float32.  Everything is computed in float64 from the IQ to the display cells, as
 * in pss_frame_pipeline_f64 (compute_fft signal_processing.py:243-264, the caller's smoothing + median clamp pyspecsdr.py:2278-2283, the
 * accumulators :1342-1406 / :1512-1564), but what is written per bin is compute_fft's float64 value ROUNDED ONCE to float32 — d_db32
 * [n_frames][n], the spectrum output's own contract (1e-4 relative; the rounding is ~6e-8) and SURVEY's 4 bytes per bin — and, only if
 * d_db64 != NULL, the float64 value as well.  1024-point frames: the transform and the post-process of a frame are ONE kernel
 * (k_spectrum_post: the frame's float64 dB values go from the transform's registers through LDS into the select; option "fuse_post" = 0:
 * two kernels) — the float64 rows never travel through HBM unless asked for.  Other lengths: the float64 rows go through d_db64 or a
 * context-owned scratch, and a conversion pass writes d_db32.  The float64 entry points above use the fused kernel too (their d_db = d_db64).
 *   pss_frame_pipeline_cells: pss_frame_pipeline_f64's arguments with (d_db32, d_db64) in place of d_db and no d_post; this is the step bench.py times
 *   pss_spectrum_cells:       its display half alone — compute_fft -> post-process -> display line of every frame, no demodulator
 *                             (d_db32 or d_db64 may be NULL, not both) */
int pss_frame_pipeline_cells(pss_ctx *ctx, int mode, const float *d_iq, long n_frames, int n, double fs, float *d_db32, double *d_db64,
                             double *d_row_lo, double *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w,
                             int8_t *d_line_a, int8_t *d_line_b, int16_t *d_pcm);
int pss_spectrum_cells(pss_ctx *ctx, const float *d_iq, long n_frames, int n, float *d_db32, double *d_db64, double *d_row_lo,
                       double *d_row_hi, int n_halo, int window, int display, int disp_h, int disp_w, int8_t *d_line_a, int8_t *d_line_b);

/* complex128 read buffers.  The reference's SDR buffer is complex64 (pyspecsdr.py:1887), but its functions accept any array, and handed
 * complex128 they compute in float64 from the first statement on.  These entry points serve compute_fft (signal_processing.py:243-264: the
 * window product is float64 x float64) and demodulate_am (:179-195: np.abs / np.mean / the subtraction in float64 — the same scaled hypot and
 * pairwise tree as the complex64 loops, one type wider) for such buffers: float64 dB rows to ~1e-10 of NumPy's, AM audio and int16 PCM bit
 * for bit (tests/golden/c128.npz).  Plain kernels — a conformance path for callers that hold float64 samples, not a throughput path.
 * The other demodulators narrow complex128 input to complex64 (np.angle on complex128 is libm's float64 atan2: not restated).
 *   d_iq / h_iq: interleaved float64 (re, im); n_fft a power of two in [16, 65536] (PSS_E_ARG otherwise); demodulate_am: any n >= 1
 *   pss_demod_am_c128: d_pcm int16 [n_frames][n][2] and / or d_audio float64 [n_frames][n] (mono); pss_h_*: one buffer, host pointers,
 *   h_audio_stereo float64 [n][2] */
int pss_spectrum_db_c128(pss_ctx *ctx, const double *d_iq, long n_frames, int n_fft, double *d_db);
int pss_demod_am_c128(pss_ctx *ctx, const double *d_iq, long n_frames, int n, int16_t *d_pcm, double *d_audio);
int pss_h_compute_fft_c128(pss_ctx *ctx, const double *h_iq, int n, double *h_db);
int pss_h_demodulate_am_c128(pss_ctx *ctx, const double *h_iq, int n, double *h_audio_stereo, int16_t *h_pcm);
/* demodulate_ssb (signal_processing.py:198-217) of complex128 frames: lfilter's complex128 convolution on the samples as they are (a complex64 buffer is
 * widened first by the reference: the same kernels behind a float64 loader), hilbert() round trip, normalisation, int16 (tests/golden/c128.npz keys
 * ssb_*: int16 equal; float64 audio bit for bit with option "hilbert_exact" = 1 at power-of-two lengths, to ~1e-15 otherwise — as for complex64). */
int pss_demod_ssb_c128(pss_ctx *ctx, int lower, const double *d_iq, long n_frames, int n, double fs, int16_t *d_pcm, double *d_audio);
int pss_h_demodulate_ssb_c128(pss_ctx *ctx, int lower, const double *h_iq, int n, double fs, double *h_audio_stereo, int16_t *h_pcm);
/* measure_signal_power (signal_processing.py:325-328) of complex128 frames, the ARRAY part: d_power[f] = np.mean(np.abs(x) ** 2) in float64 (np.abs's
 * scaled hypot, x * x, NumPy's pairwise sum, / n — tests/golden/c128.npz keys mp_*).  The scalar the reference finishes with,
 * 10 * log10(power + 1e-10), is left to the caller: it is NumPy's float64 log10 of ONE number (the Python shim applies NumPy's own, which gives the
 * reference's bits on the host it runs on: keys pw_*). */
int pss_mean_power_c128(pss_ctx *ctx, const double *d_iq, long n_frames, int n, double *d_power);
int pss_h_mean_power_c128(pss_ctx *ctx, const double *h_iq, int n, double *h_power);

/* Waterfall / persistence quantisers over a ring of post-processed rows (pyspecsdr.py:1342-1406,
 * :1512-1564).  d_rows float32 [n_rows][len], oldest first (n_rows <= 30 / <= 10).
 * d_glyph/d_colour int8 [disp_h][disp_w] (-1 = not drawn; persistence: 0 = empty). */
int pss_waterfall_cells(pss_ctx *ctx, const float *d_rows, int n_rows, int len, int disp_h, int disp_w,
                        int8_t *d_glyph, int8_t *d_colour);
int pss_persistence_cells(pss_ctx *ctx, const float *d_rows, int n_rows, int len, int disp_h, int disp_w,
                          int8_t *d_colour);
/* Spectrum display quantiser — draw_spectrogram (pyspecsdr.py:398-498), one independent post-processed dB row per display:
 * 20th-percentile noise floor, display range, clip + x**0.7, resample to disp_w columns, bar height and the glyph / colour
 * of every cell.  d_rows [n_rows][len]; d_glyph / d_colour int8 [n_rows][disp_h][disp_w]: glyph 0 '.', 1 '-', 2 '=',
 * 3 '#', 4 ' '; colour = curses pair number (1 = cleared cell); -1 where the column was not drawn (non-finite value).
 * d_range (nullable) double [n_rows][2] = (display_min, display_max), the dB range of the scale labels (:424-436). */
int pss_spectrogram_cells(pss_ctx *ctx, const float *d_rows, long n_rows, int len, int disp_h, int disp_w, int8_t *d_glyph,
                          int8_t *d_colour, double *d_range);
int pss_spectrogram_cells_f64(pss_ctx *ctx, const double *d_rows, long n_rows, int len, int disp_h, int disp_w,
                              int8_t *d_glyph, int8_t *d_colour, double *d_range);

/* Gradient waterfall — draw_gradient_waterfall (pyspecsdr.py:1640-1716) over the same ring of rows: glyph = index into
 * ' ._-=+*#@' (0..8), colour index 0..5, -1 = not drawn.  d_glyph / d_colour int8 [disp_h][disp_w], disp_w = max_width - 10. */
int pss_gradient_cells(pss_ctx *ctx, const float *d_rows, int n_rows, int len, int disp_h, int disp_w, int8_t *d_glyph,
                       int8_t *d_colour);
int pss_gradient_cells_f64(pss_ctx *ctx, const double *d_rows, int n_rows, int len, int disp_h, int disp_w, int8_t *d_glyph,
                           int8_t *d_colour);
/* Surface plot — draw_surface_plot (pyspecsdr.py:1567-1616) of one row on the whole screen: d_colour int8 [max_h][max_w],
 * 0 = empty, else the curses pair 1..5 of the '#' drawn there. */
int pss_surface_cells(pss_ctx *ctx, const float *d_row, int len, int max_h, int max_w, int8_t *d_colour);
int pss_surface_cells_f64(pss_ctx *ctx, const double *d_row, int len, int max_h, int max_w, int8_t *d_colour);

/* Constellation display — draw_vector_display (pyspecsdr.py:1718-1752): d_grid int8 [max_h][max_w], 1 where the '.' of
 * one of the n IQ samples lands (float32 arithmetic as in the reference). */
int pss_vector_cells(pss_ctx *ctx, const float *d_iq, int n, int max_h, int max_w, int8_t *d_grid);

/* decode_morse, front half (decoders.py:149-165): envelope = |x| / max|x| in float32, 20 log10(envelope + 1e-10) > threshold,
 * and the indices of the rising / falling transitions of that mask (np.diff + np.where), i.e. the arrays rise_times /
 * fall_times the timing logic of decode_morse (:167 ff., stays in the reference's own decoders.py) starts from.
 * d_iq: interleaved complex64 [n_frames][n]; d_rise / d_fall: int32 [n_frames][cap], in increasing order; d_counts: int32
 * [n_frames][2] = the TRUE numbers of rises and falls (entries beyond cap are dropped).  threshold_db: any value — the mask is
 * float32(20 log10(envelope + 1e-10)) > float32(threshold_db) with NumPy's float32 log10 bit for bit (the reference calls it with -20,
 * pyspecsdr.py:573: that case is one comparison against a precomputed float32 cut). */
int pss_morse_edges(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double threshold_db, int cap, int32_t *d_rise,
                    int32_t *d_fall, int32_t *d_counts);
int pss_h_morse_edges(pss_ctx *ctx, const float *h_iq, int n, double threshold_db, int cap, int32_t *h_rise, int32_t *h_fall,
                      int *n_rise, int *n_fall);

/* The decoders' per-message halves (host side; no GPU work, no context): what decode_morse does with its rise / fall indices
 * (decoders.py:167-231) and what decode_aprs does with its bit stream (decode_ax25_frame + decode_aprs_payload, decoders.py:6-88).
 *   pss_h_morse_decode: rise / fall = the int32 sample indices pss_morse_edges / pss_h_morse_edges return.  Pulse lengths and gaps in
 *       seconds, two pulse classes (the reference: scipy.cluster.vq.kmeans(durations, 2) from random starting points; here the partition
 *       Lloyd's iteration leaves unchanged with the smallest mean distance, class means summed in observation order — the reference's
 *       centroids wherever its answer does not depend on its random draw, i.e. for every keyed signal), symbols, letter gaps (> 3 dots),
 *       word gaps (> 7 dots), ITU table ('?' for an unknown symbol).  text: NUL-terminated ASCII, returns its length (0: no pulse);
 *       timing3 = (dot, dash, mean gap) in seconds.  PSS_E_ARG: bad arguments, text_cap too small, edges that do not alternate.
 *   pss_h_ax25_frame: bits = one 0/1 value per byte.  First flag 01111110, bits up to the next flag with stuffed zeros dropped, bytes LSB
 *       first, "SOURCE>DEST:info" (addresses: 7-bit characters shifted down by one, stripped of white space).  Returns 1 and the packet
 *       in out (*out_len bytes = the characters' code points 0..255, not NUL-terminated: info may hold NULs), 0 when there is no frame of
 *       at least 14 bytes (the reference returns None), PSS_E_ARG on bad arguments or out_cap too small.
 *   pss_h_decode_morse / pss_h_decode_aprs: the whole decoders on one host buffer — edges / normalisation + AFSK bit slicer on the GPU, the
 *       functions above behind them.  h_audio is real float64 (np.real of a complex buffer is the caller's, decoders.py:122-123). */
int pss_h_morse_decode(const int32_t *rise, long n_rise, const int32_t *fall, long n_fall, double fs, char *text, long text_cap,
                       double *timing3);
int pss_h_ax25_frame(const uint8_t *bits, long n_bits, char *out, long out_cap, long *out_len);
int pss_h_decode_morse(pss_ctx *ctx, const float *h_iq, int n, double fs, double threshold_db, char *text, long text_cap, double *timing3);
int pss_h_decode_aprs(pss_ctx *ctx, const double *h_audio, int n, double fs, const double *sos1200, const double *sos2200, int nsec,
                      char *out, long out_cap, long *out_len);

/* classify_signal (signal_processing.py:296-322; helpers estimate_bandwidth :267-280, estimate_modulation_index :283-293) for
 * a batch of scanner reads, as the function runs once its missing `welch` import is supplied (in the reference it raises
 * NameError on every call: SURVEY App. C2, §8(f) #3).  d_iq: interleaved complex64 [n_frames][n], n >= 1.  n >= 1024: Welch
 * segments of 1024 samples every 512; shorter reads: ONE segment of n samples (SciPy's nperseg = n fallback: Hann window and
 * transform of length n, n PSD bins — the first n entries of a d_psd row).  n < 1: PSS_E_ARG (welch raises).  Outputs (each may be
 * NULL): d_label int32 [n_frames] (PSS_CLASS_*), d_bw float64 (the "bandwidth" of estimate_bandwidth: last minus first bin
 * above max - 20 dB in FFT order, in Hz), d_mi float32 (modulation index, bit-exact with NumPy's float32 evaluation),
 * d_flat float32 (spectral flatness), d_psd float32 [n_frames][1024] (Welch PSD, FFT order).  The segment FFT runs in
 * float64 (the reference's is single precision): PSD and flatness agree to the reference's own float32 FFT noise. */
enum { PSS_CLASS_UNKNOWN = 0, PSS_CLASS_FM_BROADCAST = 1, PSS_CLASS_NARROW_FM = 2, PSS_CLASS_AM_BROADCAST = 3, PSS_CLASS_SSB = 4,
       PSS_CLASS_DIGITAL = 5 };
int pss_classify(pss_ctx *ctx, const float *d_iq, long n_frames, int n, double fs, int32_t *d_label, double *d_bw, float *d_mi,
                 float *d_flat, float *d_psd);
const char *pss_class_name(int label);
/* One read buffer in host memory (what pyspecsdr.py:1061 / :2560 hand over), synchronous. */
int pss_h_classify_signal(pss_ctx *ctx, const float *h_iq, int n, double fs, int *label, double *bw, float *mi, float *flat);

/* Same quantisers over float64 rows (the reference's rows are float64; used to check cell-exact parity). */
int pss_waterfall_cells_f64(pss_ctx *ctx, const double *d_rows, int n_rows, int len, int disp_h, int disp_w,
                            int8_t *d_glyph, int8_t *d_colour);
int pss_persistence_cells_f64(pss_ctx *ctx, const double *d_rows, int n_rows, int len, int disp_h, int disp_w,
                              int8_t *d_colour);

/* Stateful accumulators: a device ring of the last max_rows post-processed rows, the counterpart of the reference's
 * WATERFALL_HISTORY (30 rows, pyspecsdr.py:130-131,1351-1353) and PERSISTENCE_HISTORY (10 rows, :151-152,1521-1523).
 * push = history.append(row) + pop(0) when full; the two quantisers then work on the ring in place. */
typedef struct pss_ring pss_ring;
int pss_ring_create(pss_ctx *ctx, int max_rows, int len, pss_ring **out);
void pss_ring_destroy(pss_ring *ring);
int pss_ring_push(pss_ring *ring, const float *d_row);
int pss_ring_count(pss_ring *ring);
int pss_ring_waterfall(pss_ring *ring, int disp_h, int disp_w, int8_t *d_glyph, int8_t *d_colour);
int pss_ring_persistence(pss_ring *ring, int disp_h, int disp_w, int8_t *d_colour);

/* Batched accumulators: the display line the reference would draw for EVERY frame of a batch, in one call.
 * draw_waterfall / draw_persistence (pyspecsdr.py:1342-1406, :1512-1564) append the frame's post-processed row to a history
 * of `window` rows (30 / 10) and normalise with the finite extremes of that history; for frame i of a batch that is the
 * sliding-window minimum / maximum of the per-row extremes over rows i-window+1 .. i.
 *   d_post            [n_frames][len]        the batch's post-processed rows
 *   d_row_lo/d_row_hi [n_halo + n_frames]    per-row finite extremes (pss_spectrum_post_extremes / pss_row_extremes); the
 *                                            first n_halo entries belong to the rows that PRECEDE the batch (the previous
 *                                            batch's tail, or the left neighbour's when frames are sharded over GPUs:
 *                                            8 bytes per row cross the link instead of the rows) — 0 for a fresh history
 *   waterfall:   d_glyph / d_colour int8 [n_frames][disp_w] = line y = 0 (the newest row) of the reference's grid at frame i
 *                (glyph 0 '.', 1 '-', 2 '=', 3 '#'; colour 0..5; -1 where the value is not finite)
 *   persistence: d_y int8 [n_frames][disp_w] = row index int((1 - norm) * (disp_h - 1)) of the newest trace's '*' in every
 *                column (-1: not drawn); disp_h <= 127. */
int pss_waterfall_rows(pss_ctx *ctx, const float *d_post, long n_frames, int len, const float *d_row_lo, const float *d_row_hi,
                       int n_halo, int window, int disp_w, int8_t *d_glyph, int8_t *d_colour);
int pss_waterfall_rows_f64(pss_ctx *ctx, const double *d_post, long n_frames, int len, const double *d_row_lo,
                           const double *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph, int8_t *d_colour);
int pss_persistence_rows(pss_ctx *ctx, const float *d_post, long n_frames, int len, const float *d_row_lo, const float *d_row_hi,
                         int n_halo, int window, int disp_h, int disp_w, int8_t *d_y);
int pss_persistence_rows_f64(pss_ctx *ctx, const double *d_post, long n_frames, int len, const double *d_row_lo,
                             const double *d_row_hi, int n_halo, int window, int disp_h, int disp_w, int8_t *d_y);
/* The same WITHOUT materialised post-processed rows (float32 path; n_fft a multiple of 4, n_fft - 4 <= 32768).
 * pss_spectrum_post_thresholds: the post-process of pyspecsdr.py:2278-2283 reduced to what the accumulators need of every row —
 * d_row_thr float32 [n_frames] (the clamp threshold float32(median - 10)) and the finite extremes of the clamped row.
 * pss_waterfall_rows_db / pss_persistence_rows_db: the display lines from the dB rows [n_frames][n_fft] and those thresholds; element j
 * of a post-processed row is rebuilt where a cell needs it — float32(sum of 5 dB values * 0.2 in np.convolve's order) clamped at the
 * threshold, bit for bit what pss_spectrum_post writes — so the lines are byte-identical to pss_waterfall_rows on materialised rows. */
int pss_spectrum_post_thresholds(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, float *d_row_thr, float *d_row_lo,
                                 float *d_row_hi);
int pss_waterfall_rows_db(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, const float *d_row_thr, const float *d_row_lo,
                          const float *d_row_hi, int n_halo, int window, int disp_w, int8_t *d_glyph, int8_t *d_colour);
int pss_persistence_rows_db(pss_ctx *ctx, const float *d_db, long n_frames, int n_fft, const float *d_row_thr, const float *d_row_lo,
                            const float *d_row_hi, int n_halo, int window, int disp_h, int disp_w, int8_t *d_y);

/* ---- host-buffer convenience (single frame, synchronous; what the drop-in Python module calls) --- */
int pss_h_compute_fft(pss_ctx *ctx, const float *h_iq, int n, double *h_db);
int pss_h_demodulate(pss_ctx *ctx, int mode, const float *h_iq, int n, double fs, double *h_audio_stereo,
                     int16_t *h_pcm);
/* scipy.signal.sosfilt(sos, x) with zero initial state on n_rows independent float64 rows of n samples (one lane per row);
 * sos: HOST pointer, nsec <= 8 rows of 6.  The building block behind bandpass_filter (signal_processing.py:34-42). */
int pss_sosfilt(pss_ctx *ctx, const double *d_x, long n_rows, int n, const double *sos, int nsec, double *d_y);
/* decode_afsk (decoders.py:94-112) for n_rows independent float64 audio rows of n samples: the two Bell-202 band-passes,
 * the energy of each band per bit period of int(fs/1200) samples, bit = e2200 > e1200.  d_bits uint8 [n_rows][n_bits],
 * n_bits = pss_afsk_n_bits(n, fs) = len(range(0, n - window, window)).  sos1200 / sos2200: HOST tables of nsec rows
 * (butter(5, [1100,1300] / [2100,2300] Hz, 'band')), or NULL to design them. */
int pss_afsk_n_bits(int n, double fs);
int pss_afsk_bits(pss_ctx *ctx, const double *d_audio, long n_rows, int n, double fs, const double *sos1200,
                  const double *sos2200, int nsec, uint8_t *d_bits);
/* The float32 primitives the path is built from, element by element, as NumPy's AVX512_SKX loops evaluate them (the
 * reference gets them from np.angle -> np.arctan2 :97/:125, np.log10 :328/:270, np.abs :185/:286): PSS_NP_ARCTAN2
 * out = arctan2(a, b) (SVML atan2f16 model), PSS_NP_LOG10 out = log10(a) (SVML log10f16 model; d_b unused),
 * PSS_NP_ABS out = abs(a + i b) (npy_hypotf).  Exposed so the models can be checked bit for bit against NumPy's outputs. */
enum { PSS_NP_ARCTAN2 = 0, PSS_NP_LOG10 = 1, PSS_NP_ABS = 2 };
int pss_np_f32(pss_ctx *ctx, int op, const float *d_a, const float *d_b, long n, float *d_out);
/* samples / np.max(np.abs(samples)) on n_rows float64 rows (decode_aprs's normalisation, decoders.py:126). */
int pss_row_normalise(pss_ctx *ctx, const double *d_x, long n_rows, int n, double *d_y);
/* decode_afsk on one HOST buffer of real float64 audio; normalise != 0 divides by max|x| on the device first, which makes
 * the result the bit stream decode_aprs (decoders.py:115-133) hands to its AX.25 framing.  h_bits uint8 [pss_afsk_n_bits]. */
int pss_h_afsk_bits(pss_ctx *ctx, const double *h_audio, int n, double fs, int normalise, const double *sos1200,
                    const double *sos2200, int nsec, uint8_t *h_bits);
/* bandpass_filter(data, lowcut, highcut, sample_rate) (signal_processing.py:34-42; used by decoders.py:100-101) on one
 * host row: lowcut <= 0 -> butter(5) low-pass at highcut, else band-pass.  sos/nsec: caller's table, or NULL to design it. */
int pss_h_bandpass_filter(pss_ctx *ctx, const double *h_x, int n, double lowcut, double highcut, double fs,
                          const double *sos, int nsec, double *h_y);
/* demodulate_signal(samples, fs, mode) on one host frame (dispatcher semantics: WFM is IQ-corrected first). */
int pss_h_demodulate_signal(pss_ctx *ctx, int mode, const float *h_iq, int n, double fs, double *h_audio_stereo,
                            int16_t *h_pcm);
/* demodulate_signal over n_frames read buffers held in host memory (e.g. an IQ recording, pyspecsdr.py:814-824, cut into
 * the reference's read size): h_pcm int16 [n_frames][n_out][2], chunk_frames frames per round trip. */
int pss_h_demodulate_batch(pss_ctx *ctx, int mode, const float *h_iq, long n_frames, int n, double fs, long chunk_frames,
                           int16_t *h_pcm);
int pss_h_measure_power(pss_ctx *ctx, const float *h_iq, int n, float *h_power);
/* iq_correction(samples) / demodulate_signal(samples, fs, 'RAW') on one host frame; either output may be NULL. */
int pss_h_iq_correction(pss_ctx *ctx, const float *h_iq, int n, float *h_out_iq, float *h_raw);

/* ---- streamed capture (BASELINE.json configs[4]) -------------------------------------------------- */
/* Pinned host memory for the streaming call (plain malloc'ed memory works too, but cannot overlap with compute). */
void *pss_host_alloc(size_t bytes);
void pss_host_free(void *p);
/* Cut a long HOST capture into n_frames frames of n samples and push it through spectrum + NFM demod in chunks of
 * chunk_frames: chunk k+1 is uploaded (hipMemcpyAsync, copy stream) while chunk k computes and chunk k-1's results
 * download — three streams, two device buffer sets.  h_db float32 [n_frames][n] (may be NULL), h_pcm int16
 * [n_frames][n_out][2].  Synchronous: returns when everything has landed in host memory.  Results are identical to
 * the device-resident pss_spectrum_nfm on the same frames. */
int pss_h_stream_spectrum_nfm(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames,
                              float *h_db, int16_t *h_pcm);

/* The same capture, returning what the application consumes per frame instead of the dB rows: the display accumulator's
 * newest line (mode 0 waterfall: h_line_a = glyph, h_line_b = colour; mode 1 persistence: h_line_a = row index of the newest
 * trace, h_line_b unused) with a history of `window` rows, and the int16 PCM.  int8 [n_frames][disp_w] each.  Post-process
 * and accumulators run on the device chunk by chunk; the history crosses chunk boundaries.  h_halo_lo / h_halo_hi (n_halo
 * floats each, nullable): extremes of the rows preceding this capture (pss_waterfall_rows).  h_db (nullable) also returns
 * the dB rows; h_row_lo / h_row_hi (nullable, n_frames floats) return the per-row extremes (the halo for whatever follows).
 * Results are identical to the device-resident calls on the same frames. */
int pss_h_stream_display_nfm(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames, int mode,
                             int window, int disp_h, int disp_w, const float *h_halo_lo, const float *h_halo_hi, int n_halo,
                             int8_t *h_line_a, int8_t *h_line_b, int16_t *h_pcm, float *h_db, float *h_row_lo, float *h_row_hi);

/* The same capture (a fresh history: no halo), and additionally the FULL display grid as the reference's screen shows it after the LAST
 * frame of every chunk: draw_waterfall / draw_persistence redraw all `window` lines / traces of the history with the history's current extremes
 * on every frame (pyspecsdr.py:1342-1406, :1512-1564), so older lines cannot be rebuilt from the per-frame lines above — a display refreshed
 * once per chunk needs this grid.  h_grid_a (+ h_grid_b: the waterfall's colour plane) int8 [n_chunks][disp_h][disp_w], n_chunks =
 * ceil(n_frames / chunk_frames); cell values as pss_waterfall_cells / pss_persistence_cells (which produce it from the last `window`
 * post-processed rows, carried from chunk to chunk on the device).  For a resident batch with materialised post-processed rows the grid after
 * frame f is pss_waterfall_cells(d_post + (f - w + 1) * len, w, ...), w = min(window, f + 1). */
int pss_h_stream_display_nfm_grids(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames, int mode,
                                   int window, int disp_h, int disp_w, int8_t *h_line_a, int8_t *h_line_b, int16_t *h_pcm,
                                   float *h_row_lo, float *h_row_hi, int8_t *h_grid_a, int8_t *h_grid_b);
/* pss_h_stream_display_nfm with compute_fft's own float64 rows from the transform to the cells (n: a power of two in [16, 65536]): the lines —
 * and, with h_grid_a (+ h_grid_b for the waterfall; then no halo), the full screens after every chunk — are the cells the reference draws
 * from this capture.  h_halo_lo / _hi, h_db, h_row_lo / _hi: float64.  The capture is PCIe-bound either way (same time as the float32 call). */
int pss_h_stream_display_nfm_f64(pss_ctx *ctx, const float *h_iq, long n_frames, int n, double fs, long chunk_frames, int mode,
                                 int window, int disp_h, int disp_w, const double *h_halo_lo, const double *h_halo_hi, int n_halo,
                                 int8_t *h_line_a, int8_t *h_line_b, int16_t *h_pcm, double *h_db, double *h_row_lo, double *h_row_hi,
                                 int8_t *h_grid_a, int8_t *h_grid_b);

/* ---- Multi-GPU: the path's exchange steps (one process per GPU, RCCL over xGMI) ----------------------------------------------------------
 * The path shards by contiguous blocks of independent frames / scanner slices (the sweep of pyspecsdr.py:2514-2590; every read buffer of
 * the loop :2236-2283 is processed on its own): every rank runs the single-GPU entry points above on its block, with NO collective in
 * the data path.  What crosses ranks is the gather of a rank's results to one rank (or all), and — for the display accumulators — the row
 * extremes of the frames just before a rank's block.  A Python host does both over torch.distributed (pyspecsdr_amd/shard.py); these
 * entry points are the same two steps for a host without it, queued on the context's stream.  librccl.so.1 is opened at the first call
 * ($PSS_RCCL_LIB if set — an explicit choice wins —, else a copy the process already holds, the loader's path, /opt/rocm/lib) — the library does not link against it,
 * and a lone rank (n_ranks = 1, id = NULL) never touches it.  Errors: PSS_E_COMM. */
#define PSS_COMM_ID_BYTES 128
/* contiguous blocks whose sizes differ by at most one: items [*start, *start + *count) belong to `rank` */
int pss_shard_range(long n_items, int rank, int n_ranks, long *start, long *count);
/* rank 0: a rendezvous id (ncclGetUniqueId) to hand to every other rank by whatever the host has — a file, a socket, its launcher */
int pss_comm_id(void *id /* PSS_COMM_ID_BYTES */);
/* every rank, after pss_create on ITS device: join the communicator (ncclCommInitRank; blocks until all n_ranks have called) */
int pss_comm_init(pss_ctx *ctx, const void *id, int rank, int n_ranks);
int pss_comm_free(pss_ctx *ctx);                              /* also done by pss_destroy */
int pss_comm_size(pss_ctx *ctx, int *rank, int *n_ranks);     /* (0, 1) without a communicator */
/* ONE collective for everything a rank produced in a sharded pass: every rank contributes `bytes` bytes (its packed result buffer, sized
 * for the largest block); d_all on the receiving rank(s): n_ranks x bytes, rank r's buffer at r * bytes.  dst < 0: all-gather; dst >= 0:
 * gather to that rank as grouped point-to-point messages (each peer's own xGMI link to the root; d_all is ignored elsewhere). */
int pss_gather_packed(pss_ctx *ctx, const void *d_local, size_t bytes, void *d_all, int dst);
/* The rows preceding this rank's block, up to `halo` of them (waterfall: the (lo, hi) extremes of 30 post-processed rows, persistence: 10
 * — pyspecsdr.py:130-132, :151-154; what pss_waterfall_rows' d_halo_lo / _hi take).  d_rows: this rank's counts[rank] rows of row_bytes
 * each, in frame order; counts: every rank's block size (the same n_ranks values on every rank, e.g. from pss_shard_range); d_halo:
 * room for `halo` rows; *n_halo = rows received = min(halo, rows before this block), in global order. */
int pss_halo_from_left(pss_ctx *ctx, const void *d_rows, const long *counts, size_t row_bytes, long halo, void *d_halo, long *n_halo);

/* Kernel-only time of the most recent batched call on this context, measured with HIP events on the
 * context's stream (ms); negative if timing is disabled.  pss_enable_timing(ctx, 1) turns it on. */
int pss_enable_timing(pss_ctx *ctx, int on);
float pss_last_kernel_ms(pss_ctx *ctx);
/* Per-kernel durations, "name=ms;name=ms;...", one entry per kernel launch since timing was enabled or since the
 * previous pss_kernel_times() call (HIP events around each launch on the context's stream). */
int pss_kernel_times(pss_ctx *ctx, char *buf, int buf_len);
/* Restrict the per-kernel events to launches of ONE kernel (the name pss_kernel_times reports, e.g. "k_nfm_fwd") and
 * drop the per-call events: every event is a barrier packet in the queue (~4 us), which a throughput measurement over
 * many launches should not pay for kernels it is not reporting.  NULL or "" = every kernel (the default). */
int pss_timing_filter(pss_ctx *ctx, const char *kernel);

#ifdef __cplusplus
}
#endif
#endif
