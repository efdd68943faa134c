/*
 * pss_oracle.h — CPU restatement (plain C, scalar) of the PySpecSDR IQ -> spectrum + demod hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (pyspecsdr_amd/) never links or calls it.
 *
 * Every function cites the reference lines it follows (paths relative to the reference tree,
 * xqtr/PySpecSDR v1.0.6) and, where the arithmetic lives in an un-vendored third-party
 * dependency (requirements.txt:1-2 -> numpy>=1.20, scipy>=1.7; pinned here by the golden
 * fixtures to NumPy 2.2.6 / SciPy 1.15.3, AVX512_SKX dispatch), the library routine it restates.
 *
 * Parity pin: tests/test_oracle_golden.py checks every function below against
 * tests/golden/ (npz files), which tools/make_goldens.py produced by importing the reference itself.
 * (The reference ships no tests or golden vectors of its own — SURVEY.md §4.)
 */
#ifndef PSS_ORACLE_H
#define PSS_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- float32 primitives (NumPy AVX512_SKX ufunc loops) ---- */
float pss_o_rcp14f(float x);                 /* VRCP14PS, bit-exact model (64-entry piecewise-linear table) */
float pss_o_atan2f(float y, float x);        /* numpy.arctan2 float32 == Intel SVML __svml_atan2f16 (la)     */
float pss_o_cabsf(float re, float im);       /* numpy.abs(complex64)                                          */
float pss_o_pairwise_sum_f32(const float *a, long n); /* numpy add.reduce float32 (8192-element chunks, pairwise inside) */
float pss_o_log10f_np(float x);              /* np.log10 float32 (SVML __svml_log10f16 model, bit-pinned: tests/golden/log10f.npz) */
void pss_o_log10f_np_many(const float *x, float *y, long n);
void pss_o_atan2f_many(const float *y, const float *x, float *out, long n);
void pss_o_cabsf_many(const float *re, const float *im, float *out, long n);
float pss_o_log10f_ref(float x);             /* = pss_o_log10f_np (kept for callers) */

/* ---- per-frame functions; iq = interleaved complex64 (I0,Q0,I1,Q1,...) ---- */

/* compute_fft — signal_processing.py:243-264.  db[n] float64, DC-centred. */
void pss_o_compute_fft(const float *iq, int n, double *db);
void pss_o_compute_fft_c128(const double *iq, int n, double *db);     /* complex128 buffer: float64 window product */
/* caller post-process — pyspecsdr.py:2278-2283.  out[n-4]. */
void pss_o_postprocess(const double *db, int n, double *out);
/* measure_signal_power — signal_processing.py:325-328 (float32 throughout). */
float pss_o_power_db(const float *iq, int n);
/* inline scanner slice — pyspecsdr.py:2542-2552 (float32 dB). Returns number of bins in the 20 dB mask. */
int pss_o_scan_slice(const float *iq, int n, double fs, float *db, float *peak, double *bw);

/* iq_correction — signal_processing.py:46-80 (DC removal, IQ amplitude/phase balance, power restore), float32
 * throughout.  out[2n] interleaved complex64.  demodulate_signal(..., 'RAW') (:222-238) = real part of this. */
void pss_o_iq_correction(const float *iq, int n, float *out);

/* demodulate_nfm — signal_processing.py:91-116.
 * taps[65] = firwin(65, 15000/(fs/2)); sos[4][6] = cheby1(8,0.05,0.8/q,'sos'); zi[4][2] = sosfilt_zi(sos).
 * audio[n_out], n_out = ceil((n-1)/q).  Returns n_out, or -1 if n-1 <= 27 (sosfiltfilt padlen ValueError).
 * work: optional stage dumps (may be NULL): disc[n-1] float32, fir[n-1] float64. */
int pss_o_demod_nfm(const float *iq, int n, double fs, int q, const double *taps, const double *sos,
                    const double *zi, double *audio, float *disc_out, double *fir_out);
/* demodulate_am — signal_processing.py:179-195 (+ bandpass_filter :34-42). sos[5][6]. audio[n]. */
void pss_o_demod_am(const float *iq, int n, const double *sos, int nsec, double *audio);
double pss_o_mean_power_c128(const double *iq, int n, double *scratch);   /* complex128 buffer: np.mean(np.abs(x) ** 2) in float64 (the array part of :325-328) */
void pss_o_demod_am_c128(const double *iq, int n, const double *sos, int nsec, double *audio);   /* complex128 buffer: float64 abs / mean */
double pss_o_cabs(double re, double im);
double pss_o_pairwise_sum_f64(const double *a, long n);
/* demodulate_ssb — signal_processing.py:198-217. taps[65] = firwin(65, 3000/fs). audio[n].
 * hilbert(real(z)).real is restated as real(z) (identity up to 1e-15 round-off). */
void pss_o_demod_ssb(const float *iq, int n, const double *taps, double *audio);
void pss_o_demod_ssb_ex(const float *iq, int n, const double *taps, double *audio, int with_hilbert);
void pss_o_demod_ssb_c128(const double *iq, int n, const double *taps, double *audio, int with_hilbert);   /* complex128 buffer */
/* pss_pocketfft.c: scipy.signal.hilbert of a real float64 row of n = 2^k samples, bit for bit (out: n complex128 values, interleaved);
 * the plan's twiddle table exp(2 pi i k / n); the forward / inverse halves alone (scipy.fft.fft of a real row, scipy.fft.ifft). */
void pss_o_hilbert(const double *x, int n, double *out);
void pss_o_pocketfft_twiddles(int n, double *tw_re_im);
void pss_o_rfft_full(const double *x, int n, double *out);
void pss_o_cifft(const double *in, int n, double *out);
/* demodulate_wfm — signal_processing.py:119-176 (called on iq_correction()'s output, :222-228).
 * lp_sos[3][6] = butter(5, 15000/(fs/2), 'low'); pilot_sos[5][6] = butter(5, [18800, 19200]/(fs/2), 'band');
 * lmr_sos[5][6] = butter(5, [23000, 53000]/(fs/2), 'band'); alpha = exp(-1/(75e-6 fs)); dec_sos/dec_zi as for NFM;
 * q = int(fs/22050).  left/right[n_out].  Returns n_out, or -1 where the reference raises ValueError. */
int pss_o_demod_wfm(const float *iq, int n, int q, const double *lp_sos, const double *pilot_sos,
                    const double *lmr_sos, double alpha, const double *dec_sos, const double *dec_zi, double *left,
                    double *right);
/* bandpass_filter — signal_processing.py:34-42 with the SOS table given (butter(5) low-/band-pass): sosfilt, zero state. */
void pss_o_sosfilt(const double *sos, int nsec, const double *x, long n, double *y);
/* decode_afsk — decoders.py:94-112 given the two band-pass SOS tables: bits[k] = energy(2200 Hz band) > energy(1200 Hz band)
 * over bit period k of int(fs/1200) samples.  Returns the number of bits. */
int pss_o_afsk_bits(const double *x, int n, double fs, const double *sos1200, const double *sos2200, int nsec, uint8_t *bits);
/* int16 conversion — io_manager.py:25-26 / audio_processing.py:37: np.int16(x*32767), stereo dup
 * (mono_to_stereo signal_processing.py:83-88). pcm[2*n] = L0 R0 L1 R1 ... */
void pss_o_pcm16_stereo(const double *audio, int n, int16_t *pcm);

/* adjust_gain — pyspecsdr.py:898-919 (index part). */
int pss_o_agc_step(float power_db, int idx, int n_gains);

/* waterfall quantiser — pyspecsdr.py:1342-1406.  rows: ring contents oldest..newest, n_rows x len.
 * glyph/colour: [disp_h][disp_w] int8, -1 = not drawn.  Row y=0 is the newest. */
void pss_o_waterfall_cells(const double *rows, int n_rows, int len, int disp_h, int disp_w,
                           int8_t *glyph, int8_t *colour);
/* persistence quantiser — pyspecsdr.py:1512-1564. colour[disp_h][disp_w], 0 = empty. */
void pss_o_persistence_cells(const double *rows, int n_rows, int len, int disp_h, int disp_w,
                             int8_t *colour);

/* gradient waterfall — pyspecsdr.py:1640-1716: glyph = index into ' ._-=+*#@' (0..8), colour 0..5, -1 = not drawn. */
void pss_o_gradient_cells(const double *rows, int n_rows, int len, int disp_h, int disp_w, int8_t *glyph, int8_t *colour);
/* surface plot — pyspecsdr.py:1567-1616: colour[max_h][max_w] over the whole screen, 0 = empty, else pair 1..5 ('#'). */
void pss_o_surface_cells(const double *row, int len, int max_h, int max_w, int8_t *colour);
/* constellation display — pyspecsdr.py:1718-1752: grid[max_h][max_w] = 1 where a sample's dot lands. */
/* classify_signal (signal_processing.py:296-322 with welch bound to scipy.signal.welch); labels as in the reference */
enum { PSS_O_CLS_UNKNOWN = 0, PSS_O_CLS_FM_BROADCAST = 1, PSS_O_CLS_NARROW_FM = 2, PSS_O_CLS_AM_BROADCAST = 3, PSS_O_CLS_SSB = 4,
       PSS_O_CLS_DIGITAL = 5 };
float pss_o_modulation_index(const float *iq, long n);
void pss_o_hann1024_f32(float *w);
void pss_o_hann_f32(float *w, int n);
int pss_o_classify(const float *iq, long n, double fs, double *bw_out, float *mi_out, float *flat_out, float *psd_out);
/* decode_morse front half (decoders.py:149-165, threshold -20 dB): indices of the rising / falling transitions */
void pss_o_morse_edges(const float *iq, long n, int32_t *rise, int32_t *fall, long cap, long *n_rise, long *n_fall);
void pss_o_morse_edges_thr(const float *iq, long n, double threshold, int32_t *rise, int32_t *fall, long cap, long *n_rise, long *n_fall);
int pss_o_scan_threshold(const float *iq, int n, double fs, double threshold_db, float *db, float *peak, double *bw);
void pss_o_vector_cells(const float *iq, int n, int max_h, int max_w, int8_t *grid);
/* spectrum display quantiser — draw_spectrogram, pyspecsdr.py:398-498.  row[len] = one post-processed dB row.
 * glyph/colour [disp_h][disp_w]: glyph 0 '.', 1 '-', 2 '=', 3 '#', 4 ' '; colour = curses pair (1 = cleared); -1 = not drawn.
 * disp_min/disp_max (nullable): the dB range of the scale labels (:424-427). */
void pss_o_spectrogram_cells(const double *row, int len, int disp_h, int disp_w, int8_t *glyph, int8_t *colour,
                             double *disp_min, double *disp_max);

/* ---- batched drivers used only by bench.py's cpu_baseline leg (OpenMP over frames if enabled) ---- */
/* spectrum (float32 dB out) + NFM -> int16 stereo PCM, the BASELINE.json headline path. */
void pss_o_batch_spectrum_nfm(const float *iq, long n_frames, int n, double fs, int q, const double *taps,
                              const double *sos, const double *zi, float *db_out, int16_t *pcm_out,
                              int n_threads);
/* the same + the caller's post-process (float32 rows, per-row finite extremes); post_out / lo_out / hi_out nullable. */
void pss_o_batch_spectrum_post_nfm(const float *iq, long n_frames, int n, double fs, int q, const double *taps,
                                   const double *sos, const double *zi, float *db_out, float *post_out, float *lo_out,
                                   float *hi_out, int16_t *pcm_out, int n_threads);
/* the same step in the reference's own row type, float64 from IQ to cells (compute_fft returns float64; pyspecsdr.py:2278-2283 and
 * draw_waterfall :1342-1406 work on those rows): dB / post-processed rows (nullable), finite extremes per row, NFM PCM (nullable); and the
 * waterfall line of every frame from float64 rows and their extremes */
void pss_o_batch_headline_f64(const float *iq, long n_frames, int n, double fs, int q, const double *taps, const double *sos,
                              const double *zi, double *db_out, double *post_out, double *lo_out, double *hi_out, int16_t *pcm_out,
                              int n_threads);
void pss_o_waterfall_rows_f64(const double *rows, const double *row_lo, const double *row_hi, long n_frames, int len, int window, int disp_w,
                              int8_t *glyph, int8_t *colour, int n_threads);
void pss_o_persistence_rows_f64(const double *rows, const double *row_lo, const double *row_hi, long n_frames, int len, int window, int disp_h,
                                int disp_w, int8_t *ycell, int n_threads);
/* batched waterfall accumulator: newest display line per frame, history of `window` rows (pyspecsdr.py:1342-1406). */
void pss_o_waterfall_rows(const float *rows, long n_frames, int len, int window, int disp_w, int8_t *glyph, int8_t *colour,
                          int n_threads);
void pss_o_persistence_rows(const float *rows, long n_frames, int len, int window, int disp_h, int disp_w, int8_t *ycell, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
