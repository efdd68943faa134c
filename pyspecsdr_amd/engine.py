"""Engine: one libpss context (one GPU, one stream) with the batched device-pointer entry points.

Device buffers are passed as raw addresses: anything with .data_ptr() (torch tensors), an int, or None.
PyTorch is only plumbing here (device memory / streams / torch.distributed); the library itself is
plain HIP behind a C ABI.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import _lib as L


class PssError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpss error {code}: {msg}")
        self.code = code


def _ptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return int(x)


class Engine:
    """order: how calls are ordered against the caller's own GPU work.
         "torch" (default when torch is loaded): the library keeps its own non-blocking stream — which the default stream does NOT
                 synchronise with — and every device-pointer entry point is bracketed with pss_order_after / pss_order_before against
                 torch.cuda.current_stream(): buffers the caller filled on its stream are ready before the kernels read them, and torch work
                 queued after the call sees the results.  No host synchronisation; ~5 us per call.
         "none":  no ordering — for pipelines that order by hand (bench.py, multi.py: events / fences around many calls); the caller must
                 torch.cuda.synchronize() (or record / wait events on stream_handle()) between its own stream's work and the calls.
       stream: run on this hipStream_t / torch stream instead of the library's own (ordering is then the stream's own)."""

    def __init__(self, device=0, stream=None, order=None):
        self.lib = L.load()
        h = C.c_void_p()
        r = self.lib.pss_create(int(device), C.byref(h))
        if r != 0:
            raise PssError(r, self.lib.pss_last_error(None).decode())
        self.h = h
        self.device = device
        if stream is not None:
            self.set_stream(stream)
        if order is None:
            order = "none" if stream is not None else "torch"
        assert order in ("torch", "none")
        self.order = order
        # PSS_OPTIONS="key=value,key=value": pss_set_option switches for A/B measurements without touching the caller
        for kv in filter(None, os.environ.get("PSS_OPTIONS", "").split(",")):
            k, _, v = kv.partition("=")
            self.set_option(k.strip(), int(v))

    def _caller_stream(self):
        """torch's current stream on this engine's device as a hipStream_t (int), or None when there is nothing to order against."""
        if self.order != "torch":
            return None
        torch = sys.modules.get("torch")
        if torch is None or not torch.cuda.is_available() or not torch.cuda.is_initialized():
            return None
        return torch.cuda.current_stream(self.device).cuda_stream

    def _dev(self, fn, *args):
        """A device-pointer entry point, ordered after the caller's stream and the caller's stream after it (order = "torch")."""
        cs = self._caller_stream()
        if cs is not None:
            self._ck(self.lib.pss_order_after(self.h, cs))
        r = fn(self.h, *args)
        rb = self.lib.pss_order_before(self.h, cs) if cs is not None else 0
        self._ck(r)        # the call's own failure first,
        self._ck(rb)       # then a failed event record / stream wait: torch's work would not be ordered after the results

    def order_after(self, stream):
        """Work queued on this engine from now on starts after everything queued on `stream` (a hipStream_t as int, or an object with
        .cuda_stream) so far — e.g. another Engine's stream_handle()."""
        self._ck(self.lib.pss_order_after(self.h, _ptr(getattr(stream, "cuda_stream", stream))))

    def order_before(self, stream):
        """Work queued on `stream` from now on starts after everything this engine has queued so far."""
        self._ck(self.lib.pss_order_before(self.h, _ptr(getattr(stream, "cuda_stream", stream))))

    def close(self):
        if getattr(self, "h", None):
            self.lib.pss_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, r):
        if r != 0:
            msg = self.lib.pss_last_error(self.h).decode()
            if r in (L.PSS_E_PADLEN, L.PSS_E_CUTOFF):
                raise ValueError(msg)  # the exception type the reference's SciPy calls raise
            raise PssError(r, msg)

    # -- plumbing
    def set_stream(self, stream):
        self._ck(self.lib.pss_set_stream(self.h, _ptr(getattr(stream, "cuda_stream", stream))))

    # ---- the exchange steps behind the C ABI (RCCL; include/pss.h "Multi-GPU") — what shard.py does over torch.distributed, for hosts without it
    def comm_id(self):
        """Rank 0: the 128-byte rendezvous id to hand to every other rank."""
        buf = C.create_string_buffer(L.COMM_ID_BYTES)
        r = self.lib.pss_comm_id(buf)
        if r != 0:
            raise PssError(r, "pss_comm_id: librccl.so.1 not available (set PSS_RCCL_LIB)")
        return buf.raw

    def comm_init(self, comm_id, rank, n_ranks):
        """Join the communicator (blocks until all n_ranks have called).  comm_id None with n_ranks 1: a lone rank, no RCCL."""
        if comm_id is not None and len(comm_id) != L.COMM_ID_BYTES:
            raise ValueError("comm_id: the 128 bytes of Engine.comm_id()")
        self._ck(self.lib.pss_comm_init(self.h, comm_id, int(rank), int(n_ranks)))

    def comm_free(self):
        self._ck(self.lib.pss_comm_free(self.h))

    def comm_size(self):
        rank, n = C.c_int(), C.c_int()
        self._ck(self.lib.pss_comm_size(self.h, C.byref(rank), C.byref(n)))
        return rank.value, n.value

    def gather_packed(self, d_local, nbytes, d_all, dst=0):
        """Every rank's `nbytes` bytes at d_local -> d_all (n_ranks x nbytes, rank order) on rank dst, or on all ranks with dst None."""
        self._dev(self.lib.pss_gather_packed, _ptr(d_local), int(nbytes), _ptr(d_all), -1 if dst is None else int(dst))

    def halo_from_left(self, d_rows, counts, row_bytes, halo, d_halo):
        """The rows that precede this rank's block (at most `halo`), from its left neighbours; returns how many arrived."""
        arr = (C.c_long * len(counts))(*[int(c) for c in counts])
        got = C.c_long()
        self._dev(self.lib.pss_halo_from_left, _ptr(d_rows), arr, int(row_bytes), int(halo), _ptr(d_halo), C.byref(got))
        return got.value

    def stream_handle(self):
        """The hipStream_t this engine queues its work on, as an integer (e.g. for torch.cuda.ExternalStream)."""
        return int(self.lib.pss_get_stream(self.h) or 0)

    def set_option(self, key, value):
        self._ck(self.lib.pss_set_option(self.h, key.encode(), int(value)))

    def sync(self):
        self._ck(self.lib.pss_sync(self.h))

    def enable_timing(self, on=True):
        self._ck(self.lib.pss_enable_timing(self.h, int(on)))

    def timing_filter(self, kernel=None):
        """Only launches of `kernel` are bracketed with events (None = every kernel)."""
        self._ck(self.lib.pss_timing_filter(self.h, kernel.encode() if kernel else None))

    def last_kernel_ms(self):
        return float(self.lib.pss_last_kernel_ms(self.h))

    def kernel_times(self):
        """{kernel name: [ms per launch, ...]} for every launch since timing was enabled / last read
        (HIP events on the engine's stream)."""
        buf = C.create_string_buffer(1 << 20)
        self._ck(self.lib.pss_kernel_times(self.h, buf, 1 << 20))
        out = {}
        for item in buf.value.decode().split(";"):
            if item:
                k, v = item.split("=")
                out.setdefault(k, []).append(float(v))
        return out

    # -- filters
    def nfm_filters(self, fs):
        taps, sos, zi = np.empty(65), np.empty((4, 6)), np.empty((4, 2))
        self._ck(self.lib.pss_get_nfm_filters(self.h, float(fs), _ptr(taps), _ptr(sos), _ptr(zi)))
        return taps, sos, zi

    def set_nfm_filters(self, fs, taps, sos, zi):
        taps = np.ascontiguousarray(taps, np.float64)
        sos = np.ascontiguousarray(sos, np.float64)
        zi = np.ascontiguousarray(zi, np.float64)
        assert taps.shape == (65,) and sos.shape == (4, 6) and zi.shape == (4, 2)
        self._ck(self.lib.pss_set_nfm_filters(self.h, float(fs), _ptr(taps), _ptr(sos), _ptr(zi)))

    def ssb_taps(self, fs):
        taps = np.empty(65)
        self._ck(self.lib.pss_get_ssb_taps(self.h, float(fs), _ptr(taps)))
        return taps

    def set_ssb_taps(self, fs, taps):
        taps = np.ascontiguousarray(taps, np.float64)
        assert taps.shape == (65,)
        self._ck(self.lib.pss_set_ssb_taps(self.h, float(fs), _ptr(taps)))

    # -- batched device entry points (asynchronous on the engine's stream)
    def spectrum_db(self, d_iq, n_frames, n_fft, d_db):
        self._dev(self.lib.pss_spectrum_db, _ptr(d_iq), n_frames, n_fft, _ptr(d_db))

    def spectrum_post(self, d_db, n_frames, n_fft, d_post):
        self._dev(self.lib.pss_spectrum_post, _ptr(d_db), n_frames, n_fft, _ptr(d_post))

    def spectrum_post_extremes(self, d_db, n_frames, n_fft, d_post, d_row_lo, d_row_hi):
        """Post-process + the finite min / max of every post-processed row (inputs of waterfall_rows / persistence_rows)."""
        self._dev(self.lib.pss_spectrum_post_extremes, _ptr(d_db), n_frames, n_fft, _ptr(d_post), _ptr(d_row_lo),
                                                     _ptr(d_row_hi))

    def spectrum_db_post(self, d_iq, n_frames, n_fft, d_db, d_post, d_row_lo=None, d_row_hi=None):
        """compute_fft + post-process (+ row extremes) in one call; one fused kernel for 1024-point frames."""
        self._dev(self.lib.pss_spectrum_db_post, _ptr(d_iq), n_frames, n_fft, _ptr(d_db), _ptr(d_post), _ptr(d_row_lo),
                                               _ptr(d_row_hi))

    def row_extremes(self, d_rows, n_rows, length, d_row_lo, d_row_hi, f64=False):
        fn = self.lib.pss_row_extremes_f64 if f64 else self.lib.pss_row_extremes
        self._dev(fn, _ptr(d_rows), n_rows, length, _ptr(d_row_lo), _ptr(d_row_hi))

    def waterfall_rows(self, d_post, n_frames, length, d_row_lo, d_row_hi, disp_w, d_glyph, d_colour, n_halo=0, window=30,
                       f64=False):
        """Batched waterfall accumulator: the newest display line of every frame (history of `window` rows)."""
        fn = self.lib.pss_waterfall_rows_f64 if f64 else self.lib.pss_waterfall_rows
        self._dev(fn, _ptr(d_post), n_frames, length, _ptr(d_row_lo), _ptr(d_row_hi), n_halo, window, disp_w,
                    _ptr(d_glyph), _ptr(d_colour))

    def persistence_rows(self, d_post, n_frames, length, d_row_lo, d_row_hi, disp_h, disp_w, d_y, n_halo=0, window=10,
                         f64=False):
        """Batched persistence accumulator: the newest trace's row index per column for every frame."""
        fn = self.lib.pss_persistence_rows_f64 if f64 else self.lib.pss_persistence_rows
        self._dev(fn, _ptr(d_post), n_frames, length, _ptr(d_row_lo), _ptr(d_row_hi), n_halo, window, disp_h, disp_w,
                    _ptr(d_y))

    def spectrum_post_thresholds(self, d_db, n_frames, n_fft, d_row_thr, d_row_lo, d_row_hi):
        """The post-process without writing the rows: clamp threshold and finite extremes per row (inputs of *_rows_db)."""
        self._dev(self.lib.pss_spectrum_post_thresholds, _ptr(d_db), n_frames, n_fft, _ptr(d_row_thr), _ptr(d_row_lo), _ptr(d_row_hi))

    def waterfall_rows_db(self, d_db, n_frames, n_fft, d_row_thr, d_row_lo, d_row_hi, disp_w, d_glyph, d_colour, n_halo=0, window=30):
        """waterfall_rows from the dB rows + clamp thresholds (the post-processed rows are never written)."""
        self._dev(self.lib.pss_waterfall_rows_db, _ptr(d_db), n_frames, n_fft, _ptr(d_row_thr), _ptr(d_row_lo), _ptr(d_row_hi), n_halo,
                                                window, disp_w, _ptr(d_glyph), _ptr(d_colour))

    def persistence_rows_db(self, d_db, n_frames, n_fft, d_row_thr, d_row_lo, d_row_hi, disp_h, disp_w, d_y, n_halo=0, window=10):
        self._dev(self.lib.pss_persistence_rows_db, _ptr(d_db), n_frames, n_fft, _ptr(d_row_thr), _ptr(d_row_lo), _ptr(d_row_hi), n_halo,
                                                  window, disp_h, disp_w, _ptr(d_y))

    def scan(self, d_iq, n_slices, n_fft, fs, d_db, d_peak, d_bw, d_count):
        self._dev(self.lib.pss_scan, _ptr(d_iq), n_slices, n_fft, float(fs), _ptr(d_db), _ptr(d_peak),
                                   _ptr(d_bw), _ptr(d_count))

    def scan_threshold(self, d_iq, n_slices, n, fs, threshold_db, d_db=None, d_peak=None, d_bw=None, d_count=None):
        """The sweep driver's per-read numbers (pyspecsdr.py:1049-1057): max power, bins above an absolute threshold, bandwidth."""
        self._dev(self.lib.pss_scan_threshold, _ptr(d_iq), n_slices, n, float(fs), float(threshold_db), _ptr(d_db), _ptr(d_peak),
                                             _ptr(d_bw), _ptr(d_count))

    def hilbert(self, d_x, n_rows, n, d_analytic):
        """scipy.signal.hilbert along float64 rows -> complex128 rows (n: power of two in 256..1048576)."""
        self._dev(self.lib.pss_hilbert, _ptr(d_x), n_rows, n, _ptr(d_analytic))

    def power_db(self, d_iq, n_frames, n, d_power):
        self._dev(self.lib.pss_power_db, _ptr(d_iq), n_frames, n, _ptr(d_power))

    def iq_correction(self, d_iq, n_frames, n, d_out_iq=None, d_raw=None):
        self._dev(self.lib.pss_iq_correction, _ptr(d_iq), n_frames, n, _ptr(d_out_iq), _ptr(d_raw))

    def agc_steps(self, d_power, n, start_idx, n_gains, d_idx):
        self._dev(self.lib.pss_agc_steps, _ptr(d_power), n, start_idx, n_gains, _ptr(d_idx))

    def demod(self, mode, d_iq, n_frames, n, fs, d_pcm=None, d_audio=None):
        self._dev(self.lib.pss_demod, mode, _ptr(d_iq), n_frames, n, float(fs), _ptr(d_pcm), _ptr(d_audio))

    def demod_signal(self, mode, d_iq, n_frames, n, fs, d_pcm=None, d_audio=None):
        """Dispatcher semantics (demodulate_signal): WFM frames are IQ-corrected first."""
        self._dev(self.lib.pss_demod_signal, mode, _ptr(d_iq), n_frames, n, float(fs), _ptr(d_pcm), _ptr(d_audio))

    def demod_power(self, mode, d_iq, n_frames, n, fs, d_pcm, d_audio, d_power):
        """measure_signal_power + demodulate of the same read buffers (pyspecsdr.py:2251, :2262); AM: one pass over the IQ for both means."""
        self._dev(self.lib.pss_demod_power, mode, _ptr(d_iq), n_frames, n, float(fs), _ptr(d_pcm), _ptr(d_audio), _ptr(d_power))

    def wfm_filters(self, fs):
        lp, pil, lmr = np.empty((3, 6)), np.empty((5, 6)), np.empty((5, 6))
        import ctypes as C
        a = C.c_double()
        self._ck(self.lib.pss_get_wfm_filters(self.h, float(fs), _ptr(lp), _ptr(pil), _ptr(lmr), C.addressof(a)))   # host-only: no stream ordering
        return lp, pil, lmr, a.value

    def set_wfm_filters(self, fs, lp, pilot, lmr, alpha):
        c = lambda x: np.ascontiguousarray(x, np.float64)
        lp, pilot, lmr = c(lp), c(pilot), c(lmr)
        assert lp.shape == (3, 6) and pilot.shape == (5, 6) and lmr.shape == (5, 6)
        self._ck(self.lib.pss_set_wfm_filters(self.h, float(fs), _ptr(lp), _ptr(pilot), _ptr(lmr), float(alpha)))

    def demod_out_len(self, mode, n, fs):
        """Output samples per frame at this engine's target rate (set_target_rate; 22050 unless changed)."""
        return int(self.lib.pss_demod_out_len_ctx(self.h, mode, n, float(fs)))

    def set_target_rate(self, target_rate):
        """demodulate_nfm / demodulate_wfm's target_rate (signal_processing.py:91, :119): decimation factor int(fs / target_rate)."""
        self._ck(self.lib.pss_set_target_rate(self.h, float(target_rate)))

    def spectrum_nfm(self, d_iq, n_frames, n, fs, d_db, d_pcm):
        self._dev(self.lib.pss_spectrum_nfm, _ptr(d_iq), n_frames, n, float(fs), _ptr(d_db), _ptr(d_pcm))

    def frame_pipeline_nfm(self, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, disp_w, d_glyph, d_colour, d_pcm,
                           n_halo=0, window=30):
        """One main-loop iteration for a batch of read buffers: NFM -> int16, dB row, post-processed row (+ extremes),
        waterfall line (pyspecsdr.py:2262-2283 + draw_waterfall).  d_post=None: the post-processed rows are not materialised."""
        self._dev(self.lib.pss_frame_pipeline_nfm, _ptr(d_iq), n_frames, n, float(fs), _ptr(d_db), _ptr(d_post), _ptr(d_row_lo),
                                                 _ptr(d_row_hi), n_halo, window, disp_w, _ptr(d_glyph), _ptr(d_colour), _ptr(d_pcm))

    def frame_pipeline(self, mode, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, disp_w, d_line_a, d_line_b, d_pcm,
                       n_halo=0, window=None, display="waterfall", disp_h=36):
        """One main-loop iteration per read buffer in any demodulation mode (dispatcher semantics: WFM is IQ-corrected first) and for either
        batched display accumulator: display "waterfall" -> (glyph, colour) lines, "persistence" -> the newest trace's row index (d_line_b
        unused).  window defaults to the reference's history length (30 / 10)."""
        disp = {"waterfall": 0, "persistence": 1}[display]
        window = (30, 10)[disp] if window is None else int(window)
        self._dev(self.lib.pss_frame_pipeline, int(mode), _ptr(d_iq), n_frames, n, float(fs), _ptr(d_db), _ptr(d_post), _ptr(d_row_lo),
                                             _ptr(d_row_hi), n_halo, window, disp, disp_h, disp_w, _ptr(d_line_a), _ptr(d_line_b), _ptr(d_pcm))

    def frame_pipeline_nfm_f64(self, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, disp_w, d_glyph, d_colour, d_pcm,
                               n_halo=0, window=30):
        """frame_pipeline_nfm with the reference's own row type: float64 dB rows, post-processed rows and extremes; the waterfall lines
        are then the cells the reference draws from this IQ (compute_fft returns float64, signal_processing.py:243-264)."""
        self._dev(self.lib.pss_frame_pipeline_nfm_f64, _ptr(d_iq), n_frames, n, float(fs), _ptr(d_db), _ptr(d_post), _ptr(d_row_lo),
                                                     _ptr(d_row_hi), n_halo, window, disp_w, _ptr(d_glyph), _ptr(d_colour), _ptr(d_pcm))

    def frame_pipeline_f64(self, mode, d_iq, n_frames, n, fs, d_db, d_post, d_row_lo, d_row_hi, disp_w, d_line_a, d_line_b, d_pcm,
                           n_halo=0, window=None, display="waterfall", disp_h=36):
        """frame_pipeline with float64 rows: any mode, waterfall line or persistence trace — the reference's cells."""
        disp = {"waterfall": 0, "persistence": 1}[display]
        window = (30, 10)[disp] if window is None else int(window)
        self._dev(self.lib.pss_frame_pipeline_f64, int(mode), _ptr(d_iq), n_frames, n, float(fs), _ptr(d_db), _ptr(d_post), _ptr(d_row_lo),
                  _ptr(d_row_hi), n_halo, window, disp, disp_h, disp_w, _ptr(d_line_a), _ptr(d_line_b), _ptr(d_pcm))

    def frame_pipeline_cells(self, mode, d_iq, n_frames, n, fs, d_db32, d_db64, d_row_lo, d_row_hi, disp_w, d_line_a, d_line_b, d_pcm,
                             n_halo=0, window=None, display="waterfall", disp_h=36):
        """The cell-exact iteration (float64 from the IQ to the cells, frame_pipeline_f64's results) with the dB rows written as float32
        (d_db32: compute_fft's float64 value rounded once) and, if d_db64 is not None, as float64 too.  1024-point frames: the transform and
        the post-process are one kernel and the float64 rows never go through HBM unless d_db64 asks for them."""
        disp = {"waterfall": 0, "persistence": 1}[display]
        window = (30, 10)[disp] if window is None else int(window)
        self._dev(self.lib.pss_frame_pipeline_cells, int(mode), _ptr(d_iq), n_frames, n, float(fs), _ptr(d_db32), _ptr(d_db64), _ptr(d_row_lo),
                  _ptr(d_row_hi), n_halo, window, disp, disp_h, disp_w, _ptr(d_line_a), _ptr(d_line_b), _ptr(d_pcm))

    def spectrum_cells(self, d_iq, n_frames, n, d_db32, d_db64, d_row_lo, d_row_hi, disp_w, d_line_a, d_line_b, n_halo=0, window=None,
                       display="waterfall", disp_h=36):
        """frame_pipeline_cells' display half alone: compute_fft -> post-process -> display line of every frame (no demodulator)."""
        disp = {"waterfall": 0, "persistence": 1}[display]
        window = (30, 10)[disp] if window is None else int(window)
        self._dev(self.lib.pss_spectrum_cells, _ptr(d_iq), n_frames, n, _ptr(d_db32), _ptr(d_db64), _ptr(d_row_lo), _ptr(d_row_hi), n_halo, window,
                  disp, disp_h, disp_w, _ptr(d_line_a), _ptr(d_line_b))

    def spectrum_db_f64(self, d_iq, n_frames, n_fft, d_db):
        """compute_fft's float64 rows (n_fft: power of two in 16..65536)."""
        self._dev(self.lib.pss_spectrum_db_f64, _ptr(d_iq), n_frames, n_fft, _ptr(d_db))

    def spectrum_post_f64(self, d_db, n_frames, n_fft, d_post, d_row_lo=None, d_row_hi=None):
        """The caller's smoothing + median clamp on float64 rows (pyspecsdr.py:2278-2283), optionally the rows' finite extremes."""
        self._dev(self.lib.pss_spectrum_post_f64, _ptr(d_db), n_frames, n_fft, _ptr(d_post), _ptr(d_row_lo), _ptr(d_row_hi))

    def waterfall_cells(self, d_rows, n_rows, length, disp_h, disp_w, d_glyph, d_colour, f64=False):
        fn = self.lib.pss_waterfall_cells_f64 if f64 else self.lib.pss_waterfall_cells
        self._dev(fn, _ptr(d_rows), n_rows, length, disp_h, disp_w, _ptr(d_glyph), _ptr(d_colour))

    def spectrogram_cells(self, d_rows, n_rows, length, disp_h, disp_w, d_glyph, d_colour, d_range=None, f64=False):
        fn = self.lib.pss_spectrogram_cells_f64 if f64 else self.lib.pss_spectrogram_cells
        self._dev(fn, _ptr(d_rows), n_rows, length, disp_h, disp_w, _ptr(d_glyph), _ptr(d_colour), _ptr(d_range))

    def gradient_cells(self, d_rows, n_rows, length, disp_h, disp_w, d_glyph, d_colour, f64=False):
        fn = self.lib.pss_gradient_cells_f64 if f64 else self.lib.pss_gradient_cells
        self._dev(fn, _ptr(d_rows), n_rows, length, disp_h, disp_w, _ptr(d_glyph), _ptr(d_colour))

    def surface_cells(self, d_row, length, max_h, max_w, d_colour, f64=False):
        fn = self.lib.pss_surface_cells_f64 if f64 else self.lib.pss_surface_cells
        self._dev(fn, _ptr(d_row), length, max_h, max_w, _ptr(d_colour))

    def vector_cells(self, d_iq, n, max_h, max_w, d_grid):
        self._dev(self.lib.pss_vector_cells, _ptr(d_iq), n, max_h, max_w, _ptr(d_grid))

    def morse_edges(self, d_iq, n_frames, n, cap, d_rise, d_fall, d_counts, threshold_db=-20.0):
        self._dev(self.lib.pss_morse_edges, _ptr(d_iq), n_frames, n, float(threshold_db), cap, _ptr(d_rise), _ptr(d_fall),
                                          _ptr(d_counts))

    def classify(self, d_iq, n_frames, n, fs, d_label=None, d_bw=None, d_mi=None, d_flat=None, d_psd=None):
        """classify_signal for a batch (pss_classify): any of label int32 / bw float64 / mi float32 / flat float32 / psd float32 [.,1024]."""
        self._dev(self.lib.pss_classify, _ptr(d_iq), n_frames, n, float(fs), _ptr(d_label), _ptr(d_bw), _ptr(d_mi),
                                       _ptr(d_flat), _ptr(d_psd))

    def class_name(self, label):
        return self.lib.pss_class_name(int(label)).decode()

    def persistence_cells(self, d_rows, n_rows, length, disp_h, disp_w, d_colour, f64=False):
        fn = self.lib.pss_persistence_cells_f64 if f64 else self.lib.pss_persistence_cells
        self._dev(fn, _ptr(d_rows), n_rows, length, disp_h, disp_w, _ptr(d_colour))

    # -- stateful display accumulators (device ring of the last rows)
    def ring_create(self, max_rows, length):
        h = C.c_void_p()
        self._ck(self.lib.pss_ring_create(self.h, max_rows, length, C.byref(h)))
        return h

    def _ring(self, fn, ring, *args):
        cs = self._caller_stream()
        if cs is not None:
            self._ck(self.lib.pss_order_after(self.h, cs))
        r = fn(ring, *args)
        if cs is not None:
            self.lib.pss_order_before(self.h, cs)
        self._ck(r)

    def ring_destroy(self, ring):
        self.lib.pss_ring_destroy(ring)

    def ring_push(self, ring, d_row):
        self._ring(self.lib.pss_ring_push, ring, _ptr(d_row))

    def ring_waterfall(self, ring, disp_h, disp_w, d_glyph, d_colour):
        self._ring(self.lib.pss_ring_waterfall, ring, disp_h, disp_w, _ptr(d_glyph), _ptr(d_colour))

    def ring_persistence(self, ring, disp_h, disp_w, d_colour):
        self._ring(self.lib.pss_ring_persistence, ring, disp_h, disp_w, _ptr(d_colour))

    # -- streamed capture from host memory (chunked, double-buffered H2D / compute / D2H)
    def pinned_empty(self, shape, dtype):
        """numpy array backed by pinned host memory (hipHostMalloc); keep a reference to it while in use."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        p = self.lib.pss_host_alloc(nbytes)
        if not p:
            raise MemoryError("hipHostMalloc failed")
        buf = (C.c_char * nbytes).from_address(p)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def pinned_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self.lib.pss_host_free(p)

    def stream_display_nfm(self, h_iq, fs, chunk_frames, mode="waterfall", window=None, disp_h=36, disp_w=112, halo=None,
                           want_db=False, out=None):
        """Stream a host capture (complex64 [n_frames, n], pinned for overlap) and get back, per frame, the display
        accumulator's newest line and the int16 PCM (BASELINE configs[4]).  mode "waterfall": lines = (glyph, colour),
        history 30; "persistence": lines = (y,), history 10.  halo: (lo, hi) float32 arrays of the rows preceding the
        capture.  Returns dict(lines=..., pcm=..., row_lo=..., row_hi=..., db=... or None)."""
        assert h_iq.dtype == np.complex64 and h_iq.ndim == 2 and h_iq.flags.c_contiguous
        nf, n = h_iq.shape
        m = 0 if mode == "waterfall" else 1
        window = (30 if m == 0 else 10) if window is None else int(window)
        n_out = self.demod_out_len(L.MODE_NFM, n, fs)
        # out: a previous call's result dict to write into again (pass arrays from pinned_empty() to keep the downloads
        # asynchronous: a copy into pageable memory is staged by the runtime and blocks the pipeline)
        if out is not None:
            la, lb = (out["lines"] + (None,))[:2]
            pcm, db, lo, hi = out["pcm"], out.get("db"), out["row_lo"], out["row_hi"]
            assert la.shape == (nf, disp_w) and pcm.shape == (nf, n_out, 2) and len(lo) == nf and len(hi) == nf
        else:
            la = np.empty((nf, disp_w), np.int8)
            lb = np.empty((nf, disp_w), np.int8) if m == 0 else None
            pcm = np.empty((nf, n_out, 2), np.int16)
            db = np.empty((nf, n), np.float32) if want_db else None
            lo, hi = np.empty(nf, np.float32), np.empty(nf, np.float32)
        hl = hh = None
        n_halo = 0
        if halo is not None and len(halo[0]):
            hl, hh = np.ascontiguousarray(halo[0], np.float32), np.ascontiguousarray(halo[1], np.float32)
            n_halo = len(hl)
        self._ck(self.lib.pss_h_stream_display_nfm(self.h, _ptr(h_iq), nf, n, float(fs), int(chunk_frames), m, window, disp_h, disp_w,
                                                   _ptr(hl), _ptr(hh), n_halo, _ptr(la), _ptr(lb), _ptr(pcm), _ptr(db), _ptr(lo), _ptr(hi)))
        return {"lines": (la, lb) if m == 0 else (la,), "pcm": pcm, "row_lo": lo, "row_hi": hi, "db": db}

    def stream_display_nfm_f64(self, h_iq, fs, chunk_frames, mode="waterfall", window=None, disp_h=36, disp_w=112, halo=None, want_db=False,
                               grids=False, out=None):
        """stream_display_nfm with compute_fft's own float64 rows from the transform to the cells: the lines (and, grids=True, the full screen
        after every chunk) the REFERENCE draws from this capture.  halo / row_lo / row_hi / db: float64.  PCIe-bound like the float32 call."""
        assert h_iq.dtype == np.complex64 and h_iq.ndim == 2 and h_iq.flags.c_contiguous
        nf, n = h_iq.shape
        m = 0 if mode == "waterfall" else 1
        window = (30 if m == 0 else 10) if window is None else int(window)
        n_out = self.demod_out_len(L.MODE_NFM, n, fs)
        n_chunks = (nf + int(chunk_frames) - 1) // int(chunk_frames) if nf else 0
        if out is not None:      # a previous call's dict / arrays from pinned_empty(): downloads into pinned memory stay asynchronous
            la, lb = (out["lines"] + (None,))[:2]
            pcm, db, lo, hi = out["pcm"], out.get("db"), out["row_lo"], out["row_hi"]
            assert la.shape == (nf, disp_w) and pcm.shape == (nf, n_out, 2) and lo.dtype == np.float64 and len(lo) == nf and len(hi) == nf
        else:
            la = np.empty((nf, disp_w), np.int8)
            lb = np.empty((nf, disp_w), np.int8) if m == 0 else None
            pcm = np.empty((nf, n_out, 2), np.int16)
            db = np.empty((nf, n), np.float64) if want_db else None
            lo, hi = np.empty(nf, np.float64), np.empty(nf, np.float64)
        ga = np.empty((n_chunks, disp_h, disp_w), np.int8) if grids else None
        gb = np.empty((n_chunks, disp_h, disp_w), np.int8) if grids and m == 0 else None
        hl = hh = None
        n_halo = 0
        if halo is not None and len(halo[0]):
            hl, hh = np.ascontiguousarray(halo[0], np.float64), np.ascontiguousarray(halo[1], np.float64)
            n_halo = len(hl)
        self._ck(self.lib.pss_h_stream_display_nfm_f64(self.h, _ptr(h_iq), nf, n, float(fs), int(chunk_frames), m, window, disp_h, disp_w,
                                                       _ptr(hl), _ptr(hh), n_halo, _ptr(la), _ptr(lb), _ptr(pcm), _ptr(db), _ptr(lo), _ptr(hi),
                                                       _ptr(ga), _ptr(gb)))
        out = {"lines": (la, lb) if m == 0 else (la,), "pcm": pcm, "row_lo": lo, "row_hi": hi, "db": db}
        if grids:
            out["grids"] = (ga, gb) if m == 0 else (ga,)
        return out

    def stream_display_nfm_grids(self, h_iq, fs, chunk_frames, mode="waterfall", window=None, disp_h=36, disp_w=112):
        """stream_display_nfm for a capture with a fresh history, plus the FULL display grid after the last frame of every chunk (what the
        reference's screen shows: every line / trace of the history redrawn with the current extremes).  Returns the stream_display_nfm dict
        + "grids": (glyph, colour) int8 [n_chunks][disp_h][disp_w] for the waterfall, (colour,) for persistence."""
        assert h_iq.dtype == np.complex64 and h_iq.ndim == 2 and h_iq.flags.c_contiguous
        nf, n = h_iq.shape
        m = 0 if mode == "waterfall" else 1
        window = (30 if m == 0 else 10) if window is None else int(window)
        n_out = self.demod_out_len(L.MODE_NFM, n, fs)
        n_chunks = (nf + int(chunk_frames) - 1) // int(chunk_frames)
        la = np.empty((nf, disp_w), np.int8)
        lb = np.empty((nf, disp_w), np.int8) if m == 0 else None
        pcm = np.empty((nf, n_out, 2), np.int16)
        lo, hi = np.empty(nf, np.float32), np.empty(nf, np.float32)
        ga = np.empty((n_chunks, disp_h, disp_w), np.int8)
        gb = np.empty((n_chunks, disp_h, disp_w), np.int8) if m == 0 else None
        self._ck(self.lib.pss_h_stream_display_nfm_grids(self.h, _ptr(h_iq), nf, n, float(fs), int(chunk_frames), m, window, disp_h, disp_w,
                                                         _ptr(la), _ptr(lb), _ptr(pcm), _ptr(lo), _ptr(hi), _ptr(ga), _ptr(gb)))
        return {"lines": (la, lb) if m == 0 else (la,), "pcm": pcm, "row_lo": lo, "row_hi": hi, "grids": (ga, gb) if m == 0 else (ga,)}

    def stream_spectrum_nfm(self, h_iq, fs, chunk_frames, h_db=None, h_pcm=None):
        """h_iq: complex64 [n_frames, n] host array (pinned for overlap).  Returns (h_db or None, h_pcm)."""
        assert h_iq.dtype == np.complex64 and h_iq.ndim == 2 and h_iq.flags.c_contiguous
        nf, n = h_iq.shape
        n_out = self.demod_out_len(L.MODE_NFM, n, fs)
        if h_pcm is None:
            h_pcm = np.empty((nf, n_out, 2), np.int16)
        self._ck(self.lib.pss_h_stream_spectrum_nfm(self.h, _ptr(h_iq), nf, n, float(fs), int(chunk_frames),
                                                     _ptr(h_db), _ptr(h_pcm)))
        return h_db, h_pcm

    # -- host convenience (single frame, synchronous)
    def h_compute_fft(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64)
        out = np.empty(len(iq), np.float64)
        self._ck(self.lib.pss_h_compute_fft(self.h, _ptr(iq), len(iq), _ptr(out)))
        return out

    def h_compute_fft_c128(self, iq):
        """compute_fft of a complex128 buffer, float64 from the window product on (len(iq): a power of two in 16..65536)."""
        iq = np.ascontiguousarray(iq, np.complex128)
        out = np.empty(len(iq), np.float64)
        self._ck(self.lib.pss_h_compute_fft_c128(self.h, _ptr(iq), len(iq), _ptr(out)))
        return out

    def h_demodulate_am_c128(self, iq):
        """demodulate_am of a complex128 buffer (float64 np.abs / np.mean): (audio float64 (n, 2), pcm int16 (n, 2))."""
        iq = np.ascontiguousarray(iq, np.complex128)
        audio = np.empty((len(iq), 2), np.float64)
        pcm = np.empty((len(iq), 2), np.int16)
        self._ck(self.lib.pss_h_demodulate_am_c128(self.h, _ptr(iq), len(iq), _ptr(audio), _ptr(pcm)))
        return audio, pcm

    def h_demodulate_ssb_c128(self, iq, fs, lower=True):
        """demodulate_ssb of a complex128 buffer (the complex128 convolution on the samples as they are): (audio float64 (n, 2), pcm int16 (n, 2))."""
        iq = np.ascontiguousarray(iq, np.complex128)
        audio = np.empty((len(iq), 2), np.float64)
        pcm = np.empty((len(iq), 2), np.int16)
        self._ck(self.lib.pss_h_demodulate_ssb_c128(self.h, 1 if lower else 0, _ptr(iq), len(iq), float(fs), _ptr(audio), _ptr(pcm)))
        return audio, pcm

    def demod_ssb_c128(self, d_iq, n_frames, n, fs, d_pcm, d_audio, lower=True):
        self._dev(self.lib.pss_demod_ssb_c128, 1 if lower else 0, _ptr(d_iq), n_frames, n, float(fs), _ptr(d_pcm), _ptr(d_audio))

    def h_mean_power_c128(self, iq):
        """np.mean(np.abs(iq) ** 2) of a complex128 buffer in float64 (the array part of measure_signal_power): np.float64."""
        iq = np.ascontiguousarray(iq, np.complex128)
        out = np.empty(1, np.float64)
        self._ck(self.lib.pss_h_mean_power_c128(self.h, _ptr(iq), len(iq), _ptr(out)))
        return out[0]

    def mean_power_c128(self, d_iq, n_frames, n, d_power):
        self._dev(self.lib.pss_mean_power_c128, _ptr(d_iq), n_frames, n, _ptr(d_power))

    def spectrum_db_c128(self, d_iq, n_frames, n_fft, d_db):
        self._dev(self.lib.pss_spectrum_db_c128, _ptr(d_iq), n_frames, n_fft, _ptr(d_db))

    def demod_am_c128(self, d_iq, n_frames, n, d_pcm, d_audio):
        self._dev(self.lib.pss_demod_am_c128, _ptr(d_iq), n_frames, n, _ptr(d_pcm), _ptr(d_audio))

    def h_demodulate(self, mode, iq, fs):
        iq = np.ascontiguousarray(iq, np.complex64)
        n_out = self.demod_out_len(mode, len(iq), fs)
        if n_out < 0:
            raise ValueError("sample rate below the target rate or unknown mode")
        audio = np.empty((n_out, 2), np.float64)
        pcm = np.empty((n_out, 2), np.int16)
        self._ck(self.lib.pss_h_demodulate(self.h, mode, _ptr(iq), len(iq), float(fs), _ptr(audio), _ptr(pcm)))
        return audio, pcm

    def h_demodulate_signal(self, mode, iq, fs):
        iq = np.ascontiguousarray(iq, np.complex64)
        n_out = self.demod_out_len(mode, len(iq), fs)
        if n_out < 0:
            raise ValueError("sample rate below the target rate or unknown mode")
        audio = np.empty((n_out, 2), np.float64)
        pcm = np.empty((n_out, 2), np.int16)
        self._ck(self.lib.pss_h_demodulate_signal(self.h, mode, _ptr(iq), len(iq), float(fs), _ptr(audio), _ptr(pcm)))
        return audio, pcm

    def h_demodulate_batch(self, mode, frames, fs, chunk_frames=4096):
        """frames: complex64 [n_frames][n] in host memory -> int16 PCM [n_frames][n_out][2] (dispatcher semantics)."""
        frames = np.ascontiguousarray(frames, np.complex64)
        nf, n = frames.shape
        n_out = self.demod_out_len(mode, n, fs)
        if n_out < 0:
            raise ValueError("sample rate below the target rate or unknown mode")
        pcm = np.empty((nf, n_out, 2), np.int16)
        self._ck(self.lib.pss_h_demodulate_batch(self.h, mode, _ptr(frames), nf, n, float(fs), int(chunk_frames), _ptr(pcm)))
        return pcm

    def h_iq_correction(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64)
        out = np.empty(len(iq), np.complex64)
        self._ck(self.lib.pss_h_iq_correction(self.h, _ptr(iq), len(iq), _ptr(out), None))
        return out

    def h_morse_edges(self, iq, threshold_db=-20.0):
        """-> (rise_times, fall_times) int32 arrays of decode_morse (decoders.py:159-161) for one buffer."""
        iq = np.ascontiguousarray(iq, np.complex64)
        cap = max(len(iq) // 2 + 1, 1)
        rise, fall = np.empty(cap, np.int32), np.empty(cap, np.int32)
        nr, nf = C.c_int(), C.c_int()
        self._ck(self.lib.pss_h_morse_edges(self.h, _ptr(iq), len(iq), float(threshold_db), cap, _ptr(rise), _ptr(fall),
                                            C.byref(nr), C.byref(nf)))
        return rise[:nr.value].copy(), fall[:nf.value].copy()

    def h_classify_signal(self, iq, fs):
        """-> (label str, signal_bw float, modulation_index np.float32, spectral_flatness np.float32) for one read buffer."""
        iq = np.ascontiguousarray(iq, np.complex64)
        lab, bw, mi, fl = C.c_int(), C.c_double(), C.c_float(), C.c_float()
        self._ck(self.lib.pss_h_classify_signal(self.h, _ptr(iq), len(iq), float(fs), C.byref(lab), C.byref(bw), C.byref(mi), C.byref(fl)))
        return self.class_name(lab.value), bw.value, np.float32(mi.value), np.float32(fl.value)

    def h_raw(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64)
        raw = np.empty(len(iq), np.float32)
        self._ck(self.lib.pss_h_iq_correction(self.h, _ptr(iq), len(iq), None, _ptr(raw)))
        return raw

    def sosfilt(self, d_x, n_rows, n, sos, d_y):
        sos = np.ascontiguousarray(sos, np.float64)
        self._dev(self.lib.pss_sosfilt, _ptr(d_x), n_rows, n, _ptr(sos), sos.shape[0], _ptr(d_y))

    def afsk_bits(self, d_audio, n_rows, n, fs, d_bits, sos1200=None, sos2200=None):
        c = lambda a: None if a is None else np.ascontiguousarray(a, np.float64)
        s1, s2 = c(sos1200), c(sos2200)
        self._dev(self.lib.pss_afsk_bits, _ptr(d_audio), n_rows, n, float(fs), _ptr(s1), _ptr(s2),
                                        5 if s1 is None else s1.shape[0], _ptr(d_bits))

    def h_afsk_bits(self, x, fs, sos1200=None, sos2200=None, normalise=False):
        """decode_afsk's bit list for one host buffer of real audio (float64) -> uint8 array; normalise=True divides by
        max|x| on the device first (what decode_aprs does before it calls decode_afsk, decoders.py:126)."""
        x = np.ascontiguousarray(x, np.float64)
        nb = self.afsk_n_bits(len(x), fs)
        if nb <= 0:
            return np.zeros(0, np.uint8)
        bits = np.empty(nb, np.uint8)
        s1 = None if sos1200 is None else np.ascontiguousarray(sos1200, np.float64)
        s2 = None if sos2200 is None else np.ascontiguousarray(sos2200, np.float64)
        self._ck(self.lib.pss_h_afsk_bits(self.h, _ptr(x), len(x), float(fs), int(bool(normalise)), _ptr(s1), _ptr(s2),
                                          0 if s1 is None else s1.shape[0], _ptr(bits)))
        return bits

    def np_f32(self, op, d_a, d_b, n, d_out):
        """NumPy's float32 arctan2 (op 0) / log10 (1) / abs of a + ib (2), element by element (pss_np_f32)."""
        self._dev(self.lib.pss_np_f32, int(op), _ptr(d_a), _ptr(d_b), int(n), _ptr(d_out))

    def row_normalise(self, d_x, n_rows, n, d_y):
        self._dev(self.lib.pss_row_normalise, _ptr(d_x), n_rows, n, _ptr(d_y))

    def afsk_n_bits(self, n, fs):
        return self.lib.pss_afsk_n_bits(int(n), float(fs))

    def h_bandpass_filter(self, data, lowcut, highcut, fs, sos=None):
        x = np.ascontiguousarray(data, np.float64)
        y = np.empty_like(x)
        if sos is not None:
            sos = np.ascontiguousarray(sos, np.float64)
        self._ck(self.lib.pss_h_bandpass_filter(self.h, _ptr(x), len(x), float(lowcut), float(highcut), float(fs),
                                                _ptr(sos), 0 if sos is None else sos.shape[0], _ptr(y)))
        return y

    def h_measure_power(self, iq):
        iq = np.ascontiguousarray(iq, np.complex64)
        out = np.empty(1, np.float32)
        self._ck(self.lib.pss_h_measure_power(self.h, _ptr(iq), len(iq), _ptr(out)))
        return out[0]
