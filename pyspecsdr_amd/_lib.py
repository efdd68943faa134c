"""ctypes loader for libpss.so — the C ABI declared in include/pss.h.

There is no CPU fallback: if the HIP library is missing or no GPU is visible, creating an Engine raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PSS_LIBRARY: load another build of the same ABI (kernel experiments: tools/build_variant.py)
LIB_PATH = os.environ.get("PSS_LIBRARY") or os.path.join(HERE, "libpss.so")

PSS_OK, PSS_E_ARG, PSS_E_HIP, PSS_E_PADLEN, PSS_E_CUTOFF, PSS_E_NOMEM, PSS_E_COMM = 0, -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128
MODE_NFM, MODE_AM, MODE_USB, MODE_LSB, MODE_WFM = 0, 1, 2, 3, 4
NP_ARCTAN2, NP_LOG10, NP_ABS = 0, 1, 2          # pss_np_f32 operations

_p = C.c_void_p
_SIGS = {
    "pss_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "pss_destroy": (None, [_p]),
    "pss_set_stream": (C.c_int, [_p, _p]),
    "pss_get_stream": (_p, [_p]),
    "pss_order_after": (C.c_int, [_p, _p]),
    "pss_order_before": (C.c_int, [_p, _p]),
    "pss_sync": (C.c_int, [_p]),
    "pss_last_error": (C.c_char_p, [_p]),
    "pss_device_count": (C.c_int, []),
    "pss_set_option": (C.c_int, [_p, C.c_char_p, C.c_int]),
    "pss_design_firwin": (C.c_int, [C.c_int, C.c_double, _p]),
    "pss_design_cheby1_sos": (C.c_int, [C.c_int, C.c_double, C.c_double, _p]),
    "pss_design_sosfilt_zi": (C.c_int, [_p, C.c_int, _p]),
    "pss_design_butter_sos": (C.c_int, [C.c_int, C.c_double, C.c_double, _p, _p]),
    "pss_am_bandpass_sos": (C.c_int, [_p]),
    "pss_set_nfm_filters": (C.c_int, [_p, C.c_double, _p, _p, _p]),
    "pss_set_ssb_taps": (C.c_int, [_p, C.c_double, _p]),
    "pss_get_nfm_filters": (C.c_int, [_p, C.c_double, _p, _p, _p]),
    "pss_get_ssb_taps": (C.c_int, [_p, C.c_double, _p]),
    "pss_spectrum_db": (C.c_int, [_p, _p, C.c_long, C.c_int, _p]),
    "pss_spectrum_post": (C.c_int, [_p, _p, C.c_long, C.c_int, _p]),
    "pss_spectrum_post_extremes": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, _p]),
    "pss_spectrum_db_post": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, _p, _p]),
    "pss_row_extremes": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p]),
    "pss_row_extremes_f64": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p]),
    "pss_waterfall_rows": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pss_waterfall_rows_f64": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pss_persistence_rows": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "pss_persistence_rows_f64": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "pss_spectrum_post_thresholds": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, _p]),
    "pss_waterfall_rows_db": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, _p, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pss_persistence_rows_db": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "pss_scan": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, _p, _p, _p, _p]),
    "pss_scan_threshold": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, C.c_double, _p, _p, _p, _p]),
    "pss_hilbert": (C.c_int, [_p, _p, C.c_long, C.c_int, _p]),
    "pss_power_db": (C.c_int, [_p, _p, C.c_long, C.c_int, _p]),
    "pss_iq_correction": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p]),
    "pss_agc_steps": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_int, _p]),
    "pss_demod": (C.c_int, [_p, C.c_int, _p, C.c_long, C.c_int, C.c_double, _p, _p]),
    "pss_demod_signal": (C.c_int, [_p, C.c_int, _p, C.c_long, C.c_int, C.c_double, _p, _p]),
    "pss_demod_power": (C.c_int, [_p, C.c_int, _p, C.c_long, C.c_int, C.c_double, _p, _p, _p]),
    "pss_set_wfm_filters": (C.c_int, [_p, C.c_double, _p, _p, _p, C.c_double]),
    "pss_get_wfm_filters": (C.c_int, [_p, C.c_double, _p, _p, _p, _p]),
    "pss_demod_out_len": (C.c_int, [C.c_int, C.c_int, C.c_double]),
    "pss_set_target_rate": (C.c_int, [_p, C.c_double]),
    "pss_demod_out_len_ctx": (C.c_int, [_p, C.c_int, C.c_int, C.c_double]),
    "pss_demod_out_len_rate": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double]),
    "pss_spectrum_nfm": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, _p, _p]),
    "pss_frame_pipeline_nfm": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "pss_frame_pipeline": (C.c_int, [_p, C.c_int, _p, C.c_long, C.c_int, C.c_double, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "pss_frame_pipeline_nfm_f64": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "pss_frame_pipeline_f64": (C.c_int, [_p, C.c_int, _p, C.c_long, C.c_int, C.c_double, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "pss_spectrum_db_c128": (C.c_int, [_p, _p, C.c_long, C.c_int, _p]),
    "pss_demod_am_c128": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p]),
    "pss_h_compute_fft_c128": (C.c_int, [_p, _p, C.c_int, _p]),
    "pss_h_demodulate_am_c128": (C.c_int, [_p, _p, C.c_int, _p, _p]),
    "pss_demod_ssb_c128": (C.c_int, [_p, C.c_int, _p, C.c_long, C.c_int, C.c_double, _p, _p]),
    "pss_h_demodulate_ssb_c128": (C.c_int, [_p, C.c_int, _p, C.c_int, C.c_double, _p, _p]),
    "pss_mean_power_c128": (C.c_int, [_p, _p, C.c_long, C.c_int, _p]),
    "pss_h_mean_power_c128": (C.c_int, [_p, _p, C.c_int, _p]),
    "pss_frame_pipeline_cells": (C.c_int, [_p, C.c_int, _p, C.c_long, C.c_int, C.c_double, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "pss_spectrum_cells": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pss_spectrum_db_f64": (C.c_int, [_p, _p, C.c_long, C.c_int, _p]),
    "pss_spectrum_post_f64": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, _p, _p]),
    "pss_h_np_f64": (C.c_int, [C.c_int, _p, C.c_long, _p]),
    "pss_waterfall_cells": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pss_persistence_cells": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "pss_spectrogram_cells": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "pss_spectrogram_cells_f64": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "pss_gradient_cells": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pss_gradient_cells_f64": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pss_surface_cells": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p]),
    "pss_surface_cells_f64": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p]),
    "pss_vector_cells": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _p]),
    "pss_morse_edges": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, C.c_int, _p, _p, _p]),
    "pss_h_morse_edges": (C.c_int, [_p, _p, C.c_int, C.c_double, C.c_int, _p, _p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pss_h_morse_decode": (C.c_int, [_p, C.c_long, _p, C.c_long, C.c_double, _p, C.c_long, _p]),
    "pss_h_ax25_frame": (C.c_int, [_p, C.c_long, _p, C.c_long, C.POINTER(C.c_long)]),
    "pss_h_decode_morse": (C.c_int, [_p, _p, C.c_int, C.c_double, C.c_double, _p, C.c_long, _p]),
    "pss_h_decode_aprs": (C.c_int, [_p, _p, C.c_int, C.c_double, _p, _p, C.c_int, _p, C.c_long, C.POINTER(C.c_long)]),
    "pss_classify": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, _p, _p, _p, _p, _p]),
    "pss_class_name": (C.c_char_p, [C.c_int]),
    "pss_h_classify_signal": (C.c_int, [_p, _p, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float),
                                        C.POINTER(C.c_float)]),
    "pss_waterfall_cells_f64": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "pss_persistence_cells_f64": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p]),
    "pss_ring_create": (C.c_int, [_p, C.c_int, C.c_int, C.POINTER(_p)]),
    "pss_ring_destroy": (None, [_p]),
    "pss_ring_push": (C.c_int, [_p, _p]),
    "pss_ring_count": (C.c_int, [_p]),
    "pss_ring_waterfall": (C.c_int, [_p, C.c_int, C.c_int, _p, _p]),
    "pss_ring_persistence": (C.c_int, [_p, C.c_int, C.c_int, _p]),
    "pss_h_compute_fft": (C.c_int, [_p, _p, C.c_int, _p]),
    "pss_h_demodulate": (C.c_int, [_p, C.c_int, _p, C.c_int, C.c_double, _p, _p]),
    "pss_h_demodulate_signal": (C.c_int, [_p, C.c_int, _p, C.c_int, C.c_double, _p, _p]),
    "pss_h_demodulate_batch": (C.c_int, [_p, C.c_int, _p, C.c_long, C.c_int, C.c_double, C.c_long, _p]),
    "pss_h_measure_power": (C.c_int, [_p, _p, C.c_int, _p]),
    "pss_sosfilt": (C.c_int, [_p, _p, C.c_long, C.c_int, _p, C.c_int, _p]),
    "pss_afsk_n_bits": (C.c_int, [C.c_int, C.c_double]),
    "pss_afsk_bits": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, _p, _p, C.c_int, _p]),
    "pss_row_normalise": (C.c_int, [_p, _p, C.c_long, C.c_int, _p]),
    "pss_np_f32": (C.c_int, [_p, C.c_int, _p, _p, C.c_long, _p]),
    "pss_h_afsk_bits": (C.c_int, [_p, _p, C.c_int, C.c_double, C.c_int, _p, _p, C.c_int, _p]),
    "pss_h_bandpass_filter": (C.c_int, [_p, _p, C.c_int, C.c_double, C.c_double, C.c_double, _p, C.c_int, _p]),
    "pss_h_iq_correction": (C.c_int, [_p, _p, C.c_int, _p, _p]),
    "pss_host_alloc": (_p, [C.c_size_t]),
    "pss_host_free": (None, [_p]),
    "pss_h_stream_spectrum_nfm": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, C.c_long, _p, _p]),
    "pss_h_stream_display_nfm": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p,
                                           C.c_int, _p, _p, _p, _p, _p, _p]),
    "pss_h_stream_display_nfm_grids": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p,
                                                 _p, _p, _p, _p, _p]),
    "pss_h_stream_display_nfm_f64": (C.c_int, [_p, _p, C.c_long, C.c_int, C.c_double, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p,
                                               C.c_int, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pss_shard_range": (C.c_int, [C.c_long, C.c_int, C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "pss_comm_id": (C.c_int, [_p]),
    "pss_comm_init": (C.c_int, [_p, _p, C.c_int, C.c_int]),
    "pss_comm_free": (C.c_int, [_p]),
    "pss_comm_size": (C.c_int, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pss_gather_packed": (C.c_int, [_p, _p, C.c_size_t, _p, C.c_int]),
    "pss_halo_from_left": (C.c_int, [_p, _p, C.POINTER(C.c_long), C.c_size_t, C.c_long, _p, C.POINTER(C.c_long)]),
    "pss_enable_timing": (C.c_int, [_p, C.c_int]),
    "pss_last_kernel_ms": (C.c_float, [_p]),
    "pss_kernel_times": (C.c_int, [_p, C.c_char_p, C.c_int]),
    "pss_timing_filter": (C.c_int, [_p, C.c_char_p]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Load libpss.so (built by pyspecsdr_amd.build).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m pyspecsdr_amd.build` (hipcc, gfx950). "
                "pyspecsdr_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as
        # /opt/rocm's).  If torch is going to be used in this process (device buffers, torch.distributed) it
        # must be loaded FIRST so that libpss.so binds to that copy; two runtimes in one process cannot both
        # open the GPU.  Without torch installed, libpss.so simply uses /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and this table ever drift apart
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
