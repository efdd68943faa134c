"""Pin the CPU oracle (oracle/pss_oracle.c) to the golden vectors captured from the reference.

The reference (xqtr/PySpecSDR) has no tests or fixtures of its own; tests/golden/*.npz were produced by
tools/make_goldens.py importing the reference's signal_processing.py / pyspecsdr.py under
NumPy 2.2.6 + SciPy 1.15.3 (AVX512_SKX dispatch).  Bit-exact where stated, tolerance elsewhere.
"""
import numpy as np
import pytest

import oracle_lib as O


def bits32(a):
    return np.asarray(a, np.float32).view(np.uint32)


def test_atan2f_bit_exact(golden):
    g = golden["atan2f"]
    y, x, th = g["y"], g["x"], g["theta"]
    mine = O.atan2f(y, x)
    same = (bits32(mine) == bits32(th)) | (np.isnan(mine) & np.isnan(th))
    assert same.all(), f"{(~same).sum()} of {len(same)} differ"
    # operands as random bit patterns (all exponents, denormals, inf, NaN): the library's special-value path
    g = golden["atan2f_bits"]
    mine = O.atan2f(g["y"], g["x"])
    same = (bits32(mine) == bits32(g["theta"])) | (np.isnan(mine) & np.isnan(g["theta"]))
    assert same.all(), f"{(~same).sum()} of {len(same)} differ"


def test_rcp14_model_properties():
    L = O.lib()
    # exact powers of two, monotone non-increasing, relative error < 2^-14 (the architectural bound)
    xs = np.float32(1.0) + np.arange(0, 1 << 23, 257, dtype=np.uint32).astype(np.float32) * np.float32(2.0 ** -23)
    r = np.array([L.pss_o_rcp14f(float(v)) for v in xs], np.float32)
    assert L.pss_o_rcp14f(1.0) == 1.0 and L.pss_o_rcp14f(4.0) == 0.25 and L.pss_o_rcp14f(-0.5) == -2.0
    assert np.all(np.diff(r) <= 0)
    assert np.max(np.abs(r.astype(np.float64) * xs.astype(np.float64) - 1)) < 2.0 ** -14
    # scale invariance in the exponent
    for e in (-100, -7, 30, 100):
        assert L.pss_o_rcp14f(float(np.ldexp(np.float32(1.37), e))) == float(np.ldexp(np.float32(L.pss_o_rcp14f(1.37)), -e))


@pytest.mark.parametrize("n", [256, 1024, 2048, 4096, 8192, 16384])
def test_compute_fft(golden, n):
    g = golden["spectrum"]
    for iq, ref in zip(g[f"iq_{n}"], g[f"db_{n}"]):
        mine = O.compute_fft(iq)
        assert np.allclose(mine, ref, rtol=1e-9, atol=1e-9)
    if n == 1024:
        assert np.array_equal(O.compute_fft(np.zeros(1024, np.complex64)), g["db_zero"])  # exactly -100.0


@pytest.mark.parametrize("n", [256, 1024, 4096])
def test_postprocess(golden, n):
    g = golden["spectrum"]
    for db, ref in zip(g[f"db_{n}"], g[f"post_{n}"]):
        mine = O.postprocess(db)
        assert mine.shape == ref.shape == (n - 4,)
        assert np.allclose(mine, ref, rtol=1e-13, atol=1e-12)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f", "g"])
def test_nfm(golden, tag):
    g = golden["nfm"]
    fs = float(g[f"fs_{tag}"])
    taps, sos, zi = g[f"taps_{tag}"], g[f"sos_{tag}"], g[f"zi_{tag}"]
    for k, iq in enumerate(g[f"iq_{tag}"]):
        audio, disc, fir = O.demod_nfm(iq, fs, taps, sos, zi, stages=True)
        if k == 0:
            # float32 front end (FMA complex multiply + SVML atan2 + scale) is bit-exact
            assert np.array_equal(bits32(disc), bits32(g[f"disc_{tag}"]))
            # FIR in OpenBLAS ddot accumulation order: bit-exact float64, edges included
            assert np.array_equal(fir, g[f"fir_{tag}"])
        ref = g[f"audio_{tag}"][k]
        assert audio.shape == ref.shape
        assert np.array_equal(audio, ref)  # float64 bit-exact
        assert np.array_equal(O.pcm16_stereo(audio), g[f"pcm_{tag}"][k])


def test_nfm_edges(golden):
    g = golden["nfm"]
    taps, sos, zi = g["taps_a"], g["sos_a"], g["zi_a"]
    with pytest.raises(ValueError):  # N-1 <= 27 -> scipy sosfiltfilt padlen ValueError
        O.demod_nfm(np.ones(28, np.complex64), 2.4e6, taps, sos, zi)
    a = O.demod_nfm(np.zeros(1024, np.complex64), 2.4e6, taps, sos, zi)  # silence -> 0/0 -> NaN -> int16 0
    assert np.isnan(a).all() and np.isnan(g["audio_silence"]).all()
    assert np.array_equal(O.pcm16_stereo(a), g["pcm_silence"])


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_am_bit_exact(golden, tag):
    g = golden["am_ssb"]
    sos = g["am_sos"]
    for k, iq in enumerate(g[f"am_iq_{tag}"]):
        audio = O.demod_am(iq, sos)
        assert np.array_equal(audio, g[f"am_audio_{tag}"][k])  # float64 bit-exact
        assert np.array_equal(O.pcm16_stereo(audio), g[f"am_pcm_{tag}"][k])
    L = O.lib()
    iq0 = g[f"am_iq_{tag}"][0]
    env = np.array([L.pss_o_cabsf(float(z.real), float(z.imag)) for z in iq0], np.float32)
    assert np.array_equal(bits32(env), bits32(g[f"am_env_{tag}"]))
    mu = np.float32(L.pss_o_pairwise_sum_f32(env, len(env))) / np.float32(len(env))
    assert bits32(mu) == bits32(g[f"am_mean_{tag}"])


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_ssb(golden, tag):
    g = golden["am_ssb"]
    taps = g[f"ssb_taps_{tag}"]
    for k, iq in enumerate(g[f"ssb_iq_{tag}"]):
        audio = O.demod_ssb(iq, taps)
        n = len(iq)
        if n & (n - 1) == 0:
            # frames of 2^k samples: the real FIR and SciPy's hilbert() round trip (pss_pocketfft.c), every bit of the float64 audio
            assert np.array_equal(audio, g[f"ssb_audio_{tag}"][k]), (tag, k)
        else:
            assert np.allclose(audio, g[f"ssb_audio_{tag}"][k], rtol=0, atol=2e-14)
        assert np.array_equal(O.pcm16_stereo(audio), g[f"ssb_pcm_{tag}"][k])


def test_pocketfft_model_is_scipy_bit_for_bit():
    """oracle/pss_pocketfft.c against SciPy itself (where this test runs): the real forward transform, the complex inverse and
    scipy.signal.hilbert for every power of two up to 2^17, and the plan's twiddle products."""
    import scipy.fft as sf
    import scipy.signal as ss
    L = O.lib()
    rng = np.random.default_rng(11)
    for n in [2 ** k for k in range(1, 18)]:
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4)
        f = np.empty(2 * n)
        L.pss_o_rfft_full(x, n, f)
        assert np.array_equal(f.view(np.complex128), sf.fft(x)), n
        z = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        o = np.empty(2 * n)
        L.pss_o_cifft(np.ascontiguousarray(z).view(np.float64), n, o)
        assert np.array_equal(o.view(np.complex128), sf.ifft(z)), n
        assert np.array_equal(O.hilbert(x), ss.hilbert(x)), n
    assert np.array_equal(O.hilbert(np.zeros(64)), ss.hilbert(np.zeros(64)))


@pytest.mark.parametrize("n", [7, 100, 1024, 16384, 20000, 32768, 40001])
def test_power(golden, n):
    g = golden["power"]
    iq = g[f"iq_{n}"]
    L = O.lib()
    a = np.array([L.pss_o_cabsf(float(z.real), float(z.imag)) for z in iq], np.float32)
    assert bits32(np.float32(L.pss_o_pairwise_sum_f32(a, n)) / np.float32(n)) == bits32(g[f"mean_abs_{n}"])
    s = a * a
    assert bits32(np.float32(L.pss_o_pairwise_sum_f32(s, n)) / np.float32(n)) == bits32(g[f"mean_abs2_{n}"])
    p = O.power_db(iq)
    assert bits32(p) == bits32(g[f"p_{n}"])        # NumPy's float32 log10 is SVML's: modelled bit for bit (test_log10f_model)
    if n == 1024:
        # numpy's float32 log10 loop returns -100.00001 for 1e-10f (not correctly rounded)
        assert bits32(O.power_db(np.zeros(1024, np.complex64))) == bits32(g["p_zero"])


def test_log10f_model(golden):
    """np.log10 on float32 = SVML __svml_log10f16 under the AVX512_SKX dispatch: the model against NumPy's own outputs, every
    bit (tools/check_log10f_model.py runs the same comparison over all 2^31 positive floats in the build container)."""
    g = golden["log10f"]
    x, y = g["x"], g["y"]
    got = O.log10f(x)
    nan = np.isnan(y)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.uint32), y[~nan].view(np.uint32))


@pytest.mark.parametrize("n", [2048, 4096])
def test_scanner(golden, n):
    g = golden["scanner"]
    for k, iq in enumerate(g[f"iq_{n}"]):
        db, pk, bw, cnt = O.scan_slice(iq, 2.4e6)
        ref = g[f"db_{n}"][k]
        # every bit: np.fft.fft on complex64 is a DOUBLE transform rounded to complex64 (NumPy 2.2), and everything behind it is
        # float32 arithmetic that is modelled exactly (np.abs, ** 2, + 1e-10, SVML log10, * 10)
        assert np.array_equal(db.view(np.uint32), ref.view(np.uint32)), (n, k)
        assert np.float32(pk).tobytes() == np.float32(g[f"peak_{n}"][k]).tobytes()
        assert cnt == int(g[f"count_{n}"][k]) and bw == g[f"bw_{n}"][k]


def test_agc(golden):
    g = golden["caller"]
    n = int(g["agc_ngains"])
    for start in (20, 0, n - 1):
        idx, traj = start, []
        for p in g["agc_powers"]:
            idx = O.agc_step(p, idx, n)
            traj.append(idx)
        assert traj == list(g[f"agc_traj_{start}"])


def test_waterfall_and_persistence(golden):
    g = golden["caller"]
    rows = g["rows"]
    H, W = [int(v) for v in g["hw"]]
    dh, dw = H - 4, W - 8
    for i in range(len(rows)):
        ring = rows[max(0, i + 1 - 30):i + 1]
        gl, co = O.waterfall_cells(ring, dh, dw)
        assert np.array_equal(gl, g["wf_glyph"][i]), i
        assert np.array_equal(co, g["wf_colour"][i]), i
    for i in range(len(g["ps_colour"])):
        ring = rows[max(0, i + 1 - 10):i + 1]
        assert np.array_equal(O.persistence_cells(ring, dh, dw), g["ps_colour"][i]), i


def test_batched_accumulator_rows_are_the_newest_line_of_the_pinned_grids(golden):
    """pss_o_waterfall_rows / pss_o_persistence_rows (the checkers of the batched display calls, float32 rows) against the grid functions the
    test above pins to the reference's own draws: waterfall line y = 0; persistence: the newest trace's cells — its colour is unique once the
    history is full (int(1 + 5 (1 - 0.7)) = 2), so the cells holding it are exactly (y[x], x)."""
    g = golden["caller"]
    rows = np.ascontiguousarray(g["rows"].astype(np.float32))
    H, W = [int(v) for v in g["hw"]]
    dh, dw = H - 4, W - 8
    gl, co = O.waterfall_rows(rows, 30, dw)
    ys = O.persistence_rows(rows, 10, dh, dw)
    for i in range(len(rows)):
        g2, c2 = O.waterfall_cells(rows[max(0, i + 1 - 30):i + 1].astype(np.float64), dh, dw)
        assert np.array_equal(gl[i], g2[0]) and np.array_equal(co[i], c2[0]), i
        ring = rows[max(0, i + 1 - 10):i + 1].astype(np.float64)
        cells = O.persistence_cells(ring, dh, dw)
        cp_new = int(1 + 5 * (1 - 0.7 ** (10 - (len(ring) - 1))))
        assert np.all(ys[i] >= 0) and np.all(cells[ys[i], np.arange(dw)] == cp_new), i
        if len(ring) == 10:
            want = np.zeros_like(cells, dtype=bool)
            want[ys[i], np.arange(dw)] = True
            assert np.array_equal(cells == cp_new, want), i


def test_reference_cells_from_iq(golden):
    """caller_iq.npz: the read buffers behind caller.npz's rows.  compute_fft -> smoothing + clamp -> draw_waterfall, the oracle from IQ
    to cells: the float64 rows agree with the reference's to rounding (the transform is not pocketfft's), the cells are equal."""
    g, q = golden["caller"], golden["caller_iq"]
    H, W = [int(v) for v in g["hw"]]
    rows = np.stack([O.postprocess(O.compute_fft(f)) for f in q["iq"]])
    assert np.allclose(rows, g["rows"], rtol=1e-9, atol=1e-9)
    for i in range(len(rows)):
        gl, co = O.waterfall_cells(rows[max(0, i + 1 - 30):i + 1], H - 4, W - 8)
        assert np.array_equal(gl, g["wf_glyph"][i]) and np.array_equal(co, g["wf_colour"][i]), i


def test_raw_and_unknown_modes(golden):
    g = golden["am_ssb"]
    # RAW goes through iq_correction first (signal_processing.py:222-225): float32 (N,), not plain real()
    assert g["raw_out"].dtype == np.float32 and g["raw_out"].shape == (64,)
    assert g["unknown_out"].shape == (64, 2) and not g["unknown_out"].any()


def test_iq_correction_bit_exact(golden):
    """signal_processing.py:46-80 — every float32 bit of the corrected frame, incl. frames longer than numpy's
    8192-element reduce buffer, odd lengths and 8-bit style samples with exact zeros."""
    g = golden["iqcorr"]
    for n in [int(v) for v in g["sizes"]] + ["u8"]:
        got = O.iq_correction(g[f"iq_{n}"])
        want = g[f"corr_{n}"]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), n
        if n != "u8":
            assert np.array_equal(got.real.view(np.uint32), g[f"raw_{n}"].view(np.uint32)), n
    for k, f in enumerate(g["iq_fuzz"]):
        assert np.array_equal(O.iq_correction(f).view(np.uint32), g["corr_fuzz"][k].view(np.uint32)), k


def wfm_filters(g, fs):
    key = str(int(fs))
    return {k: g[f"{k}_{key}"] for k in ("lp_sos", "pilot_sos", "lmr_sos", "alpha", "dec_sos", "dec_zi")}


def test_wfm_bit_exact(golden):
    """demodulate_signal(..., 'WFM') = iq_correction + demodulate_wfm (signal_processing.py:119-176): float64 audio
    bit for bit (both channels — they differ in the last bits at low sample rates), hence int16 too."""
    g = golden["wfm"]
    assert float(g["sin_pi"]) == float.fromhex("0x1.1a62633145c07p-53")
    for tag in g["tags"]:
        fs = float(g[f"fs_{tag}"])
        filt = wfm_filters(g, fs)
        for f, iq in enumerate(g[f"iq_{tag}"]):
            a = O.demod_wfm(O.iq_correction(iq), fs, filt)
            want = g[f"audio_{tag}"][f]
            assert a.shape == want.shape, tag
            assert np.array_equal(a.view(np.uint64), want.view(np.uint64)), (tag, f, np.abs(a - want).max())
            assert np.array_equal(np.int16(a * 32767), g[f"pcm_{tag}"][f])


def test_bandpass_filter_bit_exact(golden):
    g = golden["bandpass"]
    for tag in g["tags"]:
        y = O.sosfilt(g[f"sos_{tag}"], g[f"x_{tag}"])
        assert np.array_equal(y.view(np.uint64), g[f"y_{tag}"].view(np.uint64)), tag


def test_spectrogram_cells(golden):
    """draw_spectrogram (pyspecsdr.py:398-498): the final glyph / colour grid, incl. a full 32768-sample read buffer."""
    g = golden["caller"]
    for tag in g["sg_tags"]:
        hh, ww = [int(v) for v in g[f"sg_hw_{tag}"]]
        gl, co, dmin, dmax = O.spectrogram_cells(g[f"sg_row_{tag}"], hh - 4, ww - 7)
        assert np.array_equal(gl, g[f"sg_glyph_{tag}"]), tag
        assert np.array_equal(co, g[f"sg_colour_{tag}"]), tag
        assert dmin < dmax


def test_gradient_waterfall_and_surface_cells(golden):
    g = golden["caller"]
    rows = g["rows"]
    H, W = [int(v) for v in g["hw"]]
    for i in range(len(g["gw_glyph"])):
        ring = rows[max(0, i + 1 - 30):i + 1]
        gl, co = O.gradient_cells(ring, H - 4, W - 10)
        assert np.array_equal(gl, g["gw_glyph"][i]) and np.array_equal(co, g["gw_colour"][i]), i
    assert [float(v).hex() for v in g["sf_cos_sin"]] == ["0x1.6a09e667f3bcdp-1", "0x1.6a09e667f3bccp-1"]
    for i, (row, hh, ww) in enumerate(((rows[0], 40, 120), (rows[5], 40, 120), (g["sg_row_big"], 50, 200))):
        assert np.array_equal(O.surface_cells(row, hh, ww), g[f"sf_colour_{i}"]), i


def test_vector_display_cells(golden):
    g = golden["caller"]
    for tag, hh, ww in (("a", 40, 120), ("b", 25, 81)):
        assert np.array_equal(O.vector_cells(g["vec_iq"], hh, ww), g[f"vec_grid_{tag}"]), tag


def test_afsk_bits(golden):
    """decode_afsk (decoders.py:94-112): the bit list, incl. bit periods longer than one pairwise block and a buffer
    shorter than a bit period (no bits)."""
    g = golden["afsk"]
    for tag in g["tags"]:
        bits = O.afsk_bits(g[f"x_{tag}"], float(g[f"fs_{tag}"]), g[f"sos1200_{tag}"], g[f"sos2200_{tag}"])
        assert np.array_equal(bits, g[f"bits_{tag}"]), tag


def test_classify_signal(golden):
    """classify_signal with `welch` bound: label and bandwidth equal, modulation index bit-exact (float32,
    restated operation by operation), Welch PSD / flatness within the reference's own float32-FFT noise."""
    g = golden["classify"]
    fs = float(g["fs"])
    assert np.array_equal(O.hann1024(), g["win"])
    for tag in g["tags"]:
        lab, bw, mi, fl, psd = O.classify(g[f"iq_{tag}"], fs)
        assert lab == str(g[f"label_{tag}"]), tag
        assert bw == float(g[f"bw_{tag}"]), tag
        rmi = g[f"mi_{tag}"]
        assert mi.tobytes() == rmi.tobytes() or (np.isnan(mi) and np.isnan(rmi)), (tag, mi, rmi)
        ref = g[f"psd_{tag}"]
        assert np.all(np.abs(psd - ref) <= 1e-5 * (ref + 1e-10) + 1e-6 * np.sqrt(ref * np.max(ref))), tag   # float32-FFT noise of the reference: ~1e-7 sqrt(p P_peak)
        rfl = float(g[f"flat_{tag}"])
        assert float(fl) == rfl or abs(float(fl) - rfl) <= 1e-5 * abs(rfl) or (np.isnan(fl) and np.isnan(rfl)), (tag, fl, rfl)
    with pytest.raises(ValueError):
        O.classify(np.zeros(0, np.complex64), fs)


def test_classify_signal_short_reads(golden):
    """Reads shorter than Welch's 1024-sample segment (signal_processing.py:299): SciPy takes nperseg = len(x) — one segment,
    a Hann window and an FFT of that length.  Same bars as above; n = 1 (PSD exactly 0, flatness inf, index NaN) included."""
    g = golden["classify_short"]
    for n in (1000, 257, 8, 3, 1):
        assert np.array_equal(O.hann(n), g[f"win_{n}"]), n
    for tag in g["tags"]:
        fs = float(g[f"fs_{tag}"])
        lab, bw, mi, fl, psd = O.classify(g[f"iq_{tag}"], fs)
        assert lab == str(g[f"label_{tag}"]), tag
        assert bw == float(g[f"bw_{tag}"]), tag
        rmi = g[f"mi_{tag}"]
        assert mi.tobytes() == rmi.tobytes() or (np.isnan(mi) and np.isnan(rmi)), (tag, mi, rmi)
        ref = g[f"psd_{tag}"]
        assert psd.shape == ref.shape
        assert np.all(np.abs(psd - ref) <= 1e-5 * (ref + 1e-10) + 1e-6 * np.sqrt(ref * np.max(ref))), tag
        rfl = float(g[f"flat_{tag}"])
        assert float(fl) == rfl or abs(float(fl) - rfl) <= 1e-5 * abs(rfl) or (np.isnan(fl) and np.isnan(rfl)), (tag, fl, rfl)


def test_morse_edges(golden):
    """decode_morse's mask / transition indices (decoders.py:149-161, threshold -20 dB: NumPy's SVML log10f pinned at the one point
    that matters) on the goldens."""
    g = golden["decoders"]
    for tag in g["mtags"]:
        rise, fall = O.morse_edges(g[f"m_iq_{tag}"])
        assert np.array_equal(rise, g[f"m_rise_{tag}"]) and np.array_equal(fall, g[f"m_fall_{tag}"]), tag
    # values one ulp either side of the cut-off: 0x3dccccce is still "off" under SVML (a correctly rounded log10f says "on")
    vals = (np.arange(-40, 41, dtype=np.int64) + 0x3DCCCCCD).astype(np.uint32).view(np.float32)
    x = np.zeros(2 * len(vals) + 1, np.complex64); x[0] = 1.0; x[1::2] = vals
    rise, fall = O.morse_edges(x)
    on = vals.view(np.uint32) >= 0x3DCCCCCF
    assert np.array_equal(rise, 2 * np.nonzero(on)[0]) and len(fall) == on.sum() + 1


def test_lengths_that_are_not_a_power_of_two(golden):
    """compute_fft and the sweep driver's per-read arithmetic (pyspecsdr.py:1049-1057) on frame lengths NumPy's fft accepts and
    a radix-2 transform does not (1000, 3001, 24 000 = int(0.1 * 240 kS/s), 17): Bluestein in the oracle."""
    g = golden["spectrum"]
    for n in g["np2_sizes"]:
        for iq, ref in zip(g[f"iq_np2_{n}"], g[f"db_np2_{n}"]):
            db = O.compute_fft(iq)
            assert np.all(np.abs(db - ref) <= 1e-9 * np.maximum(np.abs(ref), 1.0)), n
    g = golden["scanner"]
    for n in g["sw_sizes"]:
        fs, thr = g[f"sw_args_{n}"]
        for k, iq in enumerate(g[f"sw_iq_{n}"]):
            db, pk, bw, cnt = O.scan_threshold(iq, float(fs), float(thr))
            ref = g[f"sw_db_{n}"][k]
            assert np.array_equal(db.view(np.uint32), ref.view(np.uint32)), (n, k, int((db.view(np.uint32) != ref.view(np.uint32)).sum()))
            assert np.float32(pk).tobytes() == np.float32(g[f"sw_peak_{n}"][k]).tobytes()
            assert cnt == int(g[f"sw_count_{n}"][k]) and bw == g[f"sw_bw_{n}"][k]


def test_reference_is_not_bit_stable_across_cpu_dispatch(golden):
    """"bit-exact" is relative to NumPy's AVX512_SKX dispatch (what the main goldens are stamped with and
    what the oracle / kernels replay).  tests/golden/nfm_dispatch.npz holds the REFERENCE's own NFM outputs for the same
    inputs under NPY_DISABLE_CPU_FEATURES (tools/make_goldens_dispatch.py): without AVX-512 (glibc atan2f instead of
    SVML) and without AVX2 / FMA3 as well (no fused complex multiply).  The float32 discriminator changes in a large
    share of the samples, the int16 audio in a few — by one LSB — so the oracle, pinned to the SKX goldens bit for bit
    (test_nfm_*), cannot and does not match the other dispatches exactly."""
    g, d = golden["nfm"], golden["nfm_dispatch"]
    expect = {"avx2_fma3": (0.25, 0.35), "baseline": (0.70, 0.85)}     # share of discriminator values that differ
    for name in [str(v) for v in d["variants"]]:
        tot = diff = dd = dt = 0
        for tag in ("a", "b", "c", "e", "f"):
            a, b = g["pcm_" + tag], d[f"{name}_pcm_{tag}"]
            assert a.shape == b.shape
            assert np.abs(a.astype(int) - b.astype(int)).max() <= 1          # never more than one LSB
            tot += a.size // 2
            diff += int((a[..., 0] != b[..., 0]).sum())
            x, y = g["disc_" + tag], d[f"{name}_disc_{tag}"]
            dd += int((x.view(np.uint32) != y.view(np.uint32)).sum())
            dt += x.size
        lo, hi = expect[name]
        assert lo < dd / dt < hi, (name, dd / dt)
        assert 0 < diff <= 0.01 * tot, (name, diff, tot)                      # a handful of int16 samples, not zero
    # the oracle replays the SKX dispatch: its discriminator equals the main goldens and therefore differs from these
    taps, sos, zi = g["taps_a"], g["sos_a"], g["zi_a"]
    _, disc, _ = O.demod_nfm(g["iq_a"][0], float(g["fs_a"]), taps, sos, zi, stages=True)
    assert np.array_equal(disc.view(np.uint32), g["disc_a"].view(np.uint32))
    assert not np.array_equal(disc.view(np.uint32), d["avx2_fma3_disc_a"].view(np.uint32))


def test_headline_f64_from_iq_draws_the_reference_cells(golden):
    """oracle_lib.headline_f64 — the checker bench.py, smoke() and the -m gpu tests hold the device's display lines against: the oracle's own
    step from IQ in the reference's row type (float64 compute_fft rows -> np.convolve / np.median / clamp -> draw_waterfall) must draw the cells
    the REFERENCE drew from the same read buffers (tests/golden/caller_iq.npz -> caller.npz, draw_waterfall through a fake screen), every one."""
    g, q, nfm = golden["caller"], golden["caller_iq"], golden["nfm"]
    iq = np.ascontiguousarray(q["iq"])
    H, W = [int(v) for v in g["hw"]]
    tag = [t for t in "abcdefg" if float(nfm[f"fs_{t}"]) == 2.4e6][0]
    o = O.headline_f64(iq, 2.4e6, nfm[f"taps_{tag}"], nfm[f"sos_{tag}"], nfm[f"zi_{tag}"], 30, W - 8, 2, keep_db=True)
    assert np.allclose(o["db"], q["db"], rtol=1e-9, atol=1e-9)
    assert np.allclose(o["post"], g["rows"], rtol=1e-9, atol=1e-9)
    for i in range(len(iq)):
        assert np.array_equal(o["glyph"][i], g["wf_glyph"][i][0]) and np.array_equal(o["colour"][i], g["wf_colour"][i][0]), i
    # the PCM of the same call is the golden NFM path's (frame by frame)
    for i in (0, 7, len(iq) - 1):
        assert np.array_equal(o["pcm"][i], O.pcm16_stereo(O.demod_nfm(iq[i], 2.4e6, nfm[f"taps_{tag}"], nfm[f"sos_{tag}"], nfm[f"zi_{tag}"])))


def test_round5_arguments_oracle_vs_reference(golden):
    """tests/golden/args.npz (tools/make_goldens_round5.py): the reference's demodulate_nfm / _wfm at target rates other than 22050
    (signal_processing.py:111: int(sample_rate / target_rate)) and decode_morse's mask at thresholds other than -20 dB (decoders.py:156) —
    the oracle's restatement bit for bit."""
    g = golden["args"]
    for t in g["rate_tags"]:
        fs, tr = float(g[f"fs_{t}"]), float(g[f"tr_{t}"])
        for k, x in enumerate(g[f"iq_{t}"]):
            if str(t).startswith("n"):
                a = O.demod_nfm(x, fs, g[f"taps_{t}"], g[f"sos_{t}"], g[f"zi_{t}"], target_rate=tr)
                assert np.array_equal(a, g[f"audio_{t}"][k][:, 0]), t
                assert np.array_equal(O.pcm16_stereo(a), g[f"pcm_{t}"][k]), t
            else:
                filt = dict(lp_sos=g[f"lp_{t}"], pilot_sos=g[f"pil_{t}"], lmr_sos=g[f"lmr_{t}"], alpha=float(g[f"alpha_{t}"]),
                            dec_sos=g[f"sos_{t}"], dec_zi=g[f"zi_{t}"])
                assert np.array_equal(O.demod_wfm(x, fs, filt, target_rate=tr), g[f"audio_{t}"][k]), t
    for t in g["morse_tags"]:
        r, f = O.morse_edges(g[f"m_iq_{t}"], float(g[f"m_thr_{t}"]))
        assert np.array_equal(r, g[f"m_rise_{t}"]) and np.array_equal(f, g[f"m_fall_{t}"]), t


def test_round6_complex128_buffers_oracle_vs_reference(golden):
    """tests/golden/c128.npz (tools/make_goldens_round6.py: the reference handed complex128 read buffers — it then computes compute_fft and
    demodulate_am in float64 from the first statement): the oracle's float64 restatements — np.abs(complex128) as the scaled hypot in float64,
    np.mean as the float64 pairwise tree — give the reference's audio and int16 PCM bit for bit at every length (1024 ... 32 768, 1000, 37) and
    its dB rows to 1e-9; the inputs are not representable in complex64 (narrowing them changes the results, which is what rounds 1-5 did)."""
    g = golden["c128"]
    sos = g["am_sos"]
    for t in g["tags"]:
        for k, x in enumerate(g[f"iq_{t}"]):
            assert x.dtype == np.complex128 and not np.array_equal(x, x.astype(np.complex64).astype(np.complex128))
            assert np.array_equal(np.array([O.lib().pss_o_cabs(float(v.real), float(v.imag)) for v in x[:64]]), g[f"abs_{t}"][:64]) or k > 0
            db = O.compute_fft_c128(x)
            assert np.allclose(db, g[f"db_{t}"][k], rtol=1e-9, atol=1e-9), (t, k)
            au = O.demod_am_c128(x, sos)
            assert np.array_equal(au, g[f"audio_{t}"][k]), (t, k)
            assert np.array_equal(np.int16(au * 32767), g[f"pcm_{t}"][k]), (t, k)
            # narrowing to complex64 first (rounds 1-5) does NOT give these bits
            if len(x) >= 1000:
                assert not np.array_equal(O.demod_am(x.astype(np.complex64), sos), g[f"audio_{t}"][k])
            # demodulate_ssb (:198-217): the complex128 convolution on the samples as they are, hilbert() round trip (pocketfft restatement) at the
            # power-of-two lengths: bit for bit there, int16 equal everywhere (the round trip is skipped at other lengths: ~1e-16)
            a = O.demod_ssb_c128(x, g["ssb_taps"])
            if (len(x) & (len(x) - 1)) == 0:
                assert np.array_equal(a, g[f"ssb_{t}"][k]), (t, k)
            assert np.allclose(a, g[f"ssb_{t}"][k], rtol=0, atol=2e-15) and np.array_equal(np.int16(a * 32767), g[f"ssbpcm_{t}"][k]), (t, k)
            if len(x) >= 1000:
                assert not np.array_equal(O.demod_ssb(x.astype(np.complex64), g["ssb_taps"]), g[f"ssb_{t}"][k])      # narrowing first changes the bits
            # measure_signal_power (:325-328): the array part np.mean(np.abs(x) ** 2) restated in float64, the scalar log10 with NumPy's own
            p = O.mean_power_c128(x)
            assert p == g[f"mp_{t}"][k], (t, k)
            assert 10 * np.log10(p + 1e-10) == g[f"pw_{t}"][k], (t, k)
    # demodulate_nfm at a decimation factor of one (target_rate above half the sample rate)
    for t in g["q1_tags"]:
        fs, tr = float(g[f"n_fs_{t}"]), float(g[f"n_tr_{t}"])
        for k, x in enumerate(g[f"n_iq_{t}"]):
            a = O.demod_nfm(x, fs, g[f"n_taps_{t}"], g[f"n_sos_{t}"], g[f"n_zi_{t}"], target_rate=tr)
            assert a.shape == (len(x) - 1,) and np.array_equal(a, g[f"n_audio_{t}"][k]), (t, k)
    # demodulate_wfm at a decimation factor of one: no decimate() stage at all (:152-155)
    import scipy.signal as ss
    dec = ss.cheby1(8, 0.05, 0.8, output="sos")      # (unused at factor 1; the oracle's argument list wants a table)
    for t in g["wq1_tags"]:
        fs, tr = float(g[f"w_fs_{t}"]), float(g[f"w_tr_{t}"])
        filt = dict(lp_sos=g[f"w_lp_{t}"], pilot_sos=g[f"w_pil_{t}"], lmr_sos=g[f"w_lmr_{t}"], alpha=float(g[f"w_alpha_{t}"]), dec_sos=dec, dec_zi=ss.sosfilt_zi(dec))
        for k, x in enumerate(g[f"w_iq_{t}"]):
            a = O.demod_wfm(x, fs, filt, target_rate=tr)
            assert a.shape == (len(x) - 1, 2) and np.array_equal(a, g[f"w_audio_{t}"][k]), (t, k)
    with np.errstate(all="ignore"):
        z = g["iq_z"][0]
        assert np.array_equal(O.compute_fft_c128(z), g["db_z"][0])                 # every bin exactly -100 dB
        assert np.all(np.isnan(O.demod_am_c128(z, sos))) and np.all(np.isnan(g["audio_z"][0]))   # silence -> 0 / 0
