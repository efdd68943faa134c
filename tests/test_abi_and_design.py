"""CPU-side checks: the C ABI library loads and exports every symbol include/pss.h declares; the host-side
filter designer against SciPy's coefficients (golden); error behaviour without a GPU."""
import ctypes as C
import os
import re

import math

import numpy as np
import pytest

from pyspecsdr_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pss.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pss_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pss.h but not exported by libpss.so"
    assert set(L.exported_symbols()) == set(names), "ctypes table and header drifted apart"


def test_no_gpu_is_a_loud_error_not_a_fallback():
    lib = L.load()
    if lib.pss_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.pss_create(0, C.byref(h)) == L.PSS_E_HIP
    assert b"no HIP device" in lib.pss_last_error(None)
    from pyspecsdr_amd.engine import Engine, PssError
    with pytest.raises(PssError):
        Engine(0)


def test_out_len_and_arg_checks():
    lib = L.load()
    assert lib.pss_demod_out_len(L.MODE_NFM, 1024, 2.4e6) == 10       # ceil(1023/108)
    assert lib.pss_demod_out_len(L.MODE_NFM, 2048, 10e6) == 5         # q = 453
    assert lib.pss_demod_out_len(L.MODE_NFM, 29, 2.4e6) == 1
    assert lib.pss_demod_out_len(L.MODE_AM, 16384, 2.4e6) == 16384
    assert lib.pss_demod_out_len(L.MODE_USB, 300, 1e6) == 300
    assert lib.pss_demod_out_len(L.MODE_NFM, 1024, 20000.0) < 0       # int(fs/22050) == 0
    assert lib.pss_demod_out_len(7, 1024, 2.4e6) < 0
    taps = np.empty(65)
    assert lib.pss_design_firwin(65, 1.2, taps.ctypes.data) == L.PSS_E_CUTOFF   # scipy firwin ValueError
    assert lib.pss_design_firwin(65, 0.0, taps.ctypes.data) == L.PSS_E_CUTOFF


def ulps(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.spacing(np.abs(b)), 1e-300))


@pytest.mark.parametrize("fs", [1.024e6, 2.4e6, 10e6, 2.048e6, 250e3])
def test_designer_matches_scipy(golden, fs):
    g = golden["nfm"]
    lib = L.load()
    key, q = str(int(fs)), int(fs / 22050)
    taps, sos, zi = np.empty(65), np.empty((4, 6)), np.empty((4, 2))
    assert lib.pss_design_firwin(65, 15000 / (fs / 2), taps.ctypes.data) == 0
    assert lib.pss_design_cheby1_sos(8, 0.05, 0.8 / q, sos.ctypes.data) == 0
    assert lib.pss_design_sosfilt_zi(sos.ctypes.data, 4, zi.ctypes.data) == 0
    # bit for bit since round 3: every operation in SciPy's / NumPy's order (Smith's complex division, unfused complex products, pairwise
    # sums, glibc's csinh; NumPy's SVML tan / arcsinh agree with libm on every argument this chain produces)
    assert np.array_equal(taps, g["design_taps_" + key])
    assert np.array_equal(sos, g["design_sos_" + key])
    assert np.array_equal(zi, g["design_zi_" + key])
    # given SciPy's own sos, sosfilt_zi is reproduced exactly (LAPACK dgesv arithmetic restated)
    rs = np.ascontiguousarray(g["design_sos_" + key])
    zi2 = np.empty((4, 2))
    lib.pss_design_sosfilt_zi(rs.ctypes.data, 4, zi2.ctypes.data)
    assert np.array_equal(zi2, g["design_zi_" + key])


@pytest.mark.parametrize("fs", [1.024e6, 2.4e6, 10e6, 2.048e6, 250e3])
def test_butter_designer_matches_scipy(golden, fs):
    """pss_design_butter_sos vs scipy.signal.butter(5, ..., output='sos') for the three WFM filters (:126-133): same
    section order / zero pairing, coefficients within a few ulp."""
    import ctypes as C
    g = golden["wfm"]
    lib = L.load()
    key, nyq = str(int(fs)), fs / 2
    for name, lo, hi, ns in (("lp_sos", 0.0, 15000.0, 3), ("pilot_sos", 18800.0, 19200.0, 5), ("lmr_sos", 23000.0, 53000.0, 5)):
        sos, n = np.zeros((ns, 6)), C.c_int()
        assert lib.pss_design_butter_sos(5, lo / nyq, hi / nyq, sos.ctypes.data, C.addressof(n)) == 0
        ref = g[f"{name}_{key}"]
        assert n.value == ns == ref.shape[0]
        exact = (ref == 0) | (np.abs(ref) == 1) | (np.abs(ref) == 2)       # structural entries (zeros at +-1, 0)
        assert np.array_equal(sos[exact], ref[exact])
        assert np.array_equal(sos, ref), name          # bit for bit (the pre-warp tan is NumPy's SVML routine restated, not libm's)
    sos = np.zeros((5, 6))
    assert lib.pss_design_butter_sos(5, 300 / 11025, 3000 / 11025, sos.ctypes.data, None) == 0
    assert np.array_equal(sos, golden["am_ssb"]["am_sos"])
    assert lib.pss_design_butter_sos(5, 0.0, 1.06, sos.ctypes.data, None) == L.PSS_E_CUTOFF   # scipy: 0 < Wn < 1
    assert lib.pss_design_butter_sos(5, 0.5, 0.4, sos.ctypes.data, None) == L.PSS_E_CUTOFF


def test_am_table_is_scipy_butter(golden):
    sos = np.empty((5, 6))
    L.load().pss_am_bandpass_sos(sos.ctypes.data)
    assert np.array_equal(sos, golden["am_ssb"]["am_sos"])


def test_ssb_taps_design(golden):
    g = golden["am_ssb"]
    taps = np.empty(65)
    for tag in "abc":
        fs = float(g[f"ssb_fs_{tag}"])
        assert L.load().pss_design_firwin(65, 3000 / fs, taps.ctypes.data) == 0
        assert np.array_equal(taps, g[f"ssb_taps_{tag}"])


def _numpy_on_avx512_skx():
    """The live comparisons below restate NumPy's AVX512_SKX code paths (the reference environment of the goldens: SVML tan / exp, np.square's
    fused real part); on a host whose NumPy dispatches otherwise they would compare against a different NumPy."""
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as f
    except ImportError:
        return False
    return bool(f.get("AVX512_SKX"))


needs_skx = pytest.mark.skipif(not _numpy_on_avx512_skx(), reason="NumPy is not on its AVX512_SKX dispatch here (the goldens cover the tables)")


@needs_skx
def test_designers_equal_scipy_on_sweeps():
    """The library's own designers against SciPy itself (where the test runs): firwin for the NFM / SSB cutoffs at 120 sample rates,
    scipy.signal.decimate's cheby1(8, 0.05, 0.8 / q) sections and their sosfilt_zi for q = 2 .. 400, butter(5) low / band for the WFM
    filters at 120 sample rates — every coefficient bit."""
    import ctypes as C
    import scipy.signal as ss
    lib = L.load()
    taps, sos4, zi = np.empty(65), np.empty((4, 6)), np.empty((4, 2))
    rates = list(np.linspace(240e3, 20e6, 115)) + [250e3, 1.024e6, 2.048e6, 2.4e6, 10e6]
    for fs in rates:
        for c in (15000 / (fs / 2), 3000 / fs):
            assert lib.pss_design_firwin(65, c, taps.ctypes.data) == 0
            assert np.array_equal(taps, ss.firwin(65, c)), (fs, c)
    for q in range(2, 401):
        assert lib.pss_design_cheby1_sos(8, 0.05, 0.8 / q, sos4.ctypes.data) == 0
        ref = ss.cheby1(8, 0.05, 0.8 / q, output="sos")
        assert np.array_equal(sos4, ref), q
        assert lib.pss_design_sosfilt_zi(sos4.ctypes.data, 4, zi.ctypes.data) == 0
        assert np.array_equal(zi, ss.sosfilt_zi(ref)), q
    n_exact = 0
    for fs in rates:
        nyq = fs / 2
        for lo, hi, ns in ((0.0, 15000.0, 3), (18800.0, 19200.0, 5), (23000.0, 53000.0, 5)):
            if hi / nyq >= 1:
                continue
            args = [hi / nyq] if lo == 0 else [lo / nyq, hi / nyq]
            sos, n = np.zeros((ns, 6)), C.c_int()
            assert lib.pss_design_butter_sos(5, lo / nyq, hi / nyq, sos.ctypes.data, C.addressof(n)) == 0
            ref = ss.butter(5, args[0] if lo == 0 else args, btype="low" if lo == 0 else "band", output="sos")
            assert np.array_equal(sos, ref), (fs, lo, hi)
            n_exact += 1
    assert n_exact > 340


@needs_skx
def test_numpy_float64_tan_and_exp_models():
    """pss_h_np_f64: NumPy's float64 tan / exp (SVML's __svml_tan8_ha / __svml_exp8_ha under the AVX512_SKX dispatch, an ulp from libm on
    0.5 % / 5 % of arguments) restated in the library — every bit of np.tan / np.exp on 10^6 arguments each, incl. the pre-warp and
    de-emphasis arguments of 10^5 sample rates.  Only meaningful where NumPy runs that dispatch (the reference environment of the goldens)."""
    lib = L.load()
    rng = np.random.default_rng(4)
    fs = np.linspace(100e3, 60e6, 100000)
    sets = ((0, np.tan, [rng.uniform(-1000, 1000, 400000), rng.uniform(0, 1.5707, 400000), rng.uniform(0, 1, 100000) ** 4,
                         np.pi * (53000.0 / (fs / 2)) / 2.0]),
            (1, np.exp, [rng.uniform(-700, 700, 500000), -rng.uniform(0, 1, 400000) ** 3, -1 / (75e-6 * fs)]))
    for op, ref, xs in sets:
        for x in xs:
            x = np.ascontiguousarray(x)
            o = np.empty_like(x)
            assert lib.pss_h_np_f64(op, x.ctypes.data, len(x), o.ctypes.data) == 0
            want = ref(x)
            assert np.array_equal(o.view(np.uint64), want.view(np.uint64)), (op, int(np.sum(o != want)))


def test_formats_host_side(tmp_path):
    """The byte formats either side of the path (no GPU involved): np.save'd IQ recordings cut into read buffers
    (pyspecsdr.py:814-824, :2236), WAV header fields of audio_processing.py:25-43, FIFO bytes of io_manager.py:23-27."""
    import wave
    from pyspecsdr_amd import formats
    rec = (np.arange(2500) + 1j * np.arange(2500)[::-1]).astype(np.complex64)
    p = str(tmp_path / "rec.npy")
    np.save(p, rec)
    s = formats.load_iq_recording(p)
    assert s.dtype == np.complex64 and np.array_equal(s, rec)
    fr = formats.cut_frames(s, 1024)
    assert fr.shape == (2, 1024) and np.array_equal(fr[1], rec[1024:2048])        # incomplete tail buffer dropped
    with pytest.raises(ValueError):
        np.save(p, np.zeros((3, 4), np.float32)); formats.load_iq_recording(p)
    pcm = (np.arange(20, dtype=np.int16) - 10).reshape(10, 2)
    w = str(tmp_path / "a.wav")
    formats.write_wav(w, pcm)
    with wave.open(w, "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (2, 2, 22050, 10)
        assert f.readframes(10) == pcm.tobytes() == formats.pipe_bytes(pcm)


def test_zero_edit_dropin_resolves_the_reference_module_names(tmp_path):
    """pyspecsdr_amd.run.install(): `from signal_processing import *` (pyspecsdr.py:98; decoders.py:3) resolves to the drop-in
    module, with the reference's public names and nothing else leaking through the star import; `decoders` is NOT replaced
    (the reference's own module keeps its bookkeeping and picks up the GPU band-pass through its import)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    app = tmp_path / "app.py"
    (tmp_path / "decoders.py").write_text("from signal_processing import bandpass_filter\nWHO = bandpass_filter.__module__\n")
    app.write_text("from signal_processing import *\nimport sys, decoders\nimport signal_processing as m\n"
                   "assert m.__name__ == 'pyspecsdr_amd.signal_processing'\n"
                   "assert decoders.__name__ == 'decoders' and decoders.WHO == 'pyspecsdr_amd.signal_processing'\n"
                   "for f in (compute_fft, demodulate_signal, measure_signal_power, classify_signal, bandpass_filter, iq_correction):\n"
                   "    assert callable(f)\n"
                   "assert 'get_engine' not in globals() and 'Engine' not in globals() and 'L' not in globals()\n"
                   "import sys; print('ARGS', sys.argv[1:])\n")
    out = subprocess.run([sys.executable, "-m", "pyspecsdr_amd.run", str(app), "--demod", "WFM"], cwd=root, check=True,
                         capture_output=True, text=True, timeout=300).stdout
    assert "ARGS ['--demod', 'WFM']" in out


def test_bench_profile_digest_accounting():
    """bench.py quotes PMC-derived numbers only from a digest taken on THIS source tree (hash of pyspecsdr_amd/csrc); the VALU issue
    accounting weighs float64-class instructions at 4 clocks and 32-bit ones at 2 (tools/ubench/valu_rate.hip)."""
    import json
    import bench
    dig = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    k = dig["kernels"]["k_nfm_fwd"]
    f64 = sum(k[c] for c in ("sq_insts_valu_add_f64", "sq_insts_valu_mul_f64", "sq_insts_valu_fma_f64", "sq_insts_valu_cvt"))
    want = (f64 / bench.VALU_RATE_F64 + (k["valu_insts"] - f64) / bench.VALU_RATE_B32) * 1e3
    assert abs(bench.valu_issue_ms(k) - want) < 1e-9 and 0.2 < want < 0.6          # ~0.33 ms of pure issue per launch
    traffic, busy, note = bench.profiled("k_nfm_fwd", dig["n_frames"])
    sv = bench.step_valu(["k_nfm_fwd", "k_spectrum", "k_post", "k_disp_rows", "k_nfm_bwd"], dig["n_frames"], 1.1)
    if dig["src_hash"] == bench.source_hash():
        assert traffic == k["traffic_bytes"] and 0.3 < busy < 1.0 and 0.3 < sv["frac"] < 1.0
    else:                                                                            # stale digest: nothing is quoted from it
        assert traffic is None and busy is None and sv is None and "is from source" in note
    assert bench.profiled("k_nfm_fwd", dig["n_frames"] + 1)[1] is None               # another batch size: no busy fraction


def test_microbenchmarks_compile_for_gfx950():
    """The measurement programs DESIGN.md cites (tools/ubench/*.hip) still build (hipcc cross-compiles without a GPU)."""
    import glob
    import shutil
    import subprocess
    import tempfile
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        pytest.skip("hipcc not available")
    srcs = sorted(glob.glob(os.path.join(ROOT, "tools", "ubench", "*.hip")))
    assert srcs
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([cc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-w", "-c", s, "-o",
                                   os.path.join(d, os.path.basename(s) + ".o")]) for s in srcs]
        assert all(p.wait() == 0 for p in procs)


def test_no_unprotected_wide_buffer_store_hazard():
    """gfx950 samples the data registers of a > 64-bit buffer store late even when the store's scalar offset is an SGPR, a case LLVM's
    hazard recogniser skips: a VALU write of those registers in the next two issue slots corrupts the stored value (k_hilbert_xl, round 3:
    16-48 wrong samples on a cold launch).  tools/check_store_hazard.py scans the kernels' ISA for the pattern; the checker itself is
    exercised on a hand-written positive and negative."""
    import shutil
    import subprocess
    import sys
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_store_hazard as C
    with tempfile.TemporaryDirectory() as d:
        bad = os.path.join(d, "bad.s")
        open(bad, "w").write("_Zk:\n\tbuffer_store_dwordx4 v[14:17], v106, s[20:23], s86 offen\n\tv_and_b32_e32 v16, 0x7fffffff, v75\n\ts_endpgm\n")
        ok = os.path.join(d, "ok.s")
        open(ok, "w").write("_Zk:\n\tbuffer_store_dwordx4 v[14:17], v106, s[20:23], s86 offen\n\ts_nop 1\n\tv_and_b32_e32 v16, 0x7fffffff, v75\n"
                            "\tbuffer_store_dwordx4 v[14:17], v2, s[20:23], 0 offen\n\tv_mov_b32_e32 v14, 0\n"      # constant soffset: the compiler's job
                            "\tbuffer_store_dwordx2 v[14:15], v2, s[20:23], s3 offen\n\tv_mov_b32_e32 v14, 0\n\ts_endpgm\n")
        assert len(C.scan(bad)) == 1 and C.scan(ok) == []
    cc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(cc):
        pytest.skip("hipcc not available")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], check=True, capture_output=True, timeout=900)
    hits = [h for f in ("pss_fft.s", "pss_demod.s") for h in C.scan(os.path.join(ROOT, "pyspecsdr_amd", "_build", "asm", f))]
    assert hits == [], hits


def test_r16_exchange_layouts_are_bank_conflict_free():
    """The LDS paddings of pss_fft_r16.h (Cfg<LOG_R3>: E1_STRIDE, E2_STRIDE, TW2S) against the per-instruction lane groups of the
    MI355X LDS (MI355X_MICROARCH.md): ds_read_b128 is served in four scattered groups of 16 lanes on 64 banks (16 slots of 16 bytes),
    ds_write_b128 in eight groups of 8 consecutive lanes on 32 banks (8 slots).  Every access of the two exchanges and the stage-2
    twiddle reads must put a group's lanes on different slots (rocprofv3 SQ_LDS_BANK_CONFLICT went from 29-58 % of the LDS cycles to
    0 with these layouts, profiles/r03_lds_bank_conflicts.txt).  The formulas are read back from the header."""
    import re
    src = open(os.path.join(ROOT, "pyspecsdr_amd", "csrc", "pss_fft_r16.h")).read()
    assert "static constexpr int E1_STRIDE = T + (R3 == 1 ? 4 : R3 % 16);" in src
    assert "static constexpr int E2_STRIDE = 256 + (R3 == 1 ? 2 : R3 <= 8 ? 8 / R3 : 1);" in src
    assert re.search(r"static constexpr int TW2S = 17;", src)
    read_groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    read_groups += [[l + 32 for l in g] for g in read_groups]
    write_groups = [list(range(8 * g, 8 * g + 8)) for g in range(8)]

    def conflict_free(addr_of_lane, groups, slots):
        return all(len({addr_of_lane(l) % slots for l in g}) == len(g) for g in groups)

    for R3 in (2, 4, 8, 16):
        T = 16 * R3
        E1, E2, TW2S = T + R3 % 16, 256 + (8 // R3 if R3 <= 8 else 1), 17
        EX = max(16 * E1, R3 * E2)
        fpw = max(1, 256 // T)

        def lane(l, wave=0):           # lane l of wavefront `wave` of a 256-thread workgroup -> (frame slot, thread inside the frame)
            tid = 64 * wave + l
            return (tid // T) % fpw, tid % T
        for wave in range(4):
            base = lambda l: lane(l, wave)[0] * EX
            t_of = lambda l: lane(l, wave)[1]
            for k2 in range(16):       # exchange 1: writes row k2, column t; reads row t / R3, column t % R3 + R3 m2
                assert conflict_free(lambda l: base(l) + k2 * E1 + t_of(l), write_groups, 8), (R3, "e1 write")
            for m2 in range(16):
                assert conflict_free(lambda l: base(l) + (t_of(l) // R3) * E1 + t_of(l) % R3 + R3 * m2, read_groups, 16), (R3, "e1 read")
            for j2 in range(16):       # exchange 2: writes plane t % R3, column 16 j2 + t / R3; twiddle row t % R3
                assert conflict_free(lambda l: base(l) + (t_of(l) % R3) * E2 + 16 * j2 + t_of(l) // R3, write_groups, 8), (R3, "e2 write")
                # (the twiddle table is shared by the workgroup's frames: identical addresses broadcast, so count distinct ADDRESSES per slot)
                for g in read_groups:
                    per_slot = {}
                    for l in g:
                        a = (t_of(l) % R3) * TW2S + j2
                        per_slot.setdefault(a % 16, set()).add(a)
                    assert all(len(v) == 1 for v in per_slot.values()), (R3, "twiddle read")
            for c in range(16 // R3):  # exchange 2: reads plane m1, column t + T c
                for m1 in range(R3):
                    assert conflict_free(lambda l: base(l) + m1 * E2 + t_of(l) + T * c, read_groups, 16), (R3, "e2 read")


def test_decoder_back_halves_vs_reference_goldens(golden):
    """pss_h_morse_decode / pss_h_ax25_frame (host code of the library: no GPU) against what the reference's decode_morse /
    decode_ax25_frame returned (tests/golden/decoders.npz: m_text_*, m_timing_*, ax_out; SURVEY §8(f) #5, decoders.py:6-88, :167-231).
    Morse: text and (dot, dash, mean gap) equal on every bit for the keyed signals and silence — the tags for which the reference's kmeans
    gives the same centroids whatever its random draw (probed over 30 seeds in the build container) — and for the seeded 'noisy' golden;
    for pure noise ('noise') the reference's own answer changes with the seed (its kmeans stops on a 1e-5 s threshold before converging on
    sub-millisecond glitches): there the result must be a valid decode and reproducible.  AX.25: all 40 planted / random bit streams."""
    import json
    from pyspecsdr_amd import decoders as D
    g = golden["decoders"]
    for tag in g["mtags"]:
        text, tm = D.morse_from_edges(g[f"m_rise_{tag}"], g[f"m_fall_{tag}"], float(g[f"m_fs_{tag}"]))
        want_t, want = str(g[f"m_text_{tag}"]), g[f"m_timing_{tag}"]
        got = np.array([float(tm["dot"]), float(tm["dash"]), float(tm["gap"])])
        if tag == "noise":
            text2, tm2 = D.morse_from_edges(g[f"m_rise_{tag}"], g[f"m_fall_{tag}"], float(g[f"m_fs_{tag}"]))
            assert text == text2 and tm == tm2 and 0 < got[0] < got[1] and len(text) >= 1
            continue
        assert text == want_t, (tag, text[:40], want_t[:40])
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (tag, got, want)
    # edge lists the reference returns early on: no rise, no fall, a lone fall before a lone rise
    for rise, fall in (([], []), ([5], []), ([], [5]), ([9], [3])):
        assert D.morse_from_edges(rise, fall, 24000.0) == ("", {"dot": 0, "dash": 0, "gap": 0})
    # one pulse: dash = 3 dots, no gap
    t1, m1 = D.morse_from_edges([100], [1300], 24000.0)
    assert t1 == "E" and m1["dot"] == 1200 / 24000.0 and m1["dash"] == m1["dot"] * 3 and m1["gap"] == 0
    # equal pulses: one class, dot == dash, every pulse a dash
    t2, m2 = D.morse_from_edges([0, 2000, 4000], [1000, 3000, 5000], 24000.0)
    assert m2["dot"] == m2["dash"] and set(t2) <= set("O?T M")
    outs = json.loads(str(g["ax_out"]))
    off = 0
    for k, ln in enumerate(g["ax_len"]):
        bits = g["ax_bits"][off:off + int(ln)]
        off += int(ln)
        r = D.decode_ax25_frame([int(b) for b in bits])
        assert ("<None>" if r is None else r) == outs[k], (k, r, outs[k])
    assert D.decode_ax25_frame([]) is None and D.decode_ax25_frame([0, 1, 1, 1, 1, 1, 1, 0]) is None
    assert D.decode_aprs_payload(list(range(13))) is None
    assert D.decode_aprs_payload([ord(c) << 1 for c in "APRS  "] + [0x60] + [ord(c) << 1 for c in "N0CALL"] + [0x61, 3, 0xF0] + [72, 105]) == "N0CALL>APRS:\u00f0Hi"   # (the reference's info field starts at byte 15: the PID rides along)
