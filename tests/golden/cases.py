"""Case table shared by make_golden.py (reference side, build container only) and
the parity tests (oracle / HIP side).  Inputs are regenerated from seeds."""
import hashlib

import numpy as np


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def synth(n, sr, seed, tone_hz=1000.0, tone_amp=0.5, noise_sigma=0.1):
    """float32-valued white noise + tone, returned as float64 copies (the dtype
    the reference is fed)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / sr
    y = noise_sigma * rng.standard_normal(n) + tone_amp * np.sin(2 * np.pi * tone_hz * t)
    return y.astype(np.float32).astype(np.float64)


# ---- variant S: reduce_noise(use_torch=False) -----------------------------
S_CASES = {
    # single chunk, defaults (per-stage taps stored)
    "stat_1chunk": dict(sr=48000, n=40000, seed=11, kwargs=dict(stationary=True), stages=True),
    "nonstat_1chunk": dict(sr=48000, n=40000, seed=12, kwargs=dict(stationary=False)),
    # chunk grid: 4 chunks, last one partial
    "stat_chunked": dict(sr=48000, n=90000, seed=13,
                         kwargs=dict(stationary=True, chunk_size=25000, padding=4000)),
    "nonstat_chunked": dict(sr=48000, n=90000, seed=14,
                            kwargs=dict(stationary=False, chunk_size=25000, padding=4000)),
    # 2 channels, explicit 2-D noise clip, partial reduction
    "stat_2ch_noise": dict(sr=48000, n=30000, seed=15, channels=2, noise_len=20000,
                           kwargs=dict(stationary=True, prop_decrease=0.8)),
    "nonstat_2ch_prop": dict(sr=48000, n=30000, seed=16, channels=2,
                             kwargs=dict(stationary=False, prop_decrease=0.6,
                                         thresh_n_mult_nonstationary=1.5,
                                         sigmoid_slope_nonstationary=7,
                                         time_constant_s=0.5)),
    # other FFT sizes / sample rates
    "stat_nfft512": dict(sr=16000, n=30000, seed=17, tone_hz=440.0,
                         kwargs=dict(stationary=True, n_fft=512)),
    "stat_nfft2048": dict(sr=44100, n=50000, seed=18,
                          kwargs=dict(stationary=True, n_fft=2048, n_std_thresh_stationary=2.0)),
    "nonstat_nfft256": dict(sr=8000, n=20000, seed=19, tone_hz=300.0,
                            kwargs=dict(stationary=False, n_fft=256)),
    # n_fft = 256 at 48 kHz (round 5: fast256.hpp -- four frames per register transform; the 50 ms smoothing is 37 frames here:
    # k_iir_mask<37>, tall integer-smoothing tiles), chunk grid with a partial last chunk
    "stat_nfft256_48k_chunks": dict(sr=48000, n=70000, seed=71,
                                    kwargs=dict(stationary=True, n_fft=256, chunk_size=30000, padding=3000)),
    "nonstat_nfft256_48k_chunks": dict(sr=48000, n=70000, seed=72,
                                       kwargs=dict(stationary=False, n_fft=256, chunk_size=30000, padding=3000)),
    # no smoothing / one-axis smoothing
    "stat_nosmooth": dict(sr=48000, n=30000, seed=20,
                          kwargs=dict(stationary=True, freq_mask_smooth_hz=None,
                                      time_mask_smooth_ms=None)),
    "stat_freqsmooth_only": dict(sr=48000, n=30000, seed=21,
                                 kwargs=dict(stationary=True, time_mask_smooth_ms=None)),
    # general STFT geometry (win_length < n_fft, hop not win/4)
    "stat_geom": dict(sr=48000, n=30000, seed=22,
                      kwargs=dict(stationary=True, n_fft=2048, win_length=1500, hop_length=300)),
    "nonstat_geom": dict(sr=48000, n=30000, seed=23,
                         kwargs=dict(stationary=False, n_fft=1024, win_length=800, hop_length=160)),
    # frame lengths that are not a power of two (even, odd, small) and the largest power of two
    "stat_nfft1000": dict(sr=48000, n=30000, seed=24, kwargs=dict(stationary=True, n_fft=1000)),
    "nonstat_nfft777": dict(sr=22050, n=25000, seed=25, tone_hz=700.0,
                            kwargs=dict(stationary=False, n_fft=777)),
    "stat_nfft100": dict(sr=8000, n=12000, seed=26, tone_hz=500.0,
                         kwargs=dict(stationary=True, n_fft=100, chunk_size=5000, padding=600)),
    "stat_nfft1536_geom": dict(sr=44100, n=30000, seed=27,
                               kwargs=dict(stationary=True, n_fft=1536, win_length=1200, hop_length=250,
                                           prop_decrease=0.7)),
    "stat_nfft8192": dict(sr=48000, n=70000, seed=28, kwargs=dict(stationary=True, n_fft=8192)),
    "nonstat_nfft8192": dict(sr=48000, n=70000, seed=29, kwargs=dict(stationary=False, n_fft=8192)),
    # long frames (round 3: four-step transform through HBM): powers of two > 8192, other lengths > 4096
    "stat_nfft16384": dict(sr=48000, n=100000, seed=61,
                           kwargs=dict(stationary=True, n_fft=16384, time_mask_smooth_ms=200)),
    "nonstat_nfft5000": dict(sr=48000, n=60000, seed=62, kwargs=dict(stationary=False, n_fft=5000)),
    "stat_nfft32768_chunks": dict(sr=48000, n=130000, seed=63,
                                  kwargs=dict(stationary=True, n_fft=32768, time_mask_smooth_ms=400, chunk_size=60000,
                                              padding=20000, prop_decrease=0.9)),
    "nonstat_nfft12000_geom": dict(sr=44100, n=80000, seed=64,
                                   kwargs=dict(stationary=False, n_fft=12000, win_length=10000, hop_length=2500,
                                               time_mask_smooth_ms=120)),
}


def make_input_S(case):
    n, sr = case["n"], case["sr"]
    C = case.get("channels", 1)
    chans = [synth(n, sr, case["seed"] + 100 * c, tone_hz=case.get("tone_hz", 1000.0) * (c + 1))
             for c in range(C)]
    y = chans[0] if C == 1 else np.stack(chans)
    y_noise = None
    if case.get("noise_len"):
        nl = case["noise_len"]
        y_noise = np.stack([0.1 * np.random.default_rng(case["seed"] + 7 + c).standard_normal(nl)
                            for c in range(C)]).astype(np.float32).astype(np.float64)
    return y, y_noise


# ---- variant T: TorchGate.forward (float64) ------------------------------------
T_CASES = {
    "stat": dict(sr=16000, B=3, L=9000, seed=31, kwargs=dict()),
    "stat_xn": dict(sr=16000, B=3, L=9000, seed=32, xn=(1, 5000), kwargs=dict()),
    "stat_xn_rows": dict(sr=16000, B=3, L=9000, seed=33, xn=(3, 6000), kwargs=dict(prop_decrease=0.5)),
    "nonstat": dict(sr=16000, B=3, L=9000, seed=34, kwargs=dict(nonstationary=True)),
    "nonstat_odd": dict(sr=16000, B=2, L=9001, seed=35,
                        kwargs=dict(nonstationary=True, n_movemean_nonstationary=7,
                                    n_thresh_nonstationary=1.0, temp_coeff_nonstationary=0.2,
                                    prop_decrease=0.9)),
    "stat_geom": dict(sr=16000, B=2, L=9000, seed=36,
                      kwargs=dict(n_fft=512, win_length=400, hop_length=100)),
    "stat_nosmooth": dict(sr=16000, B=2, L=6000, seed=37,
                          kwargs=dict(freq_mask_smooth_hz=None, time_mask_smooth_ms=None)),
    "stat_sr8k": dict(sr=8000, B=3, L=32000, seed=38, kwargs=dict(nonstationary=True)),
    # n_fft = 400 (25 ms at 16 kHz, the usual speech front-end frame) and an odd length
    "stat_nfft400": dict(sr=16000, B=3, L=8000, seed=39, kwargs=dict(n_fft=400)),
    "nonstat_nfft601": dict(sr=16000, B=2, L=8000, seed=40, kwargs=dict(nonstationary=True, n_fft=601)),
    # odd n_fft with a length that is a multiple of the hop: torch.stft pads one sample less than a frame,
    # so there is one frame fewer than 1 + L // hop
    "stat_nfft601_hopmult": dict(sr=16000, B=2, L=9000, seed=41, kwargs=dict(n_fft=601, hop_length=150)),
    # long frames
    "stat_nfft16384": dict(sr=48000, B=2, L=40000, seed=42, kwargs=dict(n_fft=16384, time_mask_smooth_ms=200)),
    "nonstat_nfft6000": dict(sr=48000, B=2, L=30000, seed=43,
                             kwargs=dict(nonstationary=True, n_fft=6000, n_movemean_nonstationary=5)),
}


def make_input_T(case):
    rng = np.random.default_rng(case["seed"])
    B, L, sr = case["B"], case["L"], case["sr"]
    t = np.arange(L, dtype=np.float64) / sr
    x = 0.1 * rng.standard_normal((B, L)) + 0.5 * np.sin(2 * np.pi * 440.0 * t)[None, :]
    x = x.astype(np.float32).astype(np.float64)
    xn = None
    if case.get("xn"):
        xn = (0.1 * rng.standard_normal(case["xn"])).astype(np.float32).astype(np.float64)
    return x, xn


# ---- non-finite samples: a NaN in the signal, in the noise clip, in one TorchGate row (separate tables: the
# comparisons have to be NaN-aware) ----------------------------------------------------------------------
S_NAN_CASES = {
    "signal_chunks": dict(sr=48000, n=90000, seed=51, nan_at=40000,
                          kwargs=dict(stationary=True, chunk_size=25000, padding=4000)),
    "noise_clip": dict(sr=48000, n=30000, seed=52, noise_len=20000, nan_in_noise=5000,
                       kwargs=dict(stationary=True)),
    "nonstat": dict(sr=48000, n=30000, seed=53, nan_at=12000, kwargs=dict(stationary=False)),
    "signal_nfft1000": dict(sr=48000, n=60000, seed=54, nan_at=33000,
                            kwargs=dict(stationary=True, n_fft=1000, chunk_size=20000, padding=2000)),
}
T_NAN_CASES = {
    "row": dict(sr=16000, B=3, L=9000, seed=55, nan_at=(1, 4000), kwargs=dict()),
    "row_xn": dict(sr=16000, B=3, L=9000, seed=56, xn=(1, 5000), nan_at=(2, 100), kwargs=dict()),
}


# ---- an Inf sample.  The reference's result depends on which bins of its FFT come out Inf (the
# band's chunk maximum is Inf: floor = Inf, the whole band PASSES in that chunk) and which NaN (np.max keeps the NaN:
# the band is gated) -- pocketfft's butterfly order decides.  The oracle calls the same FFT and reproduces it; the
# engine gates an Inf sample like a NaN (stated deviation): the goldens pin what is equal (the non-finite
# output samples, every chunk the Inf does not reach) and what is not (the finite rest of the affected chunk).
S_INF_CASES = {
    "signal_chunks": dict(sr=48000, n=90000, seed=51, inf_at=40000, value=np.inf,
                          kwargs=dict(stationary=True, chunk_size=25000, padding=4000)),
    "signal_chunks_neg": dict(sr=48000, n=90000, seed=57, inf_at=61000, value=-np.inf,
                              kwargs=dict(stationary=True, chunk_size=25000, padding=4000)),
}


def make_input_S_inf(case):
    y, y_noise = make_input_S(case)
    y = y.copy()
    y[..., case["inf_at"]] = case["value"]
    return y, y_noise


def make_input_S_nan(case):
    y, y_noise = make_input_S(case)
    if "nan_at" in case:
        y = y.copy()
        y[..., case["nan_at"]] = np.nan
    if "nan_in_noise" in case:
        y_noise = y_noise.copy()
        y_noise[..., case["nan_in_noise"]] = np.nan
    return y, y_noise


def make_input_T_nan(case):
    x, xn = make_input_T(case)
    x = x.copy()
    x[case["nan_at"]] = np.nan
    return x, xn


def nonfinite_agree(got, want, tol):
    """Same non-finite samples; the finite rest within tol of max(1e-3, peak).  Returns an error string or None."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    if got.shape != want.shape:
        return "shape %s vs %s" % (got.shape, want.shape)
    gn, wn = ~np.isfinite(got), ~np.isfinite(want)
    if not np.array_equal(gn, wn):
        return "non-finite samples differ: %d vs %d" % (gn.sum(), wn.sum())
    ok = ~gn
    if ok.any():
        err = np.abs(got[ok] - want[ok]).max() / max(1e-3, np.abs(want[ok]).max())
        if err > tol:
            return "finite rest differs by %.3e" % err
    return None
