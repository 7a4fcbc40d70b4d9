"""Pins oracle/spectralgate_oracle.py against the golden vectors generated from the
live reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import spectralgate_oracle as O
from tests.golden.cases import S_CASES, T_CASES, make_input_S, make_input_T, sha

TOL_S = 1e-11   # oracle and reference are both float64; differences are rounding only
TOL_T = 2e-6    # the reference builds its Hann window / OLA envelope in float32


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", sorted(S_CASES))
def test_reduce_noise_S_matches_reference(golden_dir, name):
    case = S_CASES[name]
    g = _load(golden_dir, "S_" + name)
    y, y_noise = make_input_S(case)
    assert sha(y) == str(g["in_sha"]), "seeded input drifted from the one the golden was made with"
    out = O.reduce_noise_S(y, case["sr"], y_noise=y_noise, **case["kwargs"])
    assert out.shape == g["out"].shape and out.dtype == g["out"].dtype
    assert O.rel_err(out, g["out"]) < TOL_S


def test_stage_taps_match_reference(golden_dir):
    case = S_CASES["stat_1chunk"]
    g = _load(golden_dir, "S_stat_1chunk")
    y, _ = make_input_S(case)
    thresh, _, _ = O.noise_threshold_S(y[None, :], 1024, 1024, 256, 1.5, 600000)
    assert np.max(np.abs(thresh - g["thresh"])) < 1e-10
    chunk = O.read_chunk(y[None, :], -30000, len(y) + 30000)
    nf, nt, smooth = O.mask_smoothing_widths(48000, 1024, 256, 500, 50)
    assert (nf, nt, smooth) == (5, 9, True)
    filt = O.smoothing_filter(nf, nt)
    _, stages = O.gate_stationary_S(chunk, g["thresh"], 1024, 1024, 256, 1.0, filt,
                                    return_stages=True)
    raw = stages[0]["raw"]
    assert tuple(g["raw_shape"]) == raw.shape
    assert np.array_equal(np.packbits(raw, axis=1), g["raw_bits"])       # bit-exact mask
    assert np.max(np.abs(stages[0]["mask"][[0, 3, 100, 511, 512], :] - g["smooth_rows"])) < 1e-12
    cols = [0, 1, 2, 117, 200, raw.shape[1] - 1]
    assert np.max(np.abs(stages[0]["Z"][:, cols] - g["Z_cols"])) < 1e-15


def test_fish_wav_config0(golden_dir):
    """BASELINE.json configs[0]: assets/fish.wav, mono, stationary, reference CPU path."""
    g = _load(golden_dir, "S_fish")
    data, rate = g["data"], int(g["rate"])
    out = O.reduce_noise_S(data, rate, stationary=True)
    assert out.dtype == np.int16
    assert np.array_equal(out, g["out_i16"])                              # int16: bit-exact
    out64 = O.reduce_noise_S(data.astype(np.float64), rate, stationary=True)
    assert O.rel_err(out64, g["out_f64"]) < 1e-6                          # golden stored as f32
    out_ns = O.reduce_noise_S(data.astype(np.float64), rate, stationary=False)
    assert O.rel_err(out_ns, g["out_ns"]) < 1e-6


@pytest.mark.parametrize("name", sorted(T_CASES))
def test_torchgate_T_matches_reference(golden_dir, name):
    case = T_CASES[name]
    g = _load(golden_dir, "T_" + name)
    x, xn = make_input_T(case)
    assert sha(x) == str(g["in_sha"])
    out = O.torchgate_T(x, case["sr"], xn=xn, window=g["window"], **case["kwargs"])
    assert out.shape == g["out"].shape
    assert O.rel_err(out, g["out"]) < TOL_T


@pytest.mark.parametrize("name", sorted(T_CASES))
def test_torch_cpu_port_matches_reference(golden_dir, name):
    """oracle/torchgate_torch_port.py (the torch-on-CPU restatement timed by bench.py's cpu_baseline leg for
    configs[4]) against the same golden vectors of the live reference."""
    import torch
    from oracle.torchgate_torch_port import torchgate_cpu
    case = T_CASES[name]
    g = _load(golden_dir, "T_" + name)
    x, xn = make_input_T(case)
    out = torchgate_cpu(torch.from_numpy(x), case["sr"], xn=None if xn is None else torch.from_numpy(xn),
                        **case["kwargs"]).numpy()
    assert out.shape == g["out"].shape
    assert O.rel_err(out, g["out"]) < TOL_T


def test_conv_variants_agree():
    rng = np.random.default_rng(0)
    m = (rng.random((40, 60)) > 0.5) * 1.0
    K = O.smoothing_filter(5, 9)
    assert np.max(np.abs(O.conv2_same(m, K) - O.conv2_same_direct(m, K))) < 1e-13


def test_smoothing_filter_shape_and_sum():
    K = O.smoothing_filter(5, 9)
    assert K.shape == (11, 19) and abs(K.sum() - 1) < 1e-15
    assert np.allclose(O.triangle(3), np.array([1, 2, 3, 4, 3, 2, 1]) / 4)


# ---- building blocks against the third-party primitives the reference calls ----
@pytest.mark.parametrize("n_fft,W,H", [(1024, 1024, 256), (512, 400, 100), (1000, 1000, 250), (777, 600, 151)])
def test_stft_istft_blocks_match_scipy(n_fft, W, H):
    """stft_scipy / istft_scipy restate scipy.signal.stft / istft exactly as the reference calls them
    (stationary.py:87-93,120-125)."""
    import scipy.signal
    x = O.synth_signal(9000, seed=n_fft).astype(np.float64)
    _, _, Z = scipy.signal.stft(x, nfft=n_fft, noverlap=W - H, nperseg=W, padded=False)
    Zo = O.stft_scipy(x, n_fft, W, H)
    assert Zo.shape == Z.shape and np.max(np.abs(Zo - Z)) < 1e-13 * max(1.0, np.max(np.abs(Z)))
    _, y = scipy.signal.istft(Z, nfft=n_fft, noverlap=W - H, nperseg=W)
    yo = O.istft_scipy(Z, n_fft, W, H)
    assert yo.shape == y.shape and np.max(np.abs(yo - y)) < 1e-13


@pytest.mark.parametrize("n_fft,W,H", [(1024, 1024, 256), (512, 400, 100), (601, 601, 150)])
def test_stft_istft_blocks_match_torch(n_fft, W, H):
    """stft_torch / istft_torch restate torch.stft / torch.istft(center=True) (torchgate.py:223-262),
    including the frame count and output length for an odd n_fft."""
    import torch
    for L in (3 * n_fft + 17, 40 * H):                    # generic length, exact multiple of the hop
        x = np.random.default_rng(L).standard_normal((2, L))
        w = torch.hann_window(W, dtype=torch.float64)
        Z = torch.stft(torch.from_numpy(x), n_fft, H, W, window=w, center=True, pad_mode="constant",
                       return_complex=True)
        Zo = O.stft_torch(x, n_fft, W, H, window=w.numpy())
        assert Zo.shape == tuple(Z.shape) and np.max(np.abs(Zo - Z.numpy())) < 1e-10
        y = torch.istft(Z, n_fft, H, W, window=w, center=True).numpy()
        yo = O.istft_torch(Z.numpy(), n_fft, W, H, window=w.numpy())
        assert yo.shape == y.shape and np.max(np.abs(yo - y)) < 1e-10


def test_filtfilt_and_boxcar_blocks_match_libraries():
    """One-pole forward-backward smoother == scipy.signal.filtfilt(padtype=None) (nonstationary.py:115);
    boxcar == conv1d(ones(k), padding="same") / k (torchgate.py:179-190)."""
    import scipy.signal
    import torch
    rng = np.random.default_rng(3)
    A = np.abs(rng.standard_normal((17, 300)))
    b = O.iir_coefficient(0.5, 48000, 256)
    ref = scipy.signal.filtfilt([b], [1, b - 1], A, axis=-1, padtype=None)
    assert np.max(np.abs(O.filtfilt_onepole(b, A) - ref)) < 1e-13
    for k in (3, 8, 20):
        X = torch.from_numpy(A)[None]                    # (1, F, T)
        ref = torch.nn.functional.conv1d(X.reshape(-1, 1, A.shape[1]), torch.ones(1, 1, k, dtype=torch.float64),
                                         padding="same").reshape(A.shape).numpy() / k
        assert np.max(np.abs(O.boxcar_same(A, k) - ref)) < 1e-13


# ---- non-finite samples: the oracle keeps the reference's NaN behaviour (golden vectors from the live reference) ----
from tests.golden.cases import (S_NAN_CASES, T_NAN_CASES, make_input_S_nan, make_input_T_nan,  # noqa: E402
                                nonfinite_agree)


@pytest.mark.parametrize("name", sorted(S_NAN_CASES))
def test_nan_sample_S_matches_reference(golden_dir, name):
    case = S_NAN_CASES[name]
    g = _load(golden_dir, "S_nan_" + name)
    y, y_noise = make_input_S_nan(case)
    assert sha(y) == str(g["in_sha"])
    with np.errstate(all="ignore"):
        out = O.reduce_noise_S(y, case["sr"], y_noise=y_noise, **case["kwargs"])
    assert nonfinite_agree(out, g["out"], 1e-9) is None, nonfinite_agree(out, g["out"], 1e-9)


@pytest.mark.parametrize("name", sorted(T_NAN_CASES))
def test_nan_sample_T_matches_reference(golden_dir, name):
    case = T_NAN_CASES[name]
    g = _load(golden_dir, "T_nan_" + name)
    x, xn = make_input_T_nan(case)
    assert sha(x) == str(g["in_sha"])
    with np.errstate(all="ignore"):
        out = O.torchgate_T(x, case["sr"], xn=xn, window=g["window"], **case["kwargs"])
    assert nonfinite_agree(out, g["out"], TOL_T) is None, nonfinite_agree(out, g["out"], TOL_T)


# ---- an Inf sample: the oracle calls the reference's own FFT, so it reproduces the reference's Inf / NaN pattern ----
from tests.golden.cases import S_INF_CASES, make_input_S_inf  # noqa: E402


@pytest.mark.parametrize("name", sorted(S_INF_CASES))
def test_inf_sample_S_matches_reference(golden_dir, name):
    case = S_INF_CASES[name]
    g = _load(golden_dir, "S_inf_" + name)
    y, y_noise = make_input_S_inf(case)
    assert sha(y) == str(g["in_sha"])
    with np.errstate(all="ignore"):
        out = O.reduce_noise_S(y, case["sr"], y_noise=y_noise, **case["kwargs"])
    assert nonfinite_agree(out, g["out"], 1e-9) is None, nonfinite_agree(out, g["out"], 1e-9)
