"""NaN / Inf samples against the reference's behaviour (goldens S_nan_*, S_inf_*, T_nan_*).
(grouped by subject in round 5; the tests themselves date from rounds 2-4)"""
import os
import threading

import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4   # BASELINE.json north_star: output within 1e-4 (relative to peak) of the CPU reference

SG_KW = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
             clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None,
             time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
             use_tqdm=False, n_jobs=1)


@pytest.fixture(scope="module")
def nr():
    import noisereduce_amd
    return noisereduce_amd


def _sg(y, sr, cs, pad, **over):
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    kw = dict(SG_KW, sr=sr, chunk_size=cs, padding=pad)
    kw.update(over)
    return SpectralGateStationary(y=y, **kw)


def _nonfinite_agree(got, want, tol=TOL):
    gn, wn = ~np.isfinite(got), ~np.isfinite(want)
    assert np.array_equal(gn, wn), "non-finite samples in different places: engine %d, oracle %d" % (gn.sum(), wn.sum())
    both = ~gn
    if both.any():
        assert np.abs(got[both] - want[both]).max() <= tol * max(1e-3, np.abs(want[both]).max())

TOL = 1e-4

NS_KW = dict(sr=48000, prop_decrease=1.0, chunk_size=100000, padding=8000, n_fft=1024, win_length=None,
             hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
             thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None, use_tqdm=False, n_jobs=1)


def _gate_S(stationary, y):
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    kw = dict(NS_KW)
    if stationary:
        for k in ("thresh_n_mult_nonstationary", "sigmoid_slope_nonstationary"):
            kw.pop(k)
        kw.update(y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True)
        return SpectralGateStationary(y=y, **kw)
    return SpectralGateNonStationary(y=y, **kw)


def _tg_gate(tg):
    (g,) = list(tg._gates.values())
    return g


def _rowgate_vs_float64(x, sr=16000, shape=16):
    """forward on the row gate and on the four-kernel float64 path: (y_rowgate, bits_rowgate, y_f64, bits_f64)."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.torchgate import TorchGate
    tg = TorchGate(sr=sr).cuda()
    xd = x.cuda()
    tg(xd)
    g = _tg_gate(tg)
    try:
        g.set_option(_ffi.SG_OPT_ROWGATE_SHAPE, shape)
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 2)
        y_new = tg(xd).clone()
        bits_new = g.debug_field(3)
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 1)
        y_old = tg(xd).clone()
        bits_old = g.debug_field(3)
    finally:
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0)
        g.set_option(_ffi.SG_OPT_ROWGATE_SHAPE, 16)
    return y_new, bits_new, y_old, bits_old


def _rg_inputs():
    torch.manual_seed(0)
    t16 = torch.arange(16000, dtype=torch.float64) / 16000
    x = (0.1 * torch.randn(24, 16000) + 0.5 * torch.sin(2 * np.pi * 440 * t16).float()).float()
    sp = torch.from_numpy(np.stack([O.synth_signal(16000, sr=16000, seed=s, tone_hz=300.0 + 50 * s) for s in range(8)]))
    chirp = torch.sin(2 * np.pi * (200 * t16 + 3000 * t16 * t16)).float()[None, :] * 0.7 + 0.01 * torch.randn(4, 16000)
    return {"noise+tone 24x16000": x, "T=64 5x16383": x[:5].repeat(1, 2)[:, :16383].contiguous(),
            "short rows 7x3000": x[:7, :3000].contiguous(), "2 W 3x2048": x[:3, :2048].contiguous(),
            "float64 3x16000": x[:3].double(), "synth_signal 8x16000": sp, "chirp 4x16000": chirp.float()}

from tests.golden.cases import S_INF_CASES, make_input_S_inf  # noqa: E402

# ---- one-pass gate: floor test a priori (k_unit_absmax) vs in the gate kernel (SG_OPT_FLOOR_TEST) -------------------


def _floor_inputs(kind):
    rng = np.random.default_rng(1234)
    n, cs, pad = 150000, 40000, 6000
    y = (0.05 * rng.standard_normal(n)).astype(np.float32)
    y_noise = (0.05 * rng.standard_normal(30000)).astype(np.float32)
    if kind == "benign":
        pass
    elif kind == "live":              # loud half next to digital silence, very quiet noise clip: bands lifted by the floor
        y[: n // 2] = 0.0
        y[n // 2:] *= 10.0
        y_noise = (1e-7 * rng.standard_normal(30000)).astype(np.float32)
    elif kind == "loud_in_padding":   # the only loud samples of chunk 1's window sit in its left padding (chunk 0's tail)
        y[:] = (1e-6 * rng.standard_normal(n)).astype(np.float32)
        y[cs - pad + 200: cs - pad + 1500] = (0.9 * rng.standard_normal(1300)).astype(np.float32)
        y_noise = (1e-7 * rng.standard_normal(30000)).astype(np.float32)
    elif kind == "nan_in_padding":    # a NaN that only chunk 2's right padding sees (and chunk 3's body)
        y[3 * cs + 4000] = np.nan
    elif kind == "inf_far_padding":   # an Inf near the far end of chunk 0's right padding
        y[cs + pad - 3] = np.inf
    return y, y_noise, cs, pad


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["signal", "signal_chunks", "noise_clip", "nonstationary", "torchgate_row"])
def test_nan_sample_gates_like_the_reference(nr, case):
    """A NaN sample: numpy / torch maxima and means keep it (stationary.py:75-81, 96-106; torchgate.py:140-160), so the
    whole band -- in practice every band of the chunk / row that sees the sample -- compares False, and the NaN
    itself survives the multiplication by a zero mask.  The engine's reductions use fmax (which drops a NaN):
    non-finite samples are tracked explicitly (bit-pattern maximum in k_unit_absmax, NaN-sticky row maxima,
    T2_NEVER compare constants).  Same non-finite output samples, same finite rest."""
    rng = np.random.default_rng(7)
    n = 48000 * 2
    y = (0.1 * rng.standard_normal(n)).astype(np.float32)
    with np.errstate(all="ignore"):
        if case == "torchgate_row":
            from noisereduce_amd.torchgate import TorchGate
            x = (0.1 * rng.standard_normal((6, 16000))).astype(np.float32)
            x[2, 9000] = np.nan
            got = TorchGate(sr=16000, nonstationary=False).cuda()(torch.from_numpy(x).cuda()).cpu().numpy()
            want = O.torchgate_T(x.astype(np.float64), 16000, nonstationary=False,
                                 window=torch.hann_window(1024).double().numpy())
            assert np.isnan(got[2]).any() and np.isfinite(got[[0, 1, 3, 4, 5]]).all()
        elif case == "noise_clip":
            yn = (0.1 * rng.standard_normal(30000)).astype(np.float32)
            yn[4000] = np.nan
            got = nr.reduce_noise(y=y, sr=48000, y_noise=yn, stationary=True, n_fft=1024)
            want = O.reduce_noise_S(y.astype(np.float64), 48000, y_noise=yn.astype(np.float64), stationary=True, n_fft=1024)
            assert np.isfinite(got).all() and np.abs(got).max() == 0.0   # NaN thresholds: everything is gated
        else:
            y[50000] = np.nan
            kw = dict(sr=48000, stationary=case != "nonstationary", n_fft=1024)
            if case == "signal_chunks":
                kw.update(chunk_size=20000, padding=2000)
            got = nr.reduce_noise(y=y, **kw)
            want = O.reduce_noise_S(y.astype(np.float64), **kw)
            if case == "signal_chunks":   # only the chunks that see the sample are gated
                assert np.abs(got[:20000]).max() > 0 and np.isfinite(got[:20000]).all()
    _nonfinite_agree(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,smooth", [(512, True), (2048, True), (1000, True), (4096, True), (4096, False)])
def test_nan_sample_other_frame_lengths(nr, n_fft, smooth):
    """The same NaN rule on the kernels of the other frame lengths (float32 LDS decision, chirp-z transform, the
    unfused power-field path with per-band maxima)."""
    rng = np.random.default_rng(3)
    n = 48000 * 2
    y = (0.1 * rng.standard_normal(n)).astype(np.float32)
    y[50000] = np.nan
    kw = dict(sr=48000, stationary=True, n_fft=n_fft, chunk_size=20000, padding=2000)
    if not smooth:
        kw.update(freq_mask_smooth_hz=None, time_mask_smooth_ms=None)
    got = nr.reduce_noise(y=y, **kw)
    with np.errstate(all="ignore"):
        want = O.reduce_noise_S(y.astype(np.float64), **kw)
    assert np.abs(got[:20000]).max() > 0
    _nonfinite_agree(got, want)


@pytest.mark.parametrize("name", sorted(S_INF_CASES))
def test_inf_sample_against_the_reference_golden(nr, golden_dir, name):
    """The reference's result around an Inf sample depends on which bins of ITS FFT come out Inf (band passes in that
    chunk) and which NaN (band gated): pocketfft's butterfly order.  The engine gates an Inf like a NaN (DESIGN.md,
    stated deviation).  Pinned here: (a) the same output samples are non-finite; (b) every chunk the Inf does not reach
    equals the reference's golden; (c) inside the affected chunk the engine returns exactly what it returns for a NaN at
    the same place (and that differs from the reference by more than the 1e-4 bar: the deviation is real)."""
    case = S_INF_CASES[name]
    gold = np.load(os.path.join(golden_dir, "S_inf_%s.npz" % name))["out"]
    y, _ = make_input_S_inf(case)
    out = nr.reduce_noise(y=y, sr=case["sr"], **case["kwargs"])
    nf = ~np.isfinite(out)
    assert np.array_equal(nf, ~np.isfinite(gold))                                              # (a)
    cs = case["kwargs"]["chunk_size"]
    chunk = case["inf_at"] // cs
    other = np.ones(out.shape, bool)
    other[chunk * cs:(chunk + 1) * cs] = False
    peak = np.abs(gold[np.isfinite(gold)]).max()
    assert np.abs(out[other] - gold[other]).max() / peak < TOL                                  # (b)
    ynan = y.copy()
    ynan[..., case["inf_at"]] = np.nan
    out_nan = nr.reduce_noise(y=ynan, sr=case["sr"], **case["kwargs"])
    fin = ~nf
    assert np.array_equal(~np.isfinite(out_nan), nf) and np.array_equal(out[fin], out_nan[fin])    # (c)
    inside = fin & ~other
    dev = np.abs(out[inside] - gold[inside]).max() / peak
    assert dev > TOL, dev         # if this ever fails the engine has started to match the reference: update DESIGN.md
