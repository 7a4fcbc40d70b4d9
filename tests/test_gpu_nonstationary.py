"""Non-stationary gates (rows a11-a12; TorchGate non-stationary): two-pass recurrence + one-kernel mask stages.
(grouped by subject in round 5; the tests themselves date from rounds 2-4)"""
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


def test_nonstationary_silent_chunk_is_nan_like_the_reference(nr):
    """nonstationary.py:70 divides by the smoothed magnitude: a band that is exactly zero over a WHOLE
    padded chunk gives 0/0 = NaN, the mask is NaN there, every frame's inverse transform touches a NaN
    bin and the reference returns NaN for the whole chunk (probed on the live reference: all-zero input
    -> all-NaN output; half-silent input -> finite, the forward-backward IIR spreads energy over the
    chunk).  The engine reproduces that: same NaN pattern, same finite samples."""
    z = np.zeros(30000)
    want = O.reduce_noise_S(z, 48000, stationary=False)
    assert np.isnan(want).all()
    got = nr.reduce_noise(y=z, sr=48000, stationary=False)
    assert got.shape == z.shape and np.isnan(got).all()
    # float32 samples, and a recording in which exactly one padded chunk is silent
    assert np.isnan(nr.reduce_noise(y=z.astype(np.float32), sr=48000, stationary=False)).all()
    rng = np.random.default_rng(5)
    cs, pad = 20000, 3000
    y = 0.1 * rng.standard_normal(5 * cs)
    y[2 * cs - pad - 2000:3 * cs + pad + 2000] = 0.0          # chunk 2 incl. its padding (and a margin) is silent
    y = y.astype(np.float32).astype(np.float64)
    want = O.reduce_noise_S(y, 48000, stationary=False, chunk_size=cs, padding=pad)
    got = nr.reduce_noise(y=y, sr=48000, stationary=False, chunk_size=cs, padding=pad)
    assert np.isnan(want[2 * cs:3 * cs]).all() and np.isfinite(want[:2 * cs]).all()
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = np.isfinite(want)
    assert O.rel_err(got[ok], want[ok]) < TOL
    # half-silent single chunk: finite everywhere (reference behaviour), parity as usual
    y2 = np.zeros(60000)
    y2[30000:] = 0.1 * rng.standard_normal(30000)
    want2 = O.reduce_noise_S(y2, 48000, stationary=False)
    got2 = nr.reduce_noise(y=y2, sr=48000, stationary=False)
    assert np.isfinite(want2).all() and np.isfinite(got2).all()
    assert O.rel_err(got2, want2) < TOL


@pytest.mark.parametrize("sr,kw", [
    (48000, dict()),                                         # nt = 9, nf = 5 (the default geometry)
    (48000, dict(n_fft=512)),                                # hop 128: nt = 18, nf = 2; general STFT kernels
    (64000, dict()),                                         # nt = 12
    (48000, dict(time_mask_smooth_ms=30)),                   # nt = 5
    (48000, dict(time_mask_smooth_ms=90)),                   # nt = 16
    (48000, dict(time_mask_smooth_ms=60)),                   # nt = 11: not instantiated -> segmented-scan kernels
    (22050, dict(time_constant_s=0.5, prop_decrease=0.7)),   # nt = 4, short time constant, partial reduction
    (48000, dict(time_constant_s=0.05)),                     # c^rows too small for the reverse regeneration -> old kernels
    (48000, dict(freq_mask_smooth_hz=None)),                 # nf = 1 ... time smoothing only
])
def test_nonstationary_two_pass_mask(nr, sr, kw):
    """k_iir_part / k_iir_chain / k_iir_mask (or the fall-back kernels) against the oracle's filtfilt + sigmoid +
    fftconvolve (nonstationary.py:47-115), chunked so that tiles touch both unit edges and a short last tile."""
    n, cs, pad = 170000, 50000, 6000
    y = np.stack([O.synth_signal(n, sr=sr, seed=31 + c, tone_hz=700.0 * (c + 1)) for c in range(2)]).astype(np.float32)
    got = nr.reduce_noise(y=y, sr=sr, stationary=False, chunk_size=cs, padding=pad, **kw)
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=False, chunk_size=cs, padding=pad, **kw)
    assert O.rel_err(got, want) < TOL


@pytest.mark.parametrize("sr,n_fft,tms,fhz", [
    (44100, 512, 50, 500),        # nt = 17 (no instantiation before round 3), nf = 2
    (48000, 2048, 50, 500),       # nt = 4, nf = 10 (> 8: the general kernels before round 3)
    (48000, 4096, 100, 500),      # nt = 4, nf = 21
    (48000, 1024, 110, 300),      # nt = 20, nf = 3: the widest instantiated time half-width
    (48000, 256, 50, 500),        # nt = 37: chain-based raw sigmoid (k_iir_mask<0>) + general smoothing
    (48000, 1024, None, 800),     # frequency smoothing only (the reference's 3-tap time filter)
])
def test_nonstationary_mask_widths(nr, sr, n_fft, tms, fhz):
    y = O.synth_signal(sr * 2 + 777, sr=sr, seed=n_fft + (tms or 0)).astype(np.float32)
    kw = dict(stationary=False, n_fft=n_fft, time_mask_smooth_ms=tms, freq_mask_smooth_hz=fhz, chunk_size=60000, padding=3000)
    got = nr.reduce_noise(y=y, sr=sr, **kw)
    want = O.reduce_noise_S(y.astype(np.float64), sr, **kw)
    assert O.rel_err(got, want) < TOL


def test_one_kernel_masks_equal_the_general_kernels(nr):
    """k_iir_mask / k_box_mask against the materialised kernels behind SG_OPT_FORCE_UNFUSED: same mask up to float32
    rounding of a different summation order; the raw field is only fetchable from the latter."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    y = torch.from_numpy(O.synth_signal(48000 * 2, seed=5).astype(np.float32)).cuda()
    kw = dict(NS_KW); kw.update(y=y, chunk_size=48000, padding=4000)
    sg = SpectralGateNonStationary(**kw)
    out = sg.get_traces().cpu().numpy()
    M = sg._gate.debug_field(1)
    with pytest.raises(RuntimeError, match="FORCE_UNFUSED"):
        sg._gate.debug_field(0)
    sg._gate.set_option(_ffi.SG_OPT_FORCE_UNFUSED, 1)
    try:
        out_u = SpectralGateNonStationary(**kw).get_traces().cpu().numpy()
        M_u = sg._gate.debug_field(1)
        raw = sg._gate.debug_field(0)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_UNFUSED, 0)
    assert raw.shape == M.shape and np.isfinite(raw).all()
    assert np.max(np.abs(M - M_u)) < 2e-5
    assert np.max(np.abs(out - out_u)) < 1e-5 * np.max(np.abs(out_u))


@pytest.mark.parametrize("sr,L,kw", [
    (48000, 48000, dict()),                                          # nt = 9, nf = 5: interior and edge tiles
    (16000, 16000, dict()),                                          # nt = 3, nf = 16: half of every wave is halo columns
    (8000, 9000, dict(prop_decrease=0.7)),                           # nt = 1
    (48000, 20000, dict(time_mask_smooth_ms=None)),                  # one-axis smoothing
    (22050, 30000, dict(freq_mask_smooth_hz=None, time_mask_smooth_ms=None, prop_decrease=0.5)),   # no smoothing: NT = 0
    (48000, 5000, dict(n_movemean_nonstationary=7)),                 # another moving-mean length: k_boxcar_sigmoid path
])
def test_torchgate_nonstationary_one_kernel_mask(nr, sr, L, kw):
    from noisereduce_amd.torchgate import TorchGate
    x = np.stack([O.synth_signal(L, sr=sr, seed=s + L, tone_hz=300.0 * (s + 1)) for s in range(3)]).astype(np.float64)
    y = TorchGate(sr=sr, nonstationary=True, **kw).cuda()(torch.from_numpy(x).cuda()).cpu().numpy()
    want = O.torchgate_T(x, sr, nonstationary=True, window=torch.hann_window(1024).double().numpy(), **kw)
    assert O.rel_err(y, want) < TOL


def test_nonstationary_window_of_24_minutes(nr):
    """One window (chunk_size=None) of 24 min at 48 kHz = 270 k frames = 4219 time tiles: k_iir_chain used to keep
    a per-tile table in LDS and did not launch beyond 4096 tiles (23 min).  Against the materialised kernels behind
    SG_OPT_FORCE_UNFUSED (no chain), whole output; tests/tools/long_window_check.py holds an hour against the oracle."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    sr, n = 48000, 48000 * 60 * 24
    g = torch.Generator(device="cuda").manual_seed(3)
    y = 0.05 * torch.randn(n, device="cuda", generator=g)
    y += 0.3 * torch.sin(2 * np.pi * 700.0 * torch.arange(n, device="cuda", dtype=torch.float32) / sr)
    kw = dict(NS_KW); kw.update(y=y, chunk_size=None, padding=30000)
    sg = SpectralGateNonStationary(**kw)
    a = sg.get_traces()
    sg._gate.set_option(_ffi.SG_OPT_FORCE_UNFUSED, 1)
    try:
        b = SpectralGateNonStationary(**kw).get_traces()
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_UNFUSED, 0)
    assert bool(torch.isfinite(a).all())
    assert float((a - b).abs().max()) < 1e-5 * float(b.abs().max())


@pytest.mark.parametrize("n_fft,n,kw", [
    (1024, 48000 * 20, dict()),                                             # 16-frame sub-tiles, 2 chunks, ragged last tile
    (1024, 70001, dict(chunk_size=30000, padding=3000)),                    # short units: a few tiles, runs of one tile
    (1024, 1500, dict()),                                                   # a single tile, fewer than four sub-tiles
    (1024, 48000 * 30, dict(chunk_size=1300000, padding=40000)),            # 85 tiles per unit: six tiles of sub-tiles exceed a run -> k_iir_comb + the chain on tile partials
    (512, 300000, dict()),                                                  # k_iir_part's tile partials (PER = 1)
    (256, 200000, dict(chunk_size=90000, padding=5000)),
    (2048, 250000, dict()),
])
def test_parallel_tile_chain_equals_the_serial_chain(nr, n_fft, n, kw):
    """k_iir_chain_par (round 5: the non-stationary gate's tile chain in 16 composed runs per band, from the 16-frame sub-tile
    partials at the default geometry) against k_iir_comb + k_iir_chain (SG_OPT_FORCE_SPLIT) and the oracle
    (nonstationary.py:106-115: filtfilt along time)."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    sr = 48000
    y = np.stack([O.synth_signal(n, sr=sr, seed=11 + c, tone_hz=500.0 * (c + 1)) for c in range(2)]).astype(np.float32)
    y[0, n // 3:n // 3 + 5000] *= 30.0
    yd = torch.from_numpy(y).cuda()

    def make():
        return SpectralGateNonStationary(y=yd, sr=sr, chunk_size=kw.get("chunk_size", 600000), padding=kw.get("padding", 30000),
                                         prop_decrease=1.0, n_fft=n_fft, win_length=None, hop_length=None, time_constant_s=2.0,
                                         freq_mask_smooth_hz=500, time_mask_smooth_ms=50, thresh_n_mult_nonstationary=2,
                                         sigmoid_slope_nonstationary=10, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = make()
    a = sg.get_traces().cpu().numpy()
    with sg._gate.with_options([(_ffi.SG_OPT_FORCE_SPLIT, 1)]):
        b = make().get_traces().cpu().numpy()
    assert O.rel_err(a, b) < 1e-6
    if n <= 300000:
        want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=False, n_fft=n_fft, **kw)
        assert O.rel_err(a, want) < 1e-4
