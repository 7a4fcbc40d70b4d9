"""Engine handles shared between objects / threads, workspace growth and sizing (the boundary's ownership and threading rules).
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


def test_two_objects_same_settings_keep_their_own_threshold(nr):
    """A = SG(yA), B = SG(yB) with IDENTICAL settings share one cached engine handle.  Each must keep
    filtering with ITS OWN noise statistics whatever the order of calls (the reference keeps
    noise_thresh per object, stationary.py:47-81)."""
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    kw = dict(SG_KW, chunk_size=40000, padding=5000)
    yA = O.synth_signal(130000, seed=101, noise_sigma=0.02).astype(np.float64)
    yB = O.synth_signal(90000, seed=202, noise_sigma=0.3, tone_hz=3000.0).astype(np.float64)
    nB = (0.05 * np.random.default_rng(9).standard_normal(30000))
    wantA = O.reduce_noise_S(yA, 48000, stationary=True, chunk_size=40000, padding=5000)
    wantB = O.reduce_noise_S(yB, 48000, stationary=True, y_noise=nB, chunk_size=40000, padding=5000)
    thrA, _, _ = O.noise_threshold_S(yA[None, :], 1024, 1024, 256, 1.5, 40000)
    thrB, _, _ = O.noise_threshold_S(nB[None, :], 1024, 1024, 256, 1.5, 40000)
    assert O.rel_err(wantA, O.reduce_noise_S(yA, 48000, stationary=True, y_noise=nB, chunk_size=40000,
                                             padding=5000)) > 1e-2, "test inputs must be threshold-sensitive"

    A = SpectralGateStationary(y=yA, **kw)
    B = SpectralGateStationary(y=yB, **dict(kw, y_noise=nB))       # same handle, overwrites h->thresh
    assert A._gate is B._gate, "the test needs both objects on one cached handle"
    assert O.rel_err(A.get_traces(), wantA) < TOL                    # A after B was built
    assert np.max(np.abs(A.noise_thresh - thrA)) < 1e-9
    assert O.rel_err(B.get_traces(), wantB) < TOL
    assert np.max(np.abs(B.noise_thresh - thrB)) < 1e-9
    # a reduce_noise() call with the same settings and a third recording in between
    nr.reduce_noise(y=O.synth_signal(50000, seed=7).astype(np.float64), sr=48000, stationary=True,
                    chunk_size=40000, padding=5000)
    assert O.rel_err(A.get_traces(), wantA) < TOL
    # sub-range and operator seam of A right after B used the handle
    assert O.rel_err(B.get_traces(), wantB) < TOL
    sub = A.get_traces(start_frame=10000, end_frame=100000)
    assert O.rel_err(sub, wantA[10000:100000]) < TOL
    B.get_traces()
    chunk = O.read_chunk(yA[None, :], -5000, 45000)
    filt = O.smoothing_filter(5, 9)
    ref = O.gate_stationary_S(chunk, thrA, 1024, 1024, 256, 1.0, filt)
    assert O.rel_err(A._do_filter(chunk), ref) < TOL


def test_threads_share_a_cached_handle_safely(nr):
    """reduce_noise() from several Python threads with the same settings (one cached handle): the
    handle lock serialises statistics -> filter, every thread gets ITS recording's result."""
    ys = [O.synth_signal(70000, seed=300 + i, noise_sigma=0.02 * (1 + 3 * i)).astype(np.float64) for i in range(4)]
    wants = [O.reduce_noise_S(y, 48000, stationary=True, chunk_size=30000, padding=4000) for y in ys]
    errs = [None] * len(ys)

    def work(i):
        try:
            worst = 0.0
            for _ in range(5):
                got = nr.reduce_noise(y=ys[i], sr=48000, stationary=True, chunk_size=30000, padding=4000)
                worst = max(worst, O.rel_err(got, wants[i]))
            errs[i] = worst
        except Exception as e:   # surfaced below
            errs[i] = e

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(ys))]
    [t.start() for t in th]
    [t.join() for t in th]
    for e in errs:
        assert not isinstance(e, Exception), e
        assert e < TOL, errs


def test_workspace_regrow_is_zeroed(nr):
    """exchange buffers that grow are zero-filled even when the allocator hands the old address back --
    a small call, a larger one, the small one again, against the oracle each time."""
    sm = O.synth_signal(60000, seed=7).astype(np.float32)
    lg = np.stack([O.synth_signal(400000, seed=8 + c) for c in range(3)]).astype(np.float32)
    for stationary in (True, False):
        for y in (sm, lg, sm):
            got = nr.reduce_noise(y=y, sr=48000, stationary=stationary, chunk_size=50000, padding=4000)
            want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=stationary, chunk_size=50000, padding=4000)
            assert O.rel_err(got, want) < TOL


def test_workspace_bytes_covers_the_float64_pipeline(nr):
    """sg_workspace_bytes is an upper bound for EVERY sample type of the call to come -- an int16 recording
    takes the float64 pipeline (32 B per cell + float64 frames); the figure must not be the few MB of the bit-mask path."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y16 = (O.synth_signal(300000, seed=9) * 20000).astype(np.int16)
    kw = dict(sr=48000, y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, prop_decrease=1.0,
              chunk_size=100000, padding=8000, n_fft=1024, win_length=None, hop_length=None, time_constant_s=2.0,
              freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=torch.from_numpy(y16).cuda(), **kw)
    est = sg._gate.workspace_bytes(1, 300000, chunked=True)
    T, FS, units = 116000 // 256 + 1, 528, 3
    exact = units * (T * FS * 32 + T * 1024 * 8 + FS * 16)
    assert est >= exact, (est, exact)
    free0 = torch.cuda.mem_get_info()[0]
    out = sg.get_traces()
    torch.cuda.synchronize()
    used = free0 - torch.cuda.mem_get_info()[0]
    assert out.dtype == torch.int16
    assert used <= est + (64 << 20), (used, est)      # what the call really allocated (allocator granularity aside)
