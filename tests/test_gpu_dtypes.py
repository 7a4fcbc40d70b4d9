"""Sample types (row f2): integer recordings bit-exact, float64 pipeline (precision="float64" / SG_OPT_FORCE_EXACT).
(grouped by subject in round 5; the tests themselves date from rounds 2-4)"""
import threading

import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4

NS_KW = dict(sr=48000, prop_decrease=1.0, chunk_size=100000, padding=8000, n_fft=1024, win_length=None,
             hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
             thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None, use_tqdm=False, n_jobs=1)


@pytest.fixture(scope="module")
def nr():
    import noisereduce_amd
    return noisereduce_amd


@pytest.mark.parametrize("kw", [
    dict(stationary=True),                                                 # default geometry (fused path when float)
    dict(stationary=False),
    dict(stationary=True, prop_decrease=0.7, chunk_size=30000, padding=3000),
    dict(stationary=False, n_fft=512, chunk_size=30000, padding=3000),
    dict(stationary=True, n_fft=1000),                                     # chirp-z frames
    dict(stationary=False, n_fft=2048, win_length=1500, hop_length=300, time_mask_smooth_ms=None),
    dict(stationary=True, n_fft=8192, freq_mask_smooth_hz=None),
    dict(stationary=True, n_fft=5000, time_mask_smooth_ms=None, freq_mask_smooth_hz=None),   # long chirp-z frames, no smoothing
])
@pytest.mark.parametrize("dtype", [np.int16, np.int32])
def test_integer_recordings_are_bit_exact(nr, kw, dtype):
    """int16 / int32 in -> the same dtype out = trunc(float64 result) (base.py:217-226), for both gates and every
    family of transform kernels, against the oracle's float64 result truncated the same way."""
    n = 90000
    scale = 20000 if dtype == np.int16 else 1.5e9
    y = np.stack([np.round(O.synth_signal(n, seed=71 + c, tone_hz=500.0 * (c + 1)).astype(np.float64) * scale) for c in range(2)]).astype(dtype)
    got = nr.reduce_noise(y=y, sr=48000, **kw)
    want64 = O.reduce_noise_S(y.astype(np.float64), 48000, **kw)
    want = want64.astype(dtype)
    assert got.dtype == dtype and got.shape == y.shape
    diff = got.astype(np.int64) - want.astype(np.int64)
    # float64 evaluation order differs from numpy's (1e-16 relative): a value within 1e-9 (int16) / 1e-4 (int32 at 1.5e9:
    # 1e-13 relative) of an integer may fall on the other side -- e.g. where the mask is exactly 1 and the gate reconstructs
    # the integer input to ~1e-12, the reference's own truncation is rounding noise.  Everywhere else: equal.
    decided = np.abs(want64 - np.round(want64)) > (1e-9 if dtype == np.int16 else 1e-4)
    assert np.max(np.abs(diff)) <= 1 and np.count_nonzero(diff[decided]) == 0, np.count_nonzero(diff[decided])
    assert np.count_nonzero(decided) > 0.9 * decided.size


def test_force_exact_float64(nr):
    """SG_OPT_FORCE_EXACT: float64 recordings get float64-accurate results (1e-12 of peak instead of 2e-7)."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = O.synth_signal(120000, seed=9).astype(np.float64)
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=50000,
              clip_noise_stationary=True, padding=4000, n_fft=1024, win_length=None, hop_length=None, time_constant_s=2.0,
              freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=y, **kw)
    want = O.reduce_noise_S(y, 48000, stationary=True, chunk_size=50000, padding=4000)
    assert 1e-9 < O.rel_err(sg.get_traces(), want) < TOL
    sg._gate.set_option(_ffi.SG_OPT_FORCE_EXACT, 1)
    try:
        assert O.rel_err(sg.get_traces(), want) < 1e-12
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_EXACT, 0)


@pytest.mark.parametrize("stationary", [True, False])
def test_precision_float64_matches_the_float64_golden(nr, golden_dir, stationary, monkeypatch):
    """reduce_noise(..., precision="float64") / NOISEREDUCE_AMD_EXACT=1: a float64 recording gets what the reference
    computes for it (base.py:140: every dtype in float64) -- checked on a golden made by the LIVE reference to <= 1e-12 of
    peak; the default (fused float32 kernels, float64 container) is float32-accurate on the same input."""
    import os
    from tests.golden.cases import S_CASES, make_input_S
    name = "stat_chunked" if stationary else "nonstat_chunked"
    case = S_CASES[name]
    want = np.load(os.path.join(golden_dir, "S_" + name + ".npz"))["out"]
    y, _ = make_input_S(case)                      # float64 (float32-valued), what the reference was fed
    kw = dict(case["kwargs"], sr=case["sr"])
    got32 = nr.reduce_noise(y=y, **kw)
    got64 = nr.reduce_noise(y=y, precision="float64", **kw)
    assert got64.dtype == np.float64 and got32.dtype == np.float64
    e32, e64 = O.rel_err(got32, want), O.rel_err(got64, want)
    assert e64 <= 1e-12, e64
    assert 1e-9 < e32 < TOL, e32
    # the environment switch selects the same pipeline when precision is left at None; precision="float32" overrides it
    monkeypatch.setenv("NOISEREDUCE_AMD_EXACT", "1")
    assert np.array_equal(nr.reduce_noise(y=y, **kw), got64)
    assert np.array_equal(nr.reduce_noise(y=y, precision="float32", **kw), got32)
    with pytest.raises(ValueError):
        nr.reduce_noise(y=y, precision="float16", **kw)


def test_with_options_restores_non_zero_defaults(nr):
    """Gate.get_option reads the handle (sg_get_option): with_options restores the library's DEFAULT of an option that was
    never set from Python (SG_OPT_ROWGATE_SHAPE defaults to 16 and rejects 0)."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.torchgate import TorchGate
    tg = TorchGate(sr=16000).to("cuda:0")
    tg(torch.randn(4, 16000, device="cuda:0"))
    (g,) = list(tg._gates.values())
    assert g.get_option(_ffi.SG_OPT_ROWGATE_SHAPE) == 16
    with g.with_options([(_ffi.SG_OPT_ROWGATE_SHAPE, 8), (_ffi.SG_OPT_FORCE_NOROWGATE, 2)]):
        assert g.get_option(_ffi.SG_OPT_ROWGATE_SHAPE) == 8 and g.get_option(_ffi.SG_OPT_FORCE_NOROWGATE) == 2
    assert g.get_option(_ffi.SG_OPT_ROWGATE_SHAPE) == 16 and g.get_option(_ffi.SG_OPT_FORCE_NOROWGATE) == 0
    with pytest.raises(ValueError):
        g.get_option(9999)


@pytest.mark.parametrize("dtype", [np.int16, np.int32, np.float64])
@pytest.mark.parametrize("kw", [
    dict(),                                                   # one window
    dict(chunk_size=30000, padding=3000),                     # chunk grid, last chunk partial
    dict(chunk_size=25000, padding=4000, n_std_thresh_stationary=0.5, freq_mask_smooth_hz=1000, time_mask_smooth_ms=20),
])
def test_fused_float64_apply_equals_the_materialised_float64_pipeline(nr, kw, dtype):
    """Round 5: the stationary gate's float64 pipeline at the default geometry is mask bits (exact on the fused path) +
    k_apply_fast64 (transforms, mask multiply and overlap-add in double) instead of five float64 fields through HBM.  Same
    numbers as the materialised pipeline (exact.hpp, SG_OPT_EXACT_MATERIALISED) to float64 rounding, and -- for integer
    recordings -- the same truncated samples wherever the float64 value is not within 1e-9 of an integer."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    n = 70000
    scale = {np.int16: 20000, np.int32: 1.5e9, np.float64: 1.0}[dtype]
    y = np.stack([O.synth_signal(n, seed=81 + c, tone_hz=700.0 * (c + 1)).astype(np.float64) * scale for c in range(2)])
    y = (np.round(y) if dtype != np.float64 else y).astype(dtype)
    base = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
                clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None, time_constant_s=2.0,
                freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    base.update(kw)
    sg = SpectralGateStationary(y=y, precision="float64", **base)
    fused = sg.get_traces()
    with sg._gate.with_options([(_ffi.SG_OPT_EXACT_MATERIALISED, 1)]):
        mat = sg.get_traces()
    assert fused.dtype == dtype and fused.shape == y.shape
    okw = {k: base[k] for k in ("chunk_size", "padding", "n_std_thresh_stationary", "freq_mask_smooth_hz", "time_mask_smooth_ms")}
    want64 = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=True, **okw)
    if dtype == np.float64:
        peak = np.abs(want64).max()
        assert np.abs(fused - mat).max() <= 1e-13 * peak
        assert np.abs(fused - want64).max() <= 1e-12 * peak
    else:
        tol = 1e-9 if dtype == np.int16 else 1e-4
        decided = np.abs(want64 - np.round(want64)) > tol
        d = fused.astype(np.int64) - want64.astype(dtype).astype(np.int64)
        assert np.abs(d).max() <= 1 and np.count_nonzero(d[decided]) == 0
        dm = fused.astype(np.int64) - mat.astype(np.int64)
        assert np.abs(dm).max() <= 1 and np.count_nonzero(dm[decided]) == 0


def test_fused_float64_apply_sub_range(nr):
    """get_traces(start_frame, end_frame) on the float64 path: only the requested samples, equal to the same slice of the
    whole result (chunk grid anchored at start_frame like base.py:175-216)."""
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = np.round(O.synth_signal(120000, seed=5).astype(np.float64) * 15000).astype(np.int16)
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=20000,
              clip_noise_stationary=True, padding=3000, n_fft=1024, win_length=None, hop_length=None, time_constant_s=2.0,
              freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=y, **kw)
    part = sg.get_traces(start_frame=20000, end_frame=95000)
    want64 = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=True, chunk_size=20000, padding=3000)[20000:95000]
    assert part.shape == (75000,) and part.dtype == np.int16
    decided = np.abs(want64 - np.round(want64)) > 1e-9
    d = part.astype(np.int64) - want64.astype(np.int16).astype(np.int64)
    assert np.abs(d).max() <= 1 and np.count_nonzero(d[decided]) == 0


@pytest.mark.parametrize("dtype", [np.int16, np.float64])
@pytest.mark.parametrize("kw", [dict(), dict(chunk_size=30000, padding=3000, prop_decrease=0.8),
                                dict(stationary=True, prop_decrease=0.7, chunk_size=25000, padding=4000)])
def test_float64_pipeline_equals_its_round4_form(nr, kw, dtype):
    """The float64 pipeline of round 5 (tile-parallel recurrence, LDS-tiled smoothing, register float64 transform, k_apply_fast64
    on the float64 mask field) against the materialised one it replaces (SG_OPT_EXACT_MATERIALISED: serial recurrence, two direct
    smoothing passes, masked frames + gather): non-stationary gate, and the stationary gate with prop_decrease < 1 (which keeps
    float64 mask fields).  float64 outputs equal to 1e-13 of peak, integer outputs equal wherever the value is not within 1e-9
    of an integer; both against the oracle."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    kw = dict(kw)
    stationary = kw.pop("stationary", False)
    n = 70000
    scale = 20000 if dtype == np.int16 else 1.0
    y = np.stack([O.synth_signal(n, seed=91 + c, tone_hz=600.0 * (c + 1)).astype(np.float64) * scale for c in range(2)])
    y = (np.round(y) if dtype == np.int16 else y).astype(dtype)
    base = dict(sr=48000, prop_decrease=1.0, chunk_size=600000, padding=30000, n_fft=1024, win_length=None, hop_length=None,
                time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    base.update(kw)
    if stationary:
        sg = SpectralGateStationary(y=y, y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, precision="float64", **base)
    else:
        sg = SpectralGateNonStationary(y=y, thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, precision="float64", **base)
    new = sg.get_traces()
    with sg._gate.with_options([(_ffi.SG_OPT_EXACT_MATERIALISED, 1)]):
        old = sg.get_traces()
    okw = {k: base[k] for k in ("chunk_size", "padding", "prop_decrease")}
    want64 = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=stationary, **okw)
    if dtype == np.float64:
        peak = np.abs(want64).max()
        assert np.abs(new - old).max() <= 1e-13 * peak
        assert np.abs(new - want64).max() <= 1e-12 * peak
    else:
        decided = np.abs(want64 - np.round(want64)) > 1e-9
        for got in (new, old):
            d = got.astype(np.int64) - want64.astype(dtype).astype(np.int64)
            assert np.abs(d).max() <= 1 and np.count_nonzero(d[decided]) == 0
