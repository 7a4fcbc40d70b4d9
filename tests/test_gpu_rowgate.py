"""TorchGate.forward as one kernel per call, k_row_gate (row a13), and the float64 power kernel of the four-kernel path.
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
@pytest.mark.parametrize("B,L,xn", [(256, 16000, False), (64, 40000, False), (96, 20000, True)])
def test_float64_power_kernel_matches_the_lds_transform(B, L, xn):
    """`k_power_fast64` (float64 powers on the register FFT core; short rows: fused row statistics, long rows:
    per-band maxima by atomics) against the nine-pass LDS transform it replaces (`SG_OPT_FORCE_NOFAST`) -- same
    output -- and against the oracle (torchgate.py:200-264) on a few rows."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.torchgate import TorchGate
    torch.manual_seed(B + L)
    t = torch.arange(L, dtype=torch.float64) / 16000
    x = (0.1 * torch.randn(B, L, dtype=torch.float64) + 0.3 * torch.sin(2 * np.pi * 700 * t)).float().cuda()
    noise = (0.1 * torch.randn(B, 12000, dtype=torch.float64)).float().cuda() if xn else None
    tg = TorchGate(sr=16000, nonstationary=False).cuda()
    gate = tg._gate_for(x.device)
    y_fast = gate.process_batch(x, noise)
    try:
        gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 1)
        y_ref = gate.process_batch(x, noise)
    finally:
        gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 0)
    # one flipped mask cell would show as ~1e-4 of the peak; float32 rounding of the two apply kernels as ~1e-7
    assert O.rel_err(y_fast.cpu().numpy(), y_ref.cpu().numpy()) < 2e-6
    rows = [0, B // 2, B - 1]
    want = O.torchgate_T(x[rows].cpu().numpy().astype(np.float64), 16000, nonstationary=False,
                         xn=None if noise is None else noise[rows].cpu().numpy().astype(np.float64),
                         window=torch.hann_window(1024).double().numpy())
    assert O.rel_err(y_fast[rows].cpu().numpy(), want) < TOL


@pytest.mark.parametrize("shape", [16, 8])
@pytest.mark.parametrize("name", sorted(_rg_inputs()))
def test_rowgate_decisions_equal_the_float64_path(nr, name, shape):
    """Mask bits IDENTICAL to the float64 transform + k_row_decide on every cell, output within 1e-6 of that path and
    within the 1e-4 bar of the CPU oracle -- both workgroup shapes (16 waves x 1 quad, 8 waves x 2 quads)."""
    x = _rg_inputs()[name]
    y_new, b_new, y_old, b_old = _rowgate_vs_float64(x, shape=shape)
    assert b_new.shape == b_old.shape and np.array_equal(b_new, b_old), int((b_new != b_old).sum())
    assert O.rel_err(y_new.cpu().numpy(), y_old.cpu().numpy()) < 1e-6
    want = O.torchgate_T(x.numpy().astype(np.float64), 16000, window=torch.hann_window(1024).double().numpy())
    assert y_new.dtype == x.dtype and tuple(y_new.shape) == want.shape
    assert O.rel_err(y_new.cpu().numpy(), want) < TOL


def test_rowgate_silent_tiny_and_nan_rows(nr):
    """Digital silence (nothing passes), a row at -140 dBFS (the reference's eps matters: float64 decides), a NaN sample
    (its row is gated like the reference gates it): same bits, same NaN pattern, same numbers as the float64 path."""
    x = _rg_inputs()["noise+tone 24x16000"][:6].clone()
    x[1] = 0
    x[3] *= 1e-7
    x[4, 5000] = float("nan")
    y_new, b_new, y_old, b_old = _rowgate_vs_float64(x)
    assert np.array_equal(b_new, b_old)
    assert torch.equal(torch.isnan(y_new), torch.isnan(y_old))
    fin = torch.isfinite(y_old)
    assert float((y_new[fin] - y_old[fin]).abs().max()) < 1e-6 * float(y_old[fin].abs().max())
    assert float(y_new[1].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_rowgate_on_the_reference_golden(nr, golden_dir, dtype):
    """The reference's own TorchGate output (tests/golden/T_stat.npz, made by the live reference) through the row gate."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.torchgate import TorchGate
    from tests.golden.cases import T_CASES, make_input_T
    case = T_CASES["stat"]
    gold = np.load(os.path.join(golden_dir, "T_stat.npz"))
    x, _ = make_input_T(case)
    tg = TorchGate(sr=case["sr"], **case["kwargs"]).cuda()
    xt = torch.from_numpy(x).to(dtype).cuda()
    tg(xt)
    g = _tg_gate(tg)
    c0 = g.debug_counter(0)
    g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 2)
    try:
        out = tg(xt)
    finally:
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0)
    assert out.dtype == dtype and tuple(out.shape) == gold["out"].shape
    assert O.rel_err(out.cpu().numpy(), gold["out"]) < TOL
    assert g.debug_counter(0) >= c0          # (the counter only grows; the row gate really ran: see the next assert)
    assert g.debug_field(3).shape[1] == gold["out"].shape[1] // 256 + 1


def test_rowgate_backward_uses_the_same_mask(nr):
    """forward + backward with the row gate's float mask (natural bin order, for the adjoint kernel) against the
    four-kernel path: same gradient."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.torchgate import TorchGate
    x = _rg_inputs()["noise+tone 24x16000"][:6].cuda()
    tg = TorchGate(sr=16000).cuda()
    tg(x)
    g = _tg_gate(tg)
    w = torch.linspace(0.5, 1.5, 15872, device="cuda")
    grads = []
    for mode in (2, 1):
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, mode)
        xg = x.clone().requires_grad_()
        (tg(xg) * w).sum().backward()
        grads.append(xg.grad.clone())
    g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0)
    assert O.rel_err(grads[0].cpu().numpy(), grads[1].cpu().numpy()) < 1e-6


def test_rowgate_is_the_default_for_large_batches(nr):
    """256 x 16000 (BASELINE configs[4]) takes the row gate by itself; 8 rows take the four-kernel path."""
    from noisereduce_amd.torchgate import TorchGate
    x = _rg_inputs()["noise+tone 24x16000"]
    tg = TorchGate(sr=16000).cuda()
    big = x.repeat(11, 1)[:256].contiguous().cuda()
    tg(big)
    g = _tg_gate(tg)
    prof = lambda xx: (g.profile_read(reset=True), g.profile_enable(True), tg(xx), g.profile_read(reset=True), g.profile_enable(False))[3]
    assert any("k_row_gate" in k for k in prof(big))
    assert not any("k_row_gate" in k for k in prof(x[:8].contiguous().cuda()))
