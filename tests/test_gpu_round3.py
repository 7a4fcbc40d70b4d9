"""Round-3 GPU tests (VERDICT r2 "Next round" item 1):

* the fused apply kernel's in-launch hand-off (`k_apply_fast<LEAN>`: configs[2] non-stationary, TorchGate forward
  and backward, every float-mask path) is placement independent (tile = ticket) -- driven under uneven load from
  two host threads on two streams, every output word checked;
* a lost hand-off is reported for the call that suffered it (`sg_check_errors`), and calls that return host arrays
  are re-run on the kernels without in-launch hand-offs.
"""
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


def test_lean_apply_handoff_under_uneven_load(nr):
    """Two host threads, two streams: a long 6-channel NON-STATIONARY recording (k_apply_fast<float mask, LEAN>: 58
    tiles per unit, every tile waits for the partial hops of the tile one ticket earlier) against many short
    TorchGate forward + backward calls (k_apply_fast<K mask> and the adjoint: 4-5 tiles per row, half of them at a
    row edge).  Every output word must equal the result of the same call run alone (MI355X_MICROARCH.md: test every
    hand-off under uneven load), which in turn matches the oracle."""
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    from noisereduce_amd.torchgate import TorchGate
    big = np.stack([O.synth_signal(900000, seed=10 + c, tone_hz=300.0 * (c + 1)) for c in range(6)]).astype(np.float32)
    sb = SpectralGateNonStationary(y=torch.from_numpy(big).cuda(), **NS_KW)
    ref_b = sb.get_traces().clone()
    want = O.reduce_noise_S(big[:2, :250000].astype(np.float64), 48000, stationary=False, chunk_size=100000, padding=8000)
    got = SpectralGateNonStationary(y=torch.from_numpy(big[:2, :250000].copy()).cuda(), **NS_KW).get_traces()
    assert O.rel_err(got.cpu().numpy(), want) < TOL
    assert torch.equal(got[:, :200000], ref_b[:2, :200000])   # chunks 0, 1 do not see the shorter recording's end

    x = torch.from_numpy(np.stack([O.synth_signal(16000, sr=16000, seed=s, tone_hz=440.0) for s in range(24)])).cuda()
    tg = TorchGate(sr=16000).cuda()
    xg = x.clone().requires_grad_()
    y0 = tg(xg)
    w = torch.linspace(0.5, 1.5, y0.shape[1], device="cuda")
    (y0 * w).sum().backward()
    ref_y, ref_g = y0.detach().clone(), xg.grad.clone()
    wantT = O.torchgate_T(x.cpu().numpy().astype(np.float64), 16000, window=torch.hann_window(1024).double().numpy())
    assert O.rel_err(ref_y.cpu().numpy(), wantT) < TOL
    bad = []

    def run_big(stream):
        with torch.cuda.stream(stream):
            for _ in range(12):
                out = sb.get_traces()
                if not torch.equal(out, ref_b):
                    bad.append(("nonstationary", float((out - ref_b).abs().max())))
        stream.synchronize()

    def run_small(stream):
        with torch.cuda.stream(stream):
            xs = x.clone().requires_grad_()
            for _ in range(200):
                xs.grad = None
                y = tg(xs)
                (y * w).sum().backward()
                if not torch.equal(y.detach(), ref_y):
                    bad.append(("torchgate fwd", float((y.detach() - ref_y).abs().max())))
                if not torch.equal(xs.grad, ref_g):
                    bad.append(("torchgate bwd", float((xs.grad - ref_g).abs().max())))
        stream.synchronize()

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    th = [threading.Thread(target=run_big, args=(s1,)), threading.Thread(target=run_small, args=(s2,))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad, bad[:5]
    sb._gate.check_errors()
    for g in tg._gates.values():
        g.check_errors()


def test_lean_apply_handoff_equals_seam_kernel(nr):
    """The in-launch hand-off adds the same two partial sums in the same order as `k_ola_seam`: bit-identical output
    (SG_OPT_FORCE_NOSEAM keeps the variant without any hand-off for comparison: same hops from overlapping tiles)."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    y = np.stack([O.synth_signal(330000, seed=3 + c) for c in range(2)]).astype(np.float32)
    sg = SpectralGateNonStationary(y=torch.from_numpy(y).cuda(), **NS_KW)
    a = sg.get_traces().clone()
    sg._gate.set_option(_ffi.SG_OPT_FORCE_NOLEAN, 1)
    try:
        b = sg.get_traces().clone()
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOLEAN, 0)
    assert O.rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6    # other kernel variant: same sums, other order
    want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=False, chunk_size=100000, padding=8000)
    assert O.rel_err(a.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("stationary", [True, False])
def test_lost_handoff_is_reported_and_rerun(nr, stationary):
    """A launch that loses a hand-off (injected: SG_OPT_INJECT_HANDOFF_FAULT) is reported by sg_check_errors for THAT
    call; reduce_noise with host arrays re-runs it on the kernels without in-launch hand-offs and returns the right
    result; the handle is clean afterwards."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = O.synth_signal(260000, seed=5).astype(np.float32)
    kw = dict(NS_KW)
    if stationary:
        for k in ("thresh_n_mult_nonstationary", "sigmoid_slope_nonstationary"):
            kw.pop(k)
        kw.update(y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True)
        mk = lambda yy: SpectralGateStationary(y=yy, **kw)
    else:
        mk = lambda yy: SpectralGateNonStationary(y=yy, **kw)
    want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=stationary, chunk_size=100000, padding=8000)
    # (a) device tensors: asynchronous, the caller checks
    sg = mk(torch.from_numpy(y).cuda())
    good = sg.get_traces().clone()
    sg._gate.check_errors()
    sg._gate.set_option(_ffi.SG_OPT_INJECT_HANDOFF_FAULT, 2 if stationary else 4)
    sg.get_traces()
    with pytest.raises(_ffi.HandoffTimeout):
        sg._gate.check_errors()
    sg._gate.check_errors()                       # reported once, then clean
    assert torch.equal(sg.get_traces(), good)
    # (b) host arrays: checked and re-run inside the call
    sh = mk(y)
    sh._gate.set_option(_ffi.SG_OPT_INJECT_HANDOFF_FAULT, 1 if stationary else 4)
    out = sh.get_traces()
    assert O.rel_err(out, want) < TOL
    sh._gate.check_errors()
    assert O.rel_err(sh.get_traces(), want) < TOL
    # (c) a caller that never checks learns about it at the next call on the handle
    sg._gate.set_option(_ffi.SG_OPT_INJECT_HANDOFF_FAULT, 4 if not stationary else 2)
    sg.get_traces()
    torch.cuda.synchronize()
    with pytest.raises(_ffi.HandoffTimeout):
        sg.get_traces()
    assert torch.equal(sg.get_traces(), good)


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


# ---------------------------------------------------------------------------------------------
# long frames: the four-step transform through HBM (big.hpp)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_fft,kw", [
    (4097, dict()),                                            # first length beyond the per-workgroup chirp-z kernels (M = 16384)
    (8193, dict(time_mask_smooth_ms=100)),                     # M = 32768
    (16384, dict(time_mask_smooth_ms=200)),                    # power of two, direct (M = n)
    (20000, dict(time_mask_smooth_ms=300, win_length=16000, hop_length=3000)),   # M = 65536, window shorter than the frame
    (32768, dict(time_mask_smooth_ms=400)),
    (65536, dict(time_mask_smooth_ms=800, freq_mask_smooth_hz=None)),           # the largest frame
])
@pytest.mark.parametrize("stationary", [True, False])
def test_long_frames_match_the_oracle(nr, n_fft, kw, stationary):
    """reduce_noise with n_fft beyond 8192 (or beyond 4096 and not a power of two) against the oracle (base.py:77-86
    accepts any n_fft; scipy.signal.stft/istft use rfft(n)/irfft(n))."""
    n = max(6 * n_fft, 90000)
    y = O.synth_signal(n, seed=n_fft).astype(np.float32)
    args = dict(stationary=stationary, n_fft=n_fft, chunk_size=max(50000, 3 * n_fft), padding=max(6000, n_fft), **kw)
    got = nr.reduce_noise(y=y, sr=48000, **args)
    want = O.reduce_noise_S(y.astype(np.float64), 48000, **args)
    assert got.shape == y.shape and got.dtype == y.dtype
    assert O.rel_err(got, want) < TOL


def test_long_frames_stft_tap(nr):
    """The STFT tap (sg_stft) on a long chirp-z frame and a long power-of-two frame against scipy-style STFT of the oracle."""
    from noisereduce_amd import _ffi
    for n_fft in (5000, 16384):
        x = O.synth_signal(4 * n_fft + 123, seed=3).astype(np.float64)
        g = _ffi.Gate("cuda", variant=_ffi.SG_VARIANT_S, stationary=True, n_fft=n_fft, win_length=n_fft, hop_length=n_fft // 4)
        Z = g.stft(torch.from_numpy(x)[None].cuda())[0].cpu().numpy().T     # (F, T)
        Zo = O.stft_scipy(x, n_fft, n_fft, n_fft // 4)
        assert Z.shape == Zo.shape
        assert np.max(np.abs(Z - Zo)) < 1e-12 * max(1.0, np.max(np.abs(Zo)))
        g.close()


def test_torchgate_long_frames(nr):
    from noisereduce_amd.torchgate import TorchGate
    for n_fft, kw in ((16384, dict(time_mask_smooth_ms=200)), (6000, dict(nonstationary=True, n_movemean_nonstationary=5))):
        x = np.stack([O.synth_signal(3 * n_fft + 777, sr=48000, seed=s) for s in range(3)]).astype(np.float64)
        tg = TorchGate(sr=48000, n_fft=n_fft, **kw).cuda()
        got = tg(torch.from_numpy(x).cuda()).cpu().numpy()
        want = O.torchgate_T(x, 48000, n_fft=n_fft, window=torch.hann_window(n_fft).double().numpy(), **kw)
        assert got.shape == want.shape
        assert O.rel_err(got, want) < TOL


# ---------------------------------------------------------------------------------------------
# integer recordings: the truncated float64 result of the reference, bit for bit (exact.hpp)
# ---------------------------------------------------------------------------------------------
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


# ---------------------------------------------------------------------------------------------
# n_fft = 512 / hop 128 on the register transform (fast512.hpp): two real frames per complex transform
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sr,n,kw", [
    (16000, 30000, dict()),                                                  # one chunk
    (48000, 200000, dict(chunk_size=40000, padding=5000)),                   # chunk grid, partial last chunk
    (16000, 51234, dict(chunk_size=9000, padding=1000, prop_decrease=0.6)),  # ragged: tiles at both unit edges
    (16000, 515, dict()),                                                    # barely longer than a frame
    (8000, 20000, dict(freq_mask_smooth_hz=None, time_mask_smooth_ms=None)),
    (44100, 70000, dict(time_mask_smooth_ms=None)),
])
@pytest.mark.parametrize("stationary", [True, False])
def test_nfft512_fast_path_matches_the_oracle(nr, sr, n, kw, stationary):
    y = np.stack([O.synth_signal(n, sr=sr, seed=81 + c, tone_hz=300.0 * (c + 1)) for c in range(2)]).astype(np.float32)
    args = dict(stationary=stationary, n_fft=512, **kw)
    got = nr.reduce_noise(y=y, sr=sr, **args)
    want = O.reduce_noise_S(y.astype(np.float64), sr, **args)
    assert O.rel_err(got, want) < TOL
    # the general LDS kernels on the same input (SG_OPT_FORCE_NOFAST): same result to float32 rounding
    y1 = torch.from_numpy(y).cuda()
    a = nr.reduce_noise(y=y1, sr=sr, **args)
    assert O.rel_err(a.cpu().numpy(), want) < TOL


def test_nfft512_decisions_equal_the_float64_decisions(nr):
    """Mask bits of k_decide_fast512 (float32 + exact refinement, two frames per transform) == the all-float64 decision
    kernel, bit for bit -- incl. a loud frame next to a quiet one (the pair shares one transform) and a steady tone."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    sr, n = 16000, 120000
    y = O.synth_signal(n, sr=sr, seed=5, tone_hz=440.0).astype(np.float32)
    y[30000:30700] *= 200.0          # a burst: frames with a loud and a quiet partner
    y[60000:] = (0.3 * np.sin(2 * np.pi * 1000.0 * np.arange(n - 60000) / sr)).astype(np.float32)   # steady tone
    kw = dict(sr=sr, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=50000, clip_noise_stationary=True,
              padding=4000, n_fft=512, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
              time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
    out_fast = sg.get_traces().clone()
    bits_fast = sg._gate.debug_field(3)
    sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 1)
    try:
        out_64 = sg.get_traces().clone()
        bits_64 = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 0)
    assert bits_fast.shape == bits_64.shape and np.array_equal(bits_fast, bits_64)
    assert torch.equal(out_fast, out_64)
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=512, chunk_size=50000, padding=4000)
    assert O.rel_err(out_fast.cpu().numpy(), want) < TOL


def test_torchgate_nfft512(nr):
    from noisereduce_amd.torchgate import TorchGate
    for kw in (dict(), dict(nonstationary=True)):
        x = np.stack([O.synth_signal(16000, sr=16000, seed=s, tone_hz=440.0) for s in range(5)]).astype(np.float64)
        tg = TorchGate(sr=16000, n_fft=512, **kw).cuda()
        xt = torch.from_numpy(x).cuda().requires_grad_()
        y = tg(xt)
        want = O.torchgate_T(x, 16000, n_fft=512, window=torch.hann_window(512).double().numpy(), **kw)
        assert O.rel_err(y.detach().cpu().numpy(), want) < TOL
        # backward: the adjoint with the mask fixed against autograd through torch.stft / istft on the CPU
        w = torch.linspace(0.5, 1.5, y.shape[1], dtype=torch.float64)
        (y * w.cuda()).sum().backward()
        got_g = xt.grad.cpu()
        _, st = O.torchgate_T(x, 16000, n_fft=512, window=torch.hann_window(512).double().numpy(), return_stages=True, **kw)
        m = torch.from_numpy(st["mask"])
        xc = torch.from_numpy(x).requires_grad_()
        win = torch.hann_window(512, dtype=torch.float64)
        X = torch.stft(xc, 512, 128, 512, window=win, center=True, pad_mode="constant", return_complex=True)
        yc = torch.istft(X * m, 512, 128, 512, window=win, center=True)
        (yc * w).sum().backward()
        assert O.rel_err(got_g.numpy(), xc.grad.numpy()) < TOL


# ---------------------------------------------------------------------------------------------
# n_fft = 2048 / hop 512 on the register transform (fast2048.hpp): 1024 complex points on 32 lanes
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sr,n,kw", [
    (44100, 50000, dict()),                                                   # one chunk
    (48000, 300000, dict(chunk_size=70000, padding=9000)),                    # chunk grid, partial last chunk
    (48000, 123457, dict(chunk_size=30000, padding=4100, prop_decrease=0.6)), # ragged: tiles at both unit edges
    (48000, 2060, dict()),                                                    # barely longer than a frame
    (96000, 90000, dict(freq_mask_smooth_hz=None, time_mask_smooth_ms=None)),
])
@pytest.mark.parametrize("stationary", [True, False])
def test_nfft2048_fast_path_matches_the_oracle(nr, sr, n, kw, stationary):
    y = np.stack([O.synth_signal(n, sr=sr, seed=91 + c, tone_hz=300.0 * (c + 1)) for c in range(2)]).astype(np.float32)
    args = dict(stationary=stationary, n_fft=2048, **kw)
    got = nr.reduce_noise(y=y, sr=sr, **args)
    want = O.reduce_noise_S(y.astype(np.float64), sr, **args)
    assert O.rel_err(got, want) < TOL


def test_nfft2048_decisions_equal_the_float64_decisions(nr):
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    sr, n = 48000, 400000
    y = O.synth_signal(n, sr=sr, seed=6, tone_hz=440.0).astype(np.float32)
    y[100000:102000] *= 200.0
    y[250000:] = (0.3 * np.sin(2 * np.pi * 1000.0 * np.arange(n - 250000) / sr)).astype(np.float32)   # steady tone
    kw = dict(sr=sr, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=150000, clip_noise_stationary=True,
              padding=12000, n_fft=2048, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
              time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
    out_fast = sg.get_traces().clone()
    bits_fast = sg._gate.debug_field(3)
    sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 1)
    try:
        out_64 = sg.get_traces().clone()
        bits_64 = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 0)
    assert bits_fast.shape == bits_64.shape and np.array_equal(bits_fast, bits_64)
    assert torch.equal(out_fast, out_64)
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=2048, chunk_size=150000, padding=12000)
    assert O.rel_err(out_fast.cpu().numpy(), want) < TOL


def test_torchgate_nfft2048(nr):
    from noisereduce_amd.torchgate import TorchGate
    for kw in (dict(), dict(nonstationary=True)):
        x = np.stack([O.synth_signal(30000, sr=48000, seed=s, tone_hz=440.0) for s in range(3)]).astype(np.float64)
        tg = TorchGate(sr=48000, n_fft=2048, **kw).cuda()
        xt = torch.from_numpy(x).cuda().requires_grad_()
        y = tg(xt)
        want, st = O.torchgate_T(x, 48000, n_fft=2048, window=torch.hann_window(2048).double().numpy(), return_stages=True, **kw)
        assert O.rel_err(y.detach().cpu().numpy(), want) < TOL
        w = torch.linspace(0.5, 1.5, y.shape[1], dtype=torch.float64)
        (y * w.cuda()).sum().backward()
        m = torch.from_numpy(st["mask"])
        xc = torch.from_numpy(x).requires_grad_()
        win = torch.hann_window(2048, dtype=torch.float64)
        X = torch.stft(xc, 2048, 512, 2048, window=win, center=True, pad_mode="constant", return_complex=True)
        yc = torch.istft(X * m, 2048, 512, 2048, window=win, center=True)
        (yc * w).sum().backward()
        assert O.rel_err(xt.grad.cpu().numpy(), xc.grad.numpy()) < TOL


# ---------------------------------------------------------------------------------------------
# one-kernel non-stationary masks (nonstat_mask.hpp): k_iir_mask<NT> (variant S), k_box_mask<NT, 20> (variant T)
# ---------------------------------------------------------------------------------------------
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
