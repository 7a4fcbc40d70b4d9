"""In-launch hand-offs between workgroups (k_gate_onepass, k_apply_fast<LEAN>): equality with the seam kernel, lost hand-offs reported / re-run / poisoned.
(grouped by subject in round 5; the tests themselves date from rounds 2-4)"""
import os
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


@pytest.mark.parametrize("stationary,bits,what", [(True, 8, "one-pass gate: mask bits + partial hops"),
                                                  (False, 32, "fused-apply partial hops")])
def test_lost_handoff_poisons_the_output(nr, stationary, bits, what):
    """VERDICT r3 item 7.  The kernel's own timeout path (every poll of the next launch is treated as lost): a
    device-tensor caller that NEVER calls check_errors receives NaN in the hops the tile could not finalise --
    never a plausible partial sum -- and the error word is set by the kernel itself.  The next call is clean."""
    from noisereduce_amd import _ffi
    y = O.synth_signal(260000, seed=5).astype(np.float32)
    sg = _gate_S(stationary, torch.from_numpy(y).cuda())
    good = sg.get_traces().clone()
    sg._gate.check_errors()
    assert torch.isfinite(good).all()
    sg._gate.set_option(_ffi.SG_OPT_INJECT_HANDOFF_FAULT, bits)
    bad = sg.get_traces().clone()          # no check_errors between the call and the use of its result
    torch.cuda.synchronize()
    n_nan = int(torch.isnan(bad).sum())
    assert n_nan > 0, what
    if bits == 8:
        # every tile with a neighbour lost its mask: (almost) nothing of the recording survives
        assert n_nan > 0.9 * bad.numel()
    else:
        # only the three hops that straddle two tiles (3 of 16) are lost; every finite sample is the right one
        assert 0.1 * bad.numel() < n_nan < 0.3 * bad.numel()
        ok = ~torch.isnan(bad)
        assert torch.equal(bad[ok], good[ok])
    with pytest.raises(_ffi.HandoffTimeout):
        sg._gate.check_errors()
    sg._gate.check_errors()
    assert torch.equal(sg.get_traces(), good)


@pytest.mark.parametrize("n_fft", [256, 512, 2048])
def test_lost_handoff_poisons_the_small_onepass_gates(nr, n_fft):
    """The same for k_gate_onepass256 / 512 / 2048 (round 6): a tile whose neighbours' bits never arrive poisons every hop it
    touches (2048: complete hops AND the partial sums k_ola_seam2048 combines), sets the error word itself, and the next
    call on the handle is clean."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = O.synth_signal(300000, sr=48000, seed=9).astype(np.float32)
    kw = dict(NS_KW)
    for k in ("thresh_n_mult_nonstationary", "sigmoid_slope_nonstationary"):
        kw.pop(k)
    kw.update(y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, n_fft=n_fft)
    sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
    good = sg.get_traces().clone()
    sg._gate.check_errors()
    assert torch.isfinite(good).all()
    sg._gate.set_option(_ffi.SG_OPT_INJECT_HANDOFF_FAULT, 8)
    bad = sg.get_traces().clone()
    torch.cuda.synchronize()
    assert int(torch.isnan(bad).sum()) > 0.9 * bad.numel()
    with pytest.raises(_ffi.HandoffTimeout):
        sg._gate.check_errors()
    sg._gate.check_errors()
    assert torch.equal(sg.get_traces(), good)


def test_lost_handoff_poisons_torchgate_forward(nr):
    """The same for TorchGate.forward in a training loop (device tensors, asynchronous): NaN, not garbage."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.torchgate import TorchGate
    x = torch.from_numpy(np.stack([O.synth_signal(16000, sr=16000, seed=s, tone_hz=440.0) for s in range(8)])).cuda()
    tg = TorchGate(sr=16000).cuda()
    good = tg(x).clone()
    assert torch.isfinite(good).all()
    (gate,) = list(tg._gates.values())
    gate.check_errors()
    gate.set_option(_ffi.SG_OPT_INJECT_HANDOFF_FAULT, 32)
    bad = tg(x).clone()
    torch.cuda.synchronize()
    if not bool(torch.isnan(bad).any()):
        # the forward ran on a path without in-launch hand-offs (one kernel per row): nothing to lose, the option
        # stays armed for the next hand-off launch -- disarm it
        gate.set_option(_ffi.SG_OPT_INJECT_HANDOFF_FAULT, 0)
        assert torch.equal(bad, good)
        return
    ok = ~torch.isnan(bad)
    assert torch.equal(bad[ok], good[ok])
    with pytest.raises(_ffi.HandoffTimeout):
        gate.check_errors()
    assert torch.equal(tg(x), good)


@pytest.mark.parametrize("n_fft", [1024, 512, 2048, -1024])
def test_gates_on_three_streams_at_once(n_fft):
    """tests/tools/soak_handoff.py for a few seconds: a stationary gate, a non-stationary gate and TorchGate forward + backward
    on three host threads / HIP streams; every result equals the same call run alone, bit for bit, no hand-off is lost.  (Round 6:
    this is the test the persistent-workgroup form of k_gate_onepass first failed next to TorchGate's one-workgroup-per-CU row
    gate -- see test_persistent_gate_next_to_another_streams_kernels; it is the default kernel at n_fft = 1024 again.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, N_FFT=str(abs(n_fft)))
    env.pop("TILE_ORDER", None)
    if n_fft < 0:
        env["TILE_ORDER"] = "2"   # (-1024: the one-tile-per-workgroup form of the n_fft = 1024 gate, what serving mode runs)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "soak_handoff.py"), "4"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches: none" in r.stdout, (r.stdout[-400:], r.stderr[-400:])


@pytest.mark.parametrize("other", ["torchgate", "matmul"])
def test_persistent_gate_next_to_another_streams_kernels(other):
    """The persistent-workgroup form of k_gate_onepass (SG_OPT_TILE_ORDER 0) next to TorchGate forward + backward / a plain matmul
    loop on a second stream: every output equals the one-tile kernel's, no hand-off is lost.  Round 6: this failed within a
    second -- a missing wait for an LDS store before the barrier at the loop head, on the halo tiles' path (onepass.hpp,
    DESIGN.md section 3; tests/test_isa_audit.py guards the ISA) -- and is the test that found it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, N_FFT="1024", TILE_ORDER="0", THREADS="s,t")
    env.pop("T_MODE", None)
    if other == "matmul":
        env["T_MODE"] = "matmul"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "soak_handoff.py"), "8"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "mismatches: none" in r.stdout, (r.stdout[-400:], r.stderr[-400:])
