"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors from the
live reference and against the CPU oracle.  Tolerance: max|y - y_ref| / max|y_ref| <= 1e-4
(BASELINE.json north_star); the stationary masks must match bit-for-bit on these cases."""
import os

import numpy as np
import pytest
import torch

from oracle import spectralgate_oracle as O
from tests.golden.cases import S_CASES, T_CASES, make_input_S, make_input_T

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.fixture(scope="module")
def nr():
    import noisereduce_amd
    return noisereduce_amd


@pytest.mark.parametrize("name", sorted(S_CASES))
def test_reduce_noise_golden(nr, golden_dir, name):
    case = S_CASES[name]
    g = _load(golden_dir, "S_" + name)
    y, y_noise = make_input_S(case)
    out = nr.reduce_noise(y=y, sr=case["sr"], y_noise=y_noise, **case["kwargs"])
    assert out.shape == g["out"].shape and out.dtype == g["out"].dtype
    assert O.rel_err(out, g["out"]) < TOL


@pytest.mark.parametrize("name", sorted(S_CASES))
def test_reduce_noise_float32_input(nr, golden_dir, name):
    """float32 in -> float32 out (the BASELINE workload dtype); compared with the reference fed
    float64 copies of the same values."""
    case = S_CASES[name]
    g = _load(golden_dir, "S_" + name)
    y, y_noise = make_input_S(case)
    out = nr.reduce_noise(y=y.astype(np.float32), sr=case["sr"],
                          y_noise=None if y_noise is None else y_noise.astype(np.float32),
                          **case["kwargs"])
    assert out.dtype == np.float32 and out.shape == g["out"].shape
    assert O.rel_err(out, g["out"]) < TOL


def test_stage_taps(nr, golden_dir):
    """STFT values, per-band threshold, raw mask bits and smoothed mask of the single-chunk
    stationary case, each against the reference's own intermediate."""
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    case = S_CASES["stat_1chunk"]
    g = _load(golden_dir, "S_stat_1chunk")
    y, _ = make_input_S(case)
    sg = SpectralGateStationary(
        y=y, sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5,
        chunk_size=600000, clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None,
        hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
        tmp_folder=None, use_tqdm=False, n_jobs=1)
    thr = sg.noise_thresh
    assert np.max(np.abs(thr - g["thresh"])) < 1e-9
    # STFT tap on the padded chunk
    chunk = O.read_chunk(y[None, :], -30000, len(y) + 30000)
    Z = sg._gate.stft(torch.from_numpy(chunk).cuda()).cpu().numpy()[0].T   # (F, T)
    cols = [0, 1, 2, 117, 200, Z.shape[1] - 1]
    assert np.max(np.abs(Z[:, cols] - g["Z_cols"])) < 1e-13
    out = sg.get_traces()
    assert O.rel_err(out, g["out"]) < TOL
    raw = sg._gate.debug_field(3)[0].T                                      # (F, T) bits
    assert raw.shape == tuple(g["raw_shape"])
    ref_raw = np.unpackbits(g["raw_bits"], axis=1)[:, :raw.shape[1]].astype(bool)
    d0, d1 = sg._gate.debug_range()       # frames that reach the kept samples (+- smoothing)
    assert 0 <= d0 < d1 <= raw.shape[1] and d1 - d0 > 150
    assert np.count_nonzero(raw[:, d0:d1] != ref_raw[:, d0:d1]) == 0, "mask flips vs the reference"
    from noisereduce_amd import _ffi
    try:  # the smoothed mask as floats only exists on the general apply path
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 1)
        out2 = sg.get_traces()
        M = sg._gate.debug_field(1)[0].T
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 0)
    assert np.max(np.abs(M[[0, 3, 100, 511, 512], :] - g["smooth_rows"])) < 1e-5
    assert O.rel_err(out2, g["out"]) < TOL


def test_fish_wav_config0(nr, golden_dir):
    """BASELINE.json configs[0] on the GPU: int16 in/out BIT-EXACT against the reference (which truncates a float64
    result, base.py:217-226: integer outputs take the float64 pipeline), float64 stationary and non-stationary."""
    g = _load(golden_dir, "S_fish")
    data, rate = g["data"], int(g["rate"])
    out = nr.reduce_noise(y=data, sr=rate, stationary=True)
    assert out.dtype == np.int16 and out.shape == data.shape
    assert np.array_equal(out, g["out_i16"])
    # the opt-out (NOISEREDUCE_AMD_FAST_INT=1 / SG_OPT_FAST_INTEGER): fused float32 kernels, <= 1 LSB off on ~1 % of the samples
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    import os
    os.environ["NOISEREDUCE_AMD_FAST_INT"] = "1"
    try:
        fast = nr.reduce_noise(y=data, sr=rate, stationary=True)
    finally:
        del os.environ["NOISEREDUCE_AMD_FAST_INT"]
    d = np.abs(fast.astype(np.int32) - g["out_i16"].astype(np.int32))
    assert d.max() <= 1 and np.mean(d > 0) < 0.02
    out64 = nr.reduce_noise(y=data.astype(np.float64), sr=rate, stationary=True)
    assert out64.dtype == np.float64
    assert O.rel_err(out64, g["out_f64"]) < TOL
    out_ns = nr.reduce_noise(y=data.astype(np.float64), sr=rate, stationary=False)
    assert O.rel_err(out_ns, g["out_ns"]) < TOL


def test_do_filter_seam(nr):
    """SpectralGate._do_filter(chunk): float64 (C, Lp) in, same shape out, zero tail."""
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    y = np.stack([O.synth_signal(30000, seed=3).astype(np.float64),
                  O.synth_signal(30000, seed=4, tone_hz=333.0).astype(np.float64)])
    sg = SpectralGateNonStationary(
        y=y, sr=48000, chunk_size=600000, padding=30000, n_fft=1024, win_length=None,
        hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
        thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None,
        prop_decrease=1.0, use_tqdm=False, n_jobs=1)
    chunk = sg._read_chunk(-1000, 30000 + 1234)
    got = sg._do_filter(chunk)
    want = O.gate_nonstationary_S(chunk, 1024, 1024, 256, 1.0, O.smoothing_filter(5, 9),
                                  O.iir_coefficient(2.0, 48000, 256), 2, 10)
    assert got.shape == chunk.shape and got.dtype == np.float64
    assert O.rel_err(got, want) < TOL
    Lout = (chunk.shape[1] // 256) * 256
    assert np.all(got[:, Lout:] == 0)


def test_get_traces_subrange(nr):
    """get_traces(start_frame, end_frame) on a persistent object (base.py:167-172)."""
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
    y = O.synth_signal(100000, seed=9).astype(np.float64)
    kw = dict(sr=48000, chunk_size=20000, padding=3000, n_fft=1024, win_length=None, hop_length=None,
              time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
              thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None,
              prop_decrease=1.0, use_tqdm=False, n_jobs=1)
    sg = SpectralGateNonStationary(y=y, **kw)
    full = O.reduce_noise_S(y, 48000, stationary=False, chunk_size=20000, padding=3000)
    got = sg.get_traces(start_frame=25000, end_frame=77000)
    assert got.shape == (52000,)
    assert O.rel_err(got, full[25000:77000]) < TOL


def test_tensor_io_stays_on_device(nr):
    y = torch.from_numpy(O.synth_signal(70000, seed=2)).cuda()
    out = nr.reduce_noise(y=y, sr=48000, stationary=True, chunk_size=30000, padding=4000)
    assert isinstance(out, torch.Tensor) and out.is_cuda and out.dtype == torch.float32
    want = O.reduce_noise_S(y.cpu().numpy().astype(np.float64), 48000, stationary=True,
                            chunk_size=30000, padding=4000)
    assert O.rel_err(out.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("name", sorted(T_CASES))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_torchgate_golden(golden_dir, name, dtype):
    from noisereduce_amd.torchgate import TorchGate
    case = T_CASES[name]
    g = _load(golden_dir, "T_" + name)
    x, xn = make_input_T(case)
    tg = TorchGate(sr=case["sr"], **case["kwargs"]).cuda()
    xt = torch.from_numpy(x).to(dtype).cuda()
    xnt = None if xn is None else torch.from_numpy(xn).to(dtype).cuda()
    out = tg(xt, xnt)
    assert out.dtype == dtype and tuple(out.shape) == g["out"].shape
    assert O.rel_err(out.cpu().numpy(), g["out"]) < TOL


def test_torchgate_state_dict_and_errors():
    from noisereduce_amd.torchgate import TorchGate
    tg = TorchGate(sr=16000)
    sd = tg.state_dict()
    assert list(sd) == ["smoothing_filter"] and tuple(sd["smoothing_filter"].shape) == (1, 1, 33, 7)
    assert abs(float(sd["smoothing_filter"].sum()) - 1) < 1e-6
    assert sum(p.numel() for p in tg.parameters()) == 0
    with pytest.raises(Exception):
        tg(torch.zeros(2, 1000).cuda())
    with pytest.raises(AssertionError):
        tg(torch.zeros(4000).cuda())


def test_unfused_path_still_matches(nr, golden_dir):
    """The materialised (v1) kernels stay available behind SG_OPT_FORCE_UNFUSED and must
    give the same answer as the fused bit-mask path."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    case = S_CASES["stat_chunked"]
    g = _load(golden_dir, "S_stat_chunked")
    y, _ = make_input_S(case)
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5,
              chunk_size=25000, clip_noise_stationary=True, padding=4000, n_fft=1024, win_length=None,
              hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
              tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=y, **kw)
    fast = sg.get_traces()
    try:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 1)
        fused = sg.get_traces()
        sg._gate.set_option(_ffi.SG_OPT_FORCE_UNFUSED, 1)
        unfused = sg.get_traces()
        raw = sg._gate.debug_field(0)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_UNFUSED, 0)
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 0)
    assert raw.dtype == np.float32
    for o in (fast, fused, unfused):
        assert O.rel_err(o, g["out"]) < TOL
    assert O.rel_err(fused, unfused) < 1e-6 and O.rel_err(fast, unfused) < 1e-5


@pytest.mark.parametrize("prop", [1.0, 0.7])
def test_db_floor_live(nr, prop):
    """A band whose peak sits more than 80 dB above the noise threshold has its whole row lifted
    by _amp_to_db's floor (spectralgate/utils.py:16): digital silence next to loud noise, with a
    very quiet noise clip.  Exercises the floor pre-pass of the fused path."""
    rng = np.random.default_rng(77)
    n = 60000
    y = np.zeros(n)
    y[n // 2:] = 0.5 * rng.standard_normal(n // 2)
    y = y.astype(np.float32).astype(np.float64)
    y_noise = (1e-7 * rng.standard_normal(20000)).astype(np.float32).astype(np.float64)
    kw = dict(stationary=True, y_noise=y_noise, prop_decrease=prop, chunk_size=25000, padding=4000)
    want = O.reduce_noise_S(y, 48000, **kw)
    got = nr.reduce_noise(y=y, sr=48000, **kw)
    assert O.rel_err(got, want) < TOL
    # the same input with the floor out of reach (loud noise clip) for contrast
    y_noise2 = (0.3 * rng.standard_normal(20000)).astype(np.float32).astype(np.float64)
    kw["y_noise"] = y_noise2
    assert O.rel_err(nr.reduce_noise(y=y, sr=48000, **kw), O.reduce_noise_S(y, 48000, **kw)) < TOL


def test_silence_and_constant_inputs(nr):
    """Degenerate inputs: all-zero signal (every dB equals the threshold) and a short burst."""
    z = np.zeros(30000)
    out = nr.reduce_noise(y=z, sr=48000, stationary=True)
    assert np.all(out == 0)
    out = nr.reduce_noise(y=z, sr=48000, stationary=False)
    assert out.shape == z.shape


@pytest.mark.parametrize("n_fft", [1024, 512, 2048, 256])
@pytest.mark.parametrize("kind", ["noise_tone", "pure_tone", "steps"])
def test_fast_decide_bits_equal_f64_decide(nr, kind, n_fft):
    """The float32 + exact-refine decision kernel must produce the SAME mask bits as the
    float64 STFT decision, including on inputs built to sit on the threshold (a steady tone:
    every cell of the tone bands has dB ~= mean = threshold)."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    n = 120000
    t = np.arange(n) / 48000.0
    rng = np.random.default_rng(5)
    if kind == "noise_tone":
        y = O.synth_signal(n, seed=8).astype(np.float64)
    elif kind == "pure_tone":
        y = 0.5 * np.sin(2 * np.pi * 1000.0 * t) + 0.25 * np.sin(2 * np.pi * 5250.0 * t) \
            + 1e-5 * rng.standard_normal(n)
    else:
        y = np.where((np.arange(n) // 7000) % 2 == 0, 0.0, 1.0) * (0.3 * rng.standard_normal(n))
    y = y.astype(np.float32)
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5,
              chunk_size=50000, clip_noise_stationary=True, padding=6000, n_fft=n_fft, win_length=None,
              hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
              tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=y, **kw)
    out_fast = sg.get_traces()
    bits_fast = sg._gate.debug_field(3)
    d0, d1 = sg._gate.debug_range()
    try:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 1)
        out_f64 = sg.get_traces()
        bits_f64 = sg._gate.debug_field(3)
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 0)
    assert bits_fast.shape == bits_f64.shape and d1 - d0 > 100
    assert np.count_nonzero(bits_fast[:, d0:d1] != bits_f64[:, d0:d1]) == 0
    assert np.array_equal(out_fast, out_f64)
    want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=True, chunk_size=50000, padding=6000,
                            n_fft=n_fft)
    assert O.rel_err(out_fast, want) < TOL


@pytest.mark.parametrize("kw", [dict(), dict(nonstationary=True), dict(n_fft=512, win_length=400, hop_length=100),
                                dict(n_fft=400), dict(n_fft=601, nonstationary=True)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_torchgate_backward_matches_autograd(kw, dtype):
    """TorchGate is differentiable w.r.t. x with the mask detached (torchgate.py:126,167).  Our
    adjoint kernels against torch autograd through stft -> (x mask) -> istft with the same mask."""
    from noisereduce_amd.torchgate import TorchGate
    torch.manual_seed(3)
    B, L = 3, 7000
    x = (0.1 * torch.randn(B, L, dtype=torch.float64)
         + 0.5 * torch.sin(2 * np.pi * 440 * torch.arange(L) / 16000)).to(dtype).cuda().requires_grad_()
    tg = TorchGate(sr=16000, **kw).cuda()
    y = tg(x)
    assert y.requires_grad
    gy = torch.randn_like(y)
    y.backward(gy)
    gx = x.grad.detach().double()
    # same mask, torch's own stft/istft in float64
    from noisereduce_amd import _ffi
    gate = tg._gate_for(x.device)
    try:  # natural bin order (the fused apply kernel keeps its masks in lane order)
        gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 1)
        _, mask = gate.process_batch(x.detach(), None, save_mask=True)
    finally:
        gate.set_option(_ffi.SG_OPT_FORCE_NOFAST, 0)
    n, W, H = tg.n_fft, tg.win_length, tg.hop_length
    M = mask[:, :, :n // 2 + 1].permute(0, 2, 1).double()
    w = torch.hann_window(W).double().cuda()
    x2 = x.detach().double().clone().requires_grad_()
    X = torch.stft(x2, n, H, W, window=w, center=True, pad_mode="constant", return_complex=True)
    y2 = torch.istft(X * M, n, H, W, window=w, center=True)
    assert y2.shape == y.shape
    assert float((y2.detach() - y.detach().double()).abs().max() / y2.detach().abs().max()) < TOL
    y2.backward(gy.double())
    ref = x2.grad
    assert float((gx - ref).abs().max() / ref.abs().max()) < TOL
    # and the adjoint identity <gy, J v> == <J^T gy, v> holds for the engine's own forward
    assert x.grad.dtype == dtype and x.grad.shape == x.shape


@pytest.mark.parametrize("n_fft,nf,nt", [(1024, 5, 9), (512, 2, 18), (256, 1, 37), (2048, 10, 4)])
def test_unit_batching_is_invisible(nr, n_fft, nf, nt):
    """A workspace budget that forces the (channel, chunk) units through several batches must give
    bit-identical output (config 4 has 9216 units; the workspace is bounded).  Every geometry with a one-pass gate:
    each batch is a launch of its own with its own tickets, epoch and exchange buffers."""
    from noisereduce_amd import _ffi
    C, n = 6, 130000
    y = np.stack([O.synth_signal(n, seed=40 + c, tone_hz=200.0 * (c + 1)) for c in range(C)])
    yd = torch.from_numpy(y).cuda()
    kw = dict(variant=_ffi.SG_VARIANT_S, stationary=True, n_fft=n_fft, win_length=n_fft, hop_length=n_fft // 4,
              n_grad_freq=nf, n_grad_time=nt, smooth_mask=True, chunk_size=20000, padding=3000,
              n_std_thresh=1.5, top_db=80.0, ddof=0)
    big = _ffi.Gate("cuda", **kw)
    small = _ffi.Gate("cuda", max_workspace_bytes=3 << 20, **kw)   # a couple of units per batch
    outs = []
    for gate in (big, small):
        gate.noise_stats(yd[:, :20000])
        outs.append(gate.process_chunks(yd, chunked=True).cpu().numpy())
        gate.check_errors()
    assert np.array_equal(outs[0], outs[1])
    want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=True, chunk_size=20000, padding=3000, n_fft=n_fft)
    assert O.rel_err(outs[0], want) < TOL
    big.close(); small.close()


@pytest.mark.parametrize("dtype", [np.int32, np.float64, np.float16, np.int64])
def test_input_dtypes(nr, dtype):
    """Any real dtype in -> same dtype out (base.py:217-226): native device types, and types that
    go through float64 on the host like the reference's chunk copy."""
    y = O.synth_signal(50000, seed=21).astype(np.float64)
    yi = (y * 20000).astype(dtype) if np.issubdtype(dtype, np.integer) else y.astype(dtype)
    out = nr.reduce_noise(y=yi, sr=48000, stationary=True)
    assert out.dtype == dtype and out.shape == yi.shape
    want = O.reduce_noise_S(yi, 48000, stationary=True)
    if np.issubdtype(dtype, np.integer):
        assert np.abs(out.astype(np.int64) - want.astype(np.int64)).max() <= 1
    else:
        tol = 2e-3 if dtype == np.float16 else TOL
        assert O.rel_err(out, want) < tol


def test_short_and_ragged_inputs(nr):
    """Shortest legal input, lengths that are not a multiple of the hop, a last chunk of 1 sample."""
    for n in (1024, 1025, 1279, 5000, 25001):
        y = O.synth_signal(n, seed=n).astype(np.float64)
        kw = dict(stationary=True, chunk_size=25000, padding=2000)
        assert O.rel_err(nr.reduce_noise(y=y, sr=48000, **kw), O.reduce_noise_S(y, 48000, **kw)) < TOL
        kw["stationary"] = False
        assert O.rel_err(nr.reduce_noise(y=y, sr=48000, **kw), O.reduce_noise_S(y, 48000, **kw)) < TOL
    with pytest.raises(ValueError):
        nr.reduce_noise(y=np.zeros(500), sr=48000, stationary=True)      # shorter than win_length
    for n_fft in (40000, 131072):   # beyond the long-frame kernels: any length <= 32768, powers of two <= 65536
        with pytest.raises(NotImplementedError):
            nr.reduce_noise(y=np.zeros(300000), sr=48000, stationary=True, n_fft=n_fft, time_mask_smooth_ms=2000)


def test_use_torch_routing(nr):
    """reduce_noise(use_torch=True): StreamedTorchGate parameter mapping and chunk loop
    (streamed_torch_gate.py:66-87), against the torchgate oracle applied per chunk."""
    y = O.synth_signal(70000, seed=77).astype(np.float64)
    got = nr.reduce_noise(y=y, sr=48000, stationary=True, use_torch=True, chunk_size=30000, padding=4000)
    assert got.shape == y.shape and got.dtype == y.dtype
    # oracle: the reference's chunk loop around TorchGate (float64), xn=None
    w = torch.hann_window(1024).double().numpy()
    want = np.zeros_like(y)
    for ich in range(3):
        s0, e0 = ich * 30000, min((ich + 1) * 30000, 70000)
        chunk = O.read_chunk(y[None, :], s0 - 4000, (ich + 1) * 30000 + 4000)
        f = O.torchgate_T(chunk, 48000, window=w)
        want[s0:e0] = f[0, 4000:4000 + e0 - s0]
    assert O.rel_err(got, want) < TOL


def test_seam_tiles_equal_overlapping_tiles(nr):
    """Abutting apply tiles + seam kernel vs self-contained overlapping tiles: same samples."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = O.synth_signal(150000, seed=91)
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5,
              chunk_size=40000, clip_noise_stationary=True, padding=5000, n_fft=1024, win_length=None,
              hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
              tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=y, **kw)
    a = sg.get_traces()
    try:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOSEAM, 1)
        b = sg.get_traces()
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOSEAM, 0)
    assert O.rel_err(a, b) < 1e-6
    want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=True, chunk_size=40000, padding=5000)
    assert O.rel_err(a, want) < TOL and O.rel_err(b, want) < TOL


def test_full_size_config2_properties(nr):
    """BASELINE.json configs[1] at FULL size (28.8 M samples, 48 chunks): run-to-run determinism,
    sub-range consistency of get_traces, and the oracle on three whole chunks (first, middle, last)."""
    import bench
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = bench.synth_on_device(bench.N_PER_GPU, 1234, torch.device("cuda", 0))
    out1 = nr.reduce_noise(y=y, sr=48000, stationary=True)
    out2 = nr.reduce_noise(y=y, sr=48000, stationary=True)
    assert torch.equal(out1, out2), "not deterministic run to run"
    assert bool(torch.isfinite(out1).all())
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
              clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None,
              time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
              use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=y, **kw)
    sub = sg.get_traces(start_frame=7_000_123, end_frame=9_500_001)     # spans 5 chunks, ragged ends
    assert torch.equal(sub, out1[7_000_123:9_500_001])
    # oracle on whole chunks, threshold from the same clip
    yh = y.cpu().numpy().astype(np.float64)
    thr, _, _ = O.noise_threshold_S(yh[None, :600000], 1024, 1024, 256, 1.5, 600000)
    assert np.max(np.abs(sg.noise_thresh - thr)) < 1e-9
    filt = O.smoothing_filter(5, 9)
    o1 = out1.cpu().numpy()
    for ich in (0, 23, 47):
        chunk = O.read_chunk(yh[None, :], ich * 600000 - 30000, (ich + 1) * 600000 + 30000)
        ref = O.gate_stationary_S(chunk, thr, 1024, 1024, 256, 1.0, filt)[0, 30000:630000]
        assert O.rel_err(o1[ich * 600000:(ich + 1) * 600000], ref) < TOL


@pytest.mark.parametrize("n_fft", [64, 128, 256, 4096, 8192, 40, 250, 1023, 3000, 4095])
@pytest.mark.parametrize("stationary", [True, False])
def test_fft_size_range(nr, n_fft, stationary):
    """Smallest and largest supported transforms: powers of two on the Stockham kernels (one wavefront
    per frame; n_fft = 8192: one workgroup per frame), every other length on the chirp-z kernels."""
    sr = 48000
    y = O.synth_signal(60000, seed=n_fft).astype(np.float64)
    kw = dict(stationary=stationary, n_fft=n_fft, chunk_size=25000, padding=5000)
    if n_fft <= 128:
        kw.update(freq_mask_smooth_hz=3000, time_mask_smooth_ms=10)   # at least one bin / one frame
    if n_fft >= 3000:
        kw.update(time_mask_smooth_ms=100)                            # hop >= 750 samples
    got = nr.reduce_noise(y=y, sr=sr, **kw)
    want = O.reduce_noise_S(y, sr, **kw)
    assert O.rel_err(got, want) < TOL


def test_chunk_size_none_and_zero_padding(nr):
    """chunk_size=None takes the single-window branch (base.py:174,222); padding=0 is legal."""
    y = O.synth_signal(70000, seed=5).astype(np.float64)
    for kw in (dict(stationary=False, chunk_size=None, padding=1000),
               dict(stationary=True, chunk_size=20000, padding=0),
               dict(stationary=False, chunk_size=20000, padding=0)):
        got = nr.reduce_noise(y=y, sr=48000, **kw)
        want = O.reduce_noise_S(y, 48000, **kw)
        assert O.rel_err(got, want) < TOL, kw


@pytest.mark.parametrize("sr", [8000, 22050, 44100])
def test_torchgate_sample_rates(sr):
    """Other smoothing widths (sr=8000 -> 32 bins x 1 frame... wide frequency filter)."""
    from noisereduce_amd.torchgate import TorchGate
    rng = np.random.default_rng(sr)
    x = (0.1 * rng.standard_normal((2, 12000)) + 0.4 * np.sin(2 * np.pi * 300 * np.arange(12000) / sr)[None, :])
    x = x.astype(np.float32).astype(np.float64)
    tg = TorchGate(sr=sr).cuda()
    got = tg(torch.from_numpy(x).cuda()).cpu().numpy()
    want = O.torchgate_T(x, sr, window=torch.hann_window(1024).double().numpy())
    assert got.shape == want.shape
    assert O.rel_err(got, want) < TOL


def test_lean_apply_equals_full_slice_apply(nr):
    """The 3-waves/SIMD apply kernel (half-size slices, wave-private hop accumulators) against the
    2-waves/SIMD one that stores whole frames."""
    from noisereduce_amd import _ffi
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = O.synth_signal(150000, seed=92)
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5,
              chunk_size=40000, clip_noise_stationary=True, padding=5000, n_fft=1024, win_length=None,
              hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
              tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=y, **kw)
    a = sg.get_traces()
    try:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOLEAN, 1)
        b = sg.get_traces()
    finally:
        sg._gate.set_option(_ffi.SG_OPT_FORCE_NOLEAN, 0)
    assert O.rel_err(a, b) < 1e-6
    want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=True, chunk_size=40000, padding=5000)
    assert O.rel_err(a, want) < TOL and O.rel_err(b, want) < TOL


def test_result_buffer_pool(nr):
    """numpy results live in pooled page-locked buffers (_hostbuf.py): a buffer must not be recycled
    while any view of an earlier result is alive, and must be recycled afterwards."""
    import gc
    from noisereduce_amd import _hostbuf
    y = O.synth_signal(600000, seed=3)
    a = nr.reduce_noise(y=y, sr=48000, stationary=True)
    keep = a[1000:2000]                      # a view keeps the whole buffer alive
    snap = keep.copy()
    del a
    gc.collect()
    outs = [nr.reduce_noise(y=O.synth_signal(600000, seed=10 + i), sr=48000, stationary=True) for i in range(6)]
    assert np.array_equal(keep, snap)        # never overwritten by later results
    assert all(o.flags.writeable and o.dtype == y.dtype and o.shape == y.shape for o in outs)
    assert not any(np.shares_memory(outs[i], outs[j]) for i in range(6) for j in range(i))
    live_before = _hostbuf.pool_stats()
    del outs, keep
    gc.collect()
    st = _hostbuf.pool_stats()
    assert sum(st["idle"].values()) >= 1 and st["total_bytes"] == live_before["total_bytes"]
    b = nr.reduce_noise(y=y, sr=48000, stationary=True)   # served from the pool again
    assert _hostbuf.pool_stats()["total_bytes"] == st["total_bytes"]
    want = O.reduce_noise_S(y.astype(np.float64), 48000, stationary=True)
    assert O.rel_err(b, want) < TOL


def test_gate_cache_eviction_keeps_live_gates_usable(nr):
    """The handle cache holds 8 parameter sets; an evicted gate that an object still refers to must
    stay usable (it is freed with its last reference, not at eviction)."""
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = O.synth_signal(30000, seed=2).astype(np.float64)
    kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
              clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None,
              time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
              use_tqdm=False, n_jobs=1)
    first = SpectralGateStationary(y=y, **kw)
    ref = first.get_traces()
    for i in range(10):                               # 10 other parameter sets push `first`'s gate out
        nr.reduce_noise(y=y, sr=48000, stationary=True, prop_decrease=0.5 + 0.01 * i)
    assert np.array_equal(first.get_traces(), ref)


def test_plain_c_caller_matches_python_path(nr, tmp_path):
    """tests/c_abi/example.c: a C99 program (no Python, no torch) that calls the C ABI directly.
    Its output samples must equal those of the Python path on the same signal."""
    import subprocess
    from tests.test_host_cpu import _build_c_example
    exe = _build_c_example(tmp_path)
    n = 700_000                                      # two chunks, the second one partial
    res = subprocess.run([exe, str(n)], check=True, capture_output=True, text=True).stdout.split()
    # the program's signal: 32-bit LCG noise + a 1 kHz sawtooth
    s = np.empty(n, dtype=np.uint32)
    state = np.uint64(12345)
    a, c, m = np.uint64(1664525), np.uint64(1013904223), np.uint64(0xFFFFFFFF)
    for i in range(n):
        state = (state * a + c) & m
        s[i] = state
    y = (np.float32(0.2) * ((s >> np.uint32(8)).astype(np.float32) / np.float32(8388608.0) - np.float32(1.0))
         + np.float32(0.3) * ((np.arange(n) % 48) - 24).astype(np.float32) / np.float32(24.0)).astype(np.float32)
    out = nr.reduce_noise(y=y, sr=48000, stationary=True)
    got = [float(v) for v in res[res.index("first") + 1:res.index("first") + 4]]
    want = [float(out[1000]), float(out[n // 2]), float(out[n - 1000])]
    assert np.allclose(got, want, rtol=0, atol=1e-7 * np.max(np.abs(out))), (got, want)
    e_out = float(res[res.index("out") + 1])
    assert abs(e_out - float(np.sum(out.astype(np.float64) ** 2))) < 1e-5 * e_out


# ---- a NaN sample: against golden vectors of the live reference (tests/golden/make_golden.py gen_nan) ----
from tests.golden.cases import (S_NAN_CASES, T_NAN_CASES, make_input_S_nan, make_input_T_nan,  # noqa: E402
                                nonfinite_agree)


@pytest.mark.parametrize("name", sorted(S_NAN_CASES))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_nan_sample_golden(nr, golden_dir, name, dtype):
    case = S_NAN_CASES[name]
    g = _load(golden_dir, "S_nan_" + name)
    y, y_noise = make_input_S_nan(case)
    out = nr.reduce_noise(y=y.astype(dtype), sr=case["sr"],
                          y_noise=None if y_noise is None else y_noise.astype(dtype), **case["kwargs"])
    assert nonfinite_agree(out, g["out"], TOL) is None, nonfinite_agree(out, g["out"], TOL)


@pytest.mark.parametrize("name", sorted(T_NAN_CASES))
def test_nan_sample_torchgate_golden(golden_dir, name):
    from noisereduce_amd.torchgate import TorchGate
    case = T_NAN_CASES[name]
    g = _load(golden_dir, "T_nan_" + name)
    x, xn = make_input_T_nan(case)
    tg = TorchGate(sr=case["sr"], **case["kwargs"]).cuda()
    out = tg(torch.from_numpy(x).float().cuda(), None if xn is None else torch.from_numpy(xn).float().cuda())
    assert nonfinite_agree(out.cpu().numpy(), g["out"], TOL) is None, nonfinite_agree(out.cpu().numpy(), g["out"], TOL)


@pytest.mark.parametrize("n_fft", [256, 1024])
def test_noise_threshold_with_silent_gaps(nr, n_fft):
    """Noise statistics of a clip with digital-silence gaps: EVERY band has zero-power cells, more than two per time slice
    of the single-pass statistics -- k_colstats1_final's rescan of such slices (round 5; before: the band recomputed whole)
    against the reference's mean + 1.5 std of the floored dB (stationary.py:75-81), and the gated output against the oracle."""
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    sr, n = 48000, 700000
    y = O.synth_signal(n, sr=sr, seed=77).astype(np.float32)
    for k in range(9):
        y[k * 65000 + 1000:k * 65000 + 1000 + 3 * n_fft + 17 * k] = 0.0
    y[300000:300050] = 1e-7 * y[300000:300050]                       # a few nearly-silent samples inside a frame
    kw = dict(sr=sr, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
              clip_noise_stationary=True, padding=30000, n_fft=n_fft, win_length=None, hop_length=None,
              time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
              use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=torch.from_numpy(y).cuda(), **kw)
    thr, _, _ = O.noise_threshold_S(y.astype(np.float64)[None, :], n_fft, n_fft, n_fft // 4, 1.5, 600000)
    assert np.max(np.abs(sg.noise_thresh - thr)) < 1e-9
    got = sg.get_traces().cpu().numpy()
    want = O.reduce_noise_S(y.astype(np.float64), sr, stationary=True, n_fft=n_fft)
    assert O.rel_err(got, want) < TOL
