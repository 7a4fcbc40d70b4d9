"""Host arrays in / out: the recording crosses PCIe in chunk-aligned pieces, overlapped with the gate
(noisereduce_amd/spectralgate/base.py: _get_traces_pipelined; SURVEY.md 8 row f2 -- the reference's chunk loop over a
memmap, base.py:180-216).  Each piece is gated as a start_frame / end_frame range of the one device copy, so the chunk grid
-- and, on the float32 kernels and for integer outputs, every output bit -- is that of the one-upload path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rec(n, c, dtype, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 48000.0
    y = 0.1 * rng.standard_normal((c, n)) + 0.4 * np.sin(2 * np.pi * 700.0 * t)[None, :]
    if np.dtype(dtype).kind == "i":
        return (y * 12000).astype(dtype)
    return y.astype(dtype)


@pytest.mark.parametrize("stationary", [True, False])
@pytest.mark.parametrize("dtype,c,n", [
    (np.float32, 1, 600000 * 7 + 12345),     # ragged last piece
    (np.float32, 2, 600000 * 10),            # two channels (strided host slices), exact multiple
    (np.int16, 1, 600000 * 9 + 1),           # integer recording (float64 pipeline), one-sample tail
    (np.float64, 1, 600000 * 9 + 777),
])
def test_pipelined_host_path_is_the_one_upload_path_bit_for_bit(monkeypatch, stationary, dtype, c, n):
    import noisereduce_amd as nr
    from noisereduce_amd.spectralgate import base
    y = _rec(n, c, dtype, 5)
    y = y[0] if c == 1 else y
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE", "0")
    ref = nr.reduce_noise(y=y, sr=48000, stationary=stationary)
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE", "1")
    # small pieces: one, two and three chunks per piece
    for piece_chunks in (1, 2, 3):
        monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE_PIECE_BYTES", str(piece_chunks * 600000 * c * np.dtype(dtype).itemsize))
        calls = []
        orig = base.SpectralGate._get_traces_pipelined
        def counted(self):
            out = orig(self)
            calls.append((len(self._pipe["pieces"]), out is not None))
            return out
        monkeypatch.setattr(base.SpectralGate, "_get_traces_pipelined", counted)
        got = nr.reduce_noise(y=y, sr=48000, stationary=stationary)
        monkeypatch.setattr(base.SpectralGate, "_get_traces_pipelined", orig)
        assert calls and calls[0][0] >= 3 and calls[0][1], "the pipelined path did not run"
        assert got.dtype == ref.dtype and got.shape == ref.shape
        assert np.array_equal(got, ref), (piece_chunks, int(np.argmax(got != ref)))


def test_pipelined_path_default_piece_size_and_fallbacks(monkeypatch):
    import noisereduce_amd as nr
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    y = _rec(600000 * 40, 1, np.float32, 9)[0]
    kw = dict(y=y, sr=48000, y_noise=None, prop_decrease=1.0, time_constant_s=2.0, freq_mask_smooth_hz=500,
              time_mask_smooth_ms=50, n_std_thresh_stationary=1.5, tmp_folder=None, chunk_size=600000, padding=30000,
              n_fft=1024, win_length=None, hop_length=None, clip_noise_stationary=True, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(**kw)
    pieces = sg._pipeline_pieces()
    assert sg._pipe is not None and sg._pipe["sent"] == 1200000 + 30000, "the constructor sends up the first piece only"
    assert [(b - a) // 600000 for a, b in pieces] == [2, 10, 10, 10, 6, 2] and pieces[0][0] == 0 and pieces[-1][1] == y.size
    assert sg._y_dev is None, "the constructor uploaded the whole recording"
    a = sg.get_traces()
    assert sg._pipe is None and sg._y_dev is not None and sg._y_dev.shape == (1, y.size)
    assert np.array_equal(sg._y_dev.cpu().numpy()[0], y)
    # a sub-range request, a short recording and clip_noise_stationary=False all take the one-upload path
    sub = sg.get_traces(start_frame=600000, end_frame=1800000)
    assert np.array_equal(sub, a[600000:1800000])
    assert SpectralGateStationary(**dict(kw, y=y[:1500000]))._pipeline_pieces() is None
    sg2 = SpectralGateStationary(**dict(kw, clip_noise_stationary=False))
    assert sg2._y_dev is not None and sg2._pipe is None
    # a sub-range asked of an object whose upload is in flight: the rest goes up, then the one-upload path
    sg3 = SpectralGateStationary(**kw)
    assert np.array_equal(sg3.get_traces(start_frame=600000, end_frame=1800000), a[600000:1800000]) and sg3._pipe is None
    # the operator seam on such an object
    sg4 = SpectralGateStationary(**kw)
    ch = sg4.filter_chunk(600000, 1200000)
    assert np.allclose(ch[0], a[600000:1200000], atol=2e-6)
    assert np.array_equal(sg4._device_y().cpu().numpy()[0], y) and sg4._pipe is None    # (the device copy on demand: the rest goes up)
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE", "0")
    assert np.array_equal(SpectralGateStationary(**kw).get_traces(), a)


@pytest.mark.parametrize("kw", [dict(n_fft=512), dict(n_fft=256), dict(n_fft=400, freq_mask_smooth_hz=1000), dict(n_fft=2048),
                                dict(prop_decrease=0.6), dict(chunk_size=100000, padding=9000), dict(freq_mask_smooth_hz=None, time_mask_smooth_ms=None),
                                dict(precision="float64")])
@pytest.mark.parametrize("stationary", [True, False])
def test_pipelined_host_path_other_geometries(monkeypatch, stationary, kw):
    """Every kernel family behind get_traces stores its output once and never reads it back: the page-locked result array
    works as its destination for the generic frame lengths, the chirp-z sizes, prop_decrease < 1, unsmoothed masks and the
    float64 pipeline too."""
    import noisereduce_amd as nr
    from noisereduce_amd.spectralgate import base
    cs = kw.get("chunk_size", 600000)
    y = _rec(cs * 8 + 4321, 1, np.float32, 12)[0]
    yn = y[5000:90000].copy()
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE", "0")
    ref = nr.reduce_noise(y=y, sr=48000, stationary=stationary, **kw)
    refn = nr.reduce_noise(y=y, sr=48000, stationary=True, y_noise=yn, **kw) if stationary else None
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE", "1")
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE_PIECE_BYTES", str(2 * cs * 4))
    calls = []
    orig = base.SpectralGate._get_traces_pipelined
    monkeypatch.setattr(base.SpectralGate, "_get_traces_pipelined", lambda self: (calls.append(1), orig(self))[1])
    def same(a, b):
        if "precision" not in kw:
            return np.array_equal(a, b)
        # the float64 kernels add the four overlapping frames of a hop in an order that depends on where the request's
        # tiles start: float64 rounding (1e-16), which the float32 container shows as a last-place flip in a few samples
        # (the same holds for get_traces(start, end) on a device-resident recording)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        return d.max() <= 1.2e-7 * np.abs(b).max() and np.count_nonzero(d) <= 1e-4 * d.size
    got = nr.reduce_noise(y=y, sr=48000, stationary=stationary, **kw)
    assert calls and same(got, ref)
    if stationary:   # an explicit noise clip: the constructor uploads the clip, the recording still streams
        gotn = nr.reduce_noise(y=y, sr=48000, stationary=True, y_noise=yn, **kw)
        assert len(calls) == 2 and same(gotn, refn)


def test_pipelined_host_path_from_two_threads(monkeypatch):
    """Two threads gate different host recordings through the same cached engine handle at once: the uploads share the side
    stream, the gates serialise on the handle's lock, each thread gets its own result."""
    import threading
    import noisereduce_amd as nr
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE_PIECE_BYTES", str(2 * 600000 * 4))
    ys = [_rec(600000 * 9 + 100 * i, 1, np.float32, 30 + i)[0] for i in range(2)]
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE", "0")
    refs = [nr.reduce_noise(y=y, sr=48000, stationary=True) for y in ys]
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE", "1")
    out, errs = [None, None], []

    def work(i):
        try:
            for _ in range(4):
                out[i] = nr.reduce_noise(y=ys[i], sr=48000, stationary=True)
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    assert not errs, errs
    for i in range(2):
        assert np.array_equal(out[i], refs[i])


@pytest.mark.parametrize("stationary", [True, False])
def test_host_views_the_piecewise_upload_cannot_take(monkeypatch, stationary):
    """a time-reversed view (negative stride) made the pipelined upload raise in torch.from_numpy once the
    recording was long enough to pipeline; such views -- and Fortran-ordered / transposed multichannel inputs, whose rows
    are strided but positive -- must give what their contiguous copies give."""
    import noisereduce_amd as nr
    from noisereduce_amd.spectralgate import base
    n = 600000 * 7 + 4321
    y = _rec(n, 1, np.float32, 21)[0]
    monkeypatch.setenv("NOISEREDUCE_AMD_PIPELINE_PIECE_BYTES", str(600000 * 4))
    rev = y[::-1]
    assert rev.strides[0] < 0
    got = nr.reduce_noise(y=rev, sr=48000, stationary=stationary)
    want = nr.reduce_noise(y=np.ascontiguousarray(rev), sr=48000, stationary=stationary)
    assert np.array_equal(got, want)
    # two channels: Fortran order, a transposed (n, 2) array, and a channel-reversed view
    y2 = _rec(n, 2, np.float32, 22)
    want2 = nr.reduce_noise(y=y2, sr=48000, stationary=stationary)
    for view in (np.asfortranarray(y2), np.ascontiguousarray(y2.T).T):
        assert not view.flags["C_CONTIGUOUS"]
        assert np.array_equal(nr.reduce_noise(y=view, sr=48000, stationary=stationary), want2)
    got3 = nr.reduce_noise(y=y2[::-1], sr=48000, stationary=stationary)
    assert np.array_equal(got3, nr.reduce_noise(y=np.ascontiguousarray(y2[::-1]), sr=48000, stationary=stationary))
    # the guard itself
    sg = base.SpectralGate.__new__(base.SpectralGate)
    sg._tensor_io, sg._chunk_size, sg._dtype, sg.y, sg.n_channels, sg.n_frames = False, 600000, np.float32, y2[:, ::-1], 2, n
    assert sg._pipeline_pieces() is None
    sg.y = y2
    assert sg._pipeline_pieces() is not None
