"""CPU oracle for the spectral-gating hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy (float64) restatement of the algorithm behind
timsainb/noisereduce's ``reduce_noise()`` / ``SpectralGate._do_filter()`` /
``TorchGate.forward()``.  It exists to CHECK the HIP path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``noisereduce_amd`` never does and fails loudly when
its HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` (committed) imports the
live reference (``PYTHONPATH=/root/reference``, numpy 2.2.6 / scipy 1.15.3 /
torch 2.10.0) and stores its outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function here against them.  The
reference's own tests hold no assertions and no golden vectors
(/root/reference/test_reduction.py:17,31,45,59,73,101), so those generated
fixtures are the pin.  In addition the building blocks (STFT/ISTFT both flavours,
the one-pole forward-backward smoother, the boxcar) are compared at test time with
the scipy / torch primitives the reference calls, which are installed wherever
the tests run.

The arithmetic the reference delegates to third-party code is restated from
the libraries installed next to it (unpinned in /root/reference/setup.py:24-27;
de-facto scipy 1.15.3, torch 2.10.0):

* scipy.signal.stft / istft   -> ``stft_scipy`` / ``istft_scipy``
  (scipy/signal/_spectral_py.py:2052,2089-2094,2128,2185-2202 and :1689-1725)
* scipy.signal.fftconvolve    -> ``conv2_same``  (scipy/signal/_signaltools.py:582)
* scipy.signal.filtfilt(padtype=None) for a one-pole IIR -> ``filtfilt_onepole``
  (scipy/signal/_signaltools.py:4532-4567)
* torch.stft / torch.istft (center=True, pad_mode="constant") -> ``stft_torch`` /
  ``istft_torch``; torch conv1d/conv2d(padding="same") -> ``boxcar_same`` /
  ``conv2_same``.

Everything computes in float64, like the reference's numpy path does
(/root/reference/noisereduce/spectralgate/base.py:140 allocates float64 chunks).
"""
import math

import numpy as np

EPS64 = float(np.finfo(np.float64).eps)


# --------------------------------------------------------------------------
# windows / geometry
# --------------------------------------------------------------------------
def hann_periodic(W):
    """scipy.signal.get_window('hann', W) (fftbins=True => periodic) and
    torch.hann_window(W) (periodic=True): w[k] = 0.5 - 0.5 cos(2 pi k / W)."""
    k = np.arange(W, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / W)


def resolve_stft_params(n_fft, win_length, hop_length):
    """base.py:77-86 / torchgate.py:55-58."""
    W = n_fft if win_length is None else win_length
    H = W // 4 if hop_length is None else hop_length
    return n_fft, W, H


def n_frames_for(L, W, H):
    """Number of STFT columns for a length-L signal, zero-extended by W//2 on
    both sides, padded=False (scipy/_spectral_py.py:2052,2185-2189)."""
    return (L + 2 * (W // 2) - W) // H + 1


# --------------------------------------------------------------------------
# scipy-flavoured STFT / ISTFT  (variant "S")
# --------------------------------------------------------------------------
def stft_scipy(x, n_fft, W, H):
    """scipy.signal.stft(x, nfft=n_fft, nperseg=W, noverlap=W-H, padded=False)
    as called at stationary.py:67-73,87-93 and nonstationary.py:51-57.
    x: (L,) float64.  Returns Z (F, T) complex128, scaled by 1/sum(w)."""
    x = np.asarray(x, dtype=np.float64)
    if x.shape[-1] < W:
        raise ValueError("signal shorter than win_length")
    w = hann_periodic(W)
    ext = np.concatenate([np.zeros(W // 2), x, np.zeros(W // 2)])
    T = (ext.shape[0] - W) // H + 1
    idx = np.arange(W)[None, :] + H * np.arange(T)[:, None]
    frames = ext[idx] * w[None, :]
    Z = np.fft.rfft(frames, n=n_fft, axis=-1)  # zero-pads at the END (:2202)
    Z *= 1.0 / w.sum()
    return np.ascontiguousarray(Z.T)


def istft_scipy(Z, n_fft, W, H):
    """scipy.signal.istft(Z, nfft=n_fft, nperseg=W, noverlap=W-H)
    (scipy/_spectral_py.py:1689-1725).  Z: (F, T).  Returns (T-1)*H + W - 2*(W//2)
    samples."""
    w = hann_periodic(W)
    T = Z.shape[1]
    xs = np.fft.irfft(Z, n=n_fft, axis=0)[:W, :] * w.sum()
    out_len = W + (T - 1) * H
    x = np.zeros(out_len)
    norm = np.zeros(out_len)
    w2 = w * w
    for t in range(T):
        x[t * H:t * H + W] += xs[:, t] * w
        norm[t * H:t * H + W] += w2
    h = W // 2
    x = x[h:out_len - h]
    norm = norm[h:out_len - h]
    return x / np.where(norm > 1e-10, norm, 1.0)


# --------------------------------------------------------------------------
# torch-flavoured STFT / ISTFT  (variant "T")
# --------------------------------------------------------------------------
def _centered_window(n_fft, W, window=None):
    """torch.stft pads a win_length < n_fft window on both sides to n_fft.
    ``window``: optional (W,) table; TorchGate builds torch.hann_window(W) in
    float32 (torchgate.py:150,231,261) even for float64 input, so parity with the
    reference run in float64 needs that float32-rounded table passed in."""
    w = hann_periodic(W) if window is None else np.asarray(window, dtype=np.float64)
    if W == n_fft:
        return w
    left = (n_fft - W) // 2
    full = np.zeros(n_fft)
    full[left:left + W] = w
    return full


def stft_torch(x, n_fft, W, H, window=None):
    """torch.stft(x, n_fft, H, W, window=hann, center=True, pad_mode="constant",
    return_complex=True) as called at torchgate.py:142-151,223-232.
    x: (B, L).  Returns (B, F, T) complex128, unscaled, T = 1 + (L + 2 (n_fft // 2) - n_fft) // H
    (= 1 + L // H for even n_fft; an odd n_fft pads one sample less than a frame)."""
    x = np.asarray(x, dtype=np.float64)
    B, L = x.shape
    wf = _centered_window(n_fft, W, window)
    p = n_fft // 2
    ext = np.concatenate([np.zeros((B, p)), x, np.zeros((B, p))], axis=1)
    T = 1 + (L + 2 * p - n_fft) // H
    idx = np.arange(n_fft)[None, :] + H * np.arange(T)[:, None]
    frames = ext[:, idx] * wf[None, None, :]
    Z = np.fft.rfft(frames, axis=-1)
    return np.ascontiguousarray(np.swapaxes(Z, 1, 2))


def istft_torch(Z, n_fft, W, H, window=None):
    """torch.istft(Y, n_fft, H, W, window=hann, center=True) as called at
    torchgate.py:255-262.  Z: (B, F, T).  Returns (B, H*(T-1)) -- one sample more when n_fft is
    odd: torch trims n_fft//2 from both ends of the n_fft + H*(T-1) overlap-add buffer."""
    B, F, T = Z.shape
    wf = _centered_window(n_fft, W, window)
    xs = np.fft.irfft(Z, n=n_fft, axis=1) * wf[None, :, None]
    out_len = n_fft + (T - 1) * H
    x = np.zeros((B, out_len))
    env = np.zeros(out_len)
    w2 = wf * wf
    for t in range(T):
        x[:, t * H:t * H + n_fft] += xs[:, :, t]
        env[t * H:t * H + n_fft] += w2
    p = n_fft // 2
    end = out_len - p
    x = x[:, p:end]
    env = env[p:end]
    if not np.all(np.abs(env) > 1e-11):
        raise RuntimeError("window overlap add min: 1")  # torch's NOLA error
    return x / env[None, :]


# --------------------------------------------------------------------------
# dB / sigmoid helpers
# --------------------------------------------------------------------------
def amp_to_db(x, top_db):
    """spectralgate/utils.py:11-16 (top_db=80) and torchgate/utils.py:5-23
    (top_db=40): 20 log10(|x| + eps), floored at (row max over the LAST axis) -
    top_db."""
    x_db = 20.0 * np.log10(np.abs(x) + EPS64)
    return np.maximum(x_db, np.max(x_db, axis=-1, keepdims=True) - top_db)


def sigmoid_shifted(x, shift, mult):
    """spectralgate/utils.py:4-8."""
    return 1.0 / (1.0 + np.exp(-(x + shift) * mult))


# --------------------------------------------------------------------------
# mask smoothing filter design and application
# --------------------------------------------------------------------------
def triangle(m):
    """One axis of base.py:7-29: concat(linspace(0,1,m+1,endpoint=False),
    linspace(1,0,m+2))[1:-1]  ==  [1..m, m+1, m..1] / (m+1), length 2m+1."""
    return np.concatenate([
        np.linspace(0, 1, m + 1, endpoint=False),
        np.linspace(1, 0, m + 2),
    ])[1:-1]


def smoothing_filter(n_grad_freq, n_grad_time):
    """base.py:7-29 / torchgate.py:106-124: outer(tri(nf), tri(nt)) / sum."""
    f = np.outer(triangle(n_grad_freq), triangle(n_grad_time))
    return f / f.sum()


def mask_smoothing_widths(sr, n_fft, H, freq_mask_smooth_hz, time_mask_smooth_ms):
    """base.py:99-128 (and torchgate.py:73-104).  Returns (nf, nt, smooth_mask).
    Raises ValueError like the reference when a width is below one bin."""
    if freq_mask_smooth_hz is None and time_mask_smooth_ms is None:
        return 1, 1, False
    if freq_mask_smooth_hz is None:
        nf = 1
    else:
        nf = int(freq_mask_smooth_hz / (sr / (n_fft / 2)))
        if nf < 1:
            raise ValueError("freq_mask_smooth_hz needs to be at least {}Hz".format(
                int((sr / (n_fft / 2)))))
    if time_mask_smooth_ms is None:
        nt = 1
    else:
        nt = int(time_mask_smooth_ms / ((H / sr) * 1000))
        if nt < 1:
            raise ValueError("time_mask_smooth_ms needs to be at least {}ms".format(
                int((H / sr) * 1000)))
    return nf, nt, not (nf == 1 and nt == 1)


def conv2_same(m, K):
    """Zero-padded, centred 2-D convolution with an odd-sized kernel:
    scipy.signal.fftconvolve(m, K, mode="same") (stationary.py:114,
    nonstationary.py:80) and torch conv2d(padding="same") (torchgate.py:245-249;
    K is symmetric so correlation == convolution).  FFT-based like scipy's."""
    a, b = K.shape
    s0, s1 = m.shape[0] + a - 1, m.shape[1] + b - 1
    full = np.fft.irfft2(np.fft.rfft2(m, (s0, s1)) * np.fft.rfft2(K, (s0, s1)), (s0, s1))
    o0, o1 = (a - 1) // 2, (b - 1) // 2
    return full[o0:o0 + m.shape[0], o1:o1 + m.shape[1]]


def conv2_same_direct(m, K):
    """Same as conv2_same by direct summation (used to cross-check)."""
    a, b = K.shape
    ha, hb = (a - 1) // 2, (b - 1) // 2
    P = np.zeros((m.shape[0] + a - 1, m.shape[1] + b - 1))
    P[ha:ha + m.shape[0], hb:hb + m.shape[1]] = m
    out = np.zeros_like(m, dtype=np.float64)
    for i in range(a):
        for j in range(b):
            out += K[a - 1 - i, b - 1 - j] * P[i:i + m.shape[0], j:j + m.shape[1]]
    return out


# --------------------------------------------------------------------------
# non-stationary noise floors
# --------------------------------------------------------------------------
def iir_coefficient(time_constant_s, sr, H):
    """nonstationary.py:109-114."""
    t_frames = time_constant_s * sr / float(H)
    return (np.sqrt(1 + 4 * t_frames ** 2) - 1) / (2 * t_frames ** 2)


def filtfilt_onepole(b, A):
    """scipy.signal.filtfilt([b], [1, b-1], A, axis=-1, padtype=None)
    (nonstationary.py:115).  lfilter_zi gives zi = 1-b, so the forward pass is
    s[0] = A[0], s[t] = b A[t] + (1-b) s[t-1]; the backward pass is the same
    recurrence run from the end, seeded with the forward pass's last value."""
    A = np.asarray(A, dtype=np.float64)
    T = A.shape[-1]
    fwd = np.empty_like(A)
    prev = A[..., 0].copy()
    for t in range(T):
        prev = b * A[..., t] + (1.0 - b) * prev
        fwd[..., t] = prev
    out = np.empty_like(A)
    prev = fwd[..., T - 1].copy()
    for t in range(T - 1, -1, -1):
        prev = b * fwd[..., t] + (1.0 - b) * prev
        out[..., t] = prev
    return out


def boxcar_same(A, k):
    """conv1d(A, ones(k), padding="same") / k along the last axis
    (torchgate.py:179-190).  torch pads (k-1)//2 zeros on the left and the rest on
    the right."""
    T = A.shape[-1]
    left = (k - 1) // 2
    P = np.zeros(A.shape[:-1] + (T + k - 1,))
    P[..., left:left + T] = A
    c = np.cumsum(np.concatenate([np.zeros(A.shape[:-1] + (1,)), P], axis=-1), axis=-1)
    return (c[..., k:k + T] - c[..., 0:T]) / k


# --------------------------------------------------------------------------
# variant S: the numpy/scipy spectral gate
# --------------------------------------------------------------------------
def noise_threshold_S(y_noise_2d, n_fft, W, H, n_std, chunk_size, clip_noise=True):
    """stationary.py:47-81.  y_noise_2d: (C, N) float64 (the reference's
    self.y_noise before the channel mean).  Returns (thresh[F], mean[F], std[F])."""
    yn = np.mean(np.asarray(y_noise_2d, dtype=np.float64), axis=0)
    if clip_noise:
        yn = yn[:chunk_size]
    Zn = stft_scipy(yn, n_fft, W, H)
    db = amp_to_db(Zn, 80.0)
    mean = np.mean(db, axis=1)
    std = np.std(db, axis=1)
    return mean + std * n_std, mean, std


def gate_stationary_S(chunk, thresh, n_fft, W, H, prop_decrease, filt,
                      return_stages=False):
    """SpectralGateStationary.spectral_gating_stationary (stationary.py:83-127).
    chunk: (C, Lp) float64; filt: 2-D smoothing filter or None."""
    chunk = np.asarray(chunk, dtype=np.float64)
    out = np.zeros_like(chunk)
    stages = []
    for ci in range(chunk.shape[0]):
        Z = stft_scipy(chunk[ci], n_fft, W, H)
        db = amp_to_db(Z, 80.0)
        raw = db > thresh[:, None]
        m = raw * prop_decrease + np.ones(raw.shape) * (1.0 - prop_decrease)
        if filt is not None:
            m = conv2_same(m, filt)
        y = istft_scipy(Z * m, n_fft, W, H)
        out[ci, :len(y)] = y
        if return_stages:
            stages.append(dict(Z=Z, raw=raw, mask=m))
    return (out, stages) if return_stages else out


def gate_nonstationary_S(chunk, n_fft, W, H, prop_decrease, filt, iir_b,
                         thresh_n_mult, sigmoid_slope, return_stages=False):
    """SpectralGateNonStationary.spectral_gating_nonstationary
    (nonstationary.py:47-97)."""
    chunk = np.asarray(chunk, dtype=np.float64)
    out = np.zeros_like(chunk)
    stages = []
    for ci in range(chunk.shape[0]):
        Z = stft_scipy(chunk[ci], n_fft, W, H)
        A = np.abs(Z)
        S = filtfilt_onepole(iir_b, A)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = (A - S) / S
        m = sigmoid_shifted(r, -thresh_n_mult, sigmoid_slope)
        raw = m
        if filt is not None:
            m = conv2_same(m, filt)
        m = m * prop_decrease + np.ones(m.shape) * (1.0 - prop_decrease)
        y = istft_scipy(Z * m, n_fft, W, H)
        out[ci, :len(y)] = y
        if return_stages:
            stages.append(dict(Z=Z, raw=raw, mask=m, S=S))
    return (out, stages) if return_stages else out


def read_chunk(y2d, i1, i2):
    """SpectralGate._read_chunk (base.py:130-142): float64 window [i1, i2) of a
    (C, N) recording, zeros outside [0, N)."""
    C, N = y2d.shape
    i1b, i2b = max(i1, 0), min(i2, N)
    chunk = np.zeros((C, i2 - i1))
    if i2b > i1b:
        chunk[:, i1b - i1:i2b - i1] = y2d[:, i1b:i2b]
    return chunk


def reduce_noise_S(y, sr, stationary=False, y_noise=None, prop_decrease=1.0,
                   time_constant_s=2.0, freq_mask_smooth_hz=500,
                   time_mask_smooth_ms=50, thresh_n_mult_nonstationary=2,
                   sigmoid_slope_nonstationary=10, n_std_thresh_stationary=1.5,
                   chunk_size=600000, padding=30000, n_fft=1024, win_length=None,
                   hop_length=None, clip_noise_stationary=True):
    """reduce_noise(use_torch=False) end to end: noisereduce.py:13-185 ->
    SpectralGate.__init__ (base.py:33-97) -> get_traces (base.py:167-226) ->
    filter_chunk (base.py:144-150) -> _do_filter."""
    y = np.array(y)
    flat = y.ndim == 1
    if y.ndim > 2:
        raise ValueError("Waveform must be in shape (# frames, # channels)")
    y2 = y[None, :] if flat else y
    dtype = y.dtype
    C, N = y2.shape
    n_fft, W, H = resolve_stft_params(n_fft, win_length, hop_length)
    nf, nt, smooth = mask_smoothing_widths(sr, n_fft, H, freq_mask_smooth_hz,
                                           time_mask_smooth_ms)
    filt = smoothing_filter(nf, nt) if smooth else None

    if stationary:
        if y_noise is None:
            yn2 = y2
        else:
            yn = np.array(y_noise)
            yn2 = yn[None, :] if yn.ndim == 1 else yn
        thresh, _, _ = noise_threshold_S(yn2, n_fft, W, H, n_std_thresh_stationary,
                                         chunk_size, clip_noise_stationary)

        def do_filter(chunk):
            return gate_stationary_S(chunk, thresh, n_fft, W, H, prop_decrease, filt)
    else:
        b = iir_coefficient(time_constant_s, sr, H)

        def do_filter(chunk):
            return gate_nonstationary_S(chunk, n_fft, W, H, prop_decrease, filt, b,
                                        thresh_n_mult_nonstationary,
                                        sigmoid_slope_nonstationary)

    def filter_chunk(start, end):
        i1, i2 = start - padding, end + padding
        res = do_filter(read_chunk(y2, i1, i2))
        return res[:, start - i1:end - i1]

    if chunk_size is not None and N > chunk_size:
        out = np.zeros((C, N), dtype=dtype)
        ich2 = int((N - 1) / chunk_size)
        for ich in range(0, ich2 + 1):
            s0 = ich * chunk_size
            e0 = min((ich + 1) * chunk_size, N)
            # base.py:152-165: every chunk is filtered over its full chunk_size
            # window (zeros beyond N), then cut to [0, end0).
            full = filter_chunk(s0, (ich + 1) * chunk_size)
            out[:, s0:e0] = full[:, :e0 - s0].astype(dtype)
    else:
        out = filter_chunk(0, N).astype(dtype)
    return out.flatten() if flat else out


# --------------------------------------------------------------------------
# variant T: TorchGate
# --------------------------------------------------------------------------
def torchgate_T(x, sr, xn=None, nonstationary=False, n_std_thresh_stationary=1.5,
                n_thresh_nonstationary=1.3, temp_coeff_nonstationary=0.1,
                n_movemean_nonstationary=20, prop_decrease=1.0, n_fft=1024,
                win_length=None, hop_length=None, freq_mask_smooth_hz=500,
                time_mask_smooth_ms=50, window=None, return_stages=False):
    """TorchGate.forward (torchgate.py:200-264) evaluated in float64.
    x: (B, L); xn: None or (Bn, Ln) with Bn in {1, B}; window: optional (W,)
    table (see _centered_window)."""
    x = np.asarray(x, dtype=np.float64)
    assert x.ndim == 2
    n_fft, W, H = resolve_stft_params(n_fft, win_length, hop_length)
    if x.shape[-1] < W * 2:
        raise Exception(f"x must be bigger than {W * 2}")
    if xn is not None and np.shape(xn)[-1] < W * 2:
        raise Exception(f"xn must be bigger than {W * 2}")
    nf, nt, smooth = mask_smoothing_widths(sr, n_fft, H, freq_mask_smooth_hz,
                                           time_mask_smooth_ms)
    filt = smoothing_filter(nf, nt) if smooth else None

    X = stft_torch(x, n_fft, W, H, window)
    if nonstationary:
        A = np.abs(X)
        S = boxcar_same(A, n_movemean_nonstationary)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = (A - S) / S
        z = (r - n_thresh_nonstationary) / temp_coeff_nonstationary
        with np.errstate(over="ignore"):
            raw = 1.0 / (1.0 + np.exp(-z))
        thresh = None
    else:
        X_db = amp_to_db(X, 40.0)
        if xn is not None:
            XN_db = amp_to_db(stft_torch(np.atleast_2d(np.asarray(xn, dtype=np.float64)),
                                          n_fft, W, H, window), 40.0)
        else:
            XN_db = X_db
        mean = np.mean(XN_db, axis=-1)
        std = np.std(XN_db, axis=-1, ddof=1)
        thresh = mean + std * n_std_thresh_stationary
        raw = (X_db > thresh[:, :, None]).astype(np.float64)
    m = prop_decrease * (raw * 1.0 - 1.0) + 1.0
    if filt is not None:
        m = np.stack([conv2_same(m[b], filt) for b in range(m.shape[0])])
    y = istft_torch(X * m, n_fft, W, H, window)
    if return_stages:
        return y, dict(X=X, raw=raw, mask=m, thresh=thresh)
    return y


# --------------------------------------------------------------------------
# synthetic workloads (BASELINE.json configs)
# --------------------------------------------------------------------------
def synth_signal(n, sr=48000, seed=1234, tone_hz=1000.0, tone_amp=0.5, noise_sigma=0.1,
                 dtype=np.float32):
    """White noise + tone, the BASELINE.json synthetic workload."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / sr
    y = noise_sigma * rng.standard_normal(n) + tone_amp * np.sin(2 * np.pi * tone_hz * t)
    return y.astype(dtype)


def rel_err(a, b):
    """max |a-b| / max |b|  -- the parity metric (north_star: <= 1e-4)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.max(np.abs(a - b)) if a.size else 0.0
    s = np.max(np.abs(b)) if b.size else 0.0
    return d / s if s > 0 else d
