"""torch-CPU restatement of ``TorchGate.forward``  --  TEST INFRASTRUCTURE ONLY (like spectralgate_oracle.py).

The numpy oracle (``spectralgate_oracle.torchgate_T``) restates the algorithm in float64 numpy.  This file states the
same algorithm with the torch primitives the reference itself calls on a CPU tensor -- ``torch.stft`` /
``torch.istft`` (center=True, pad_mode="constant"), ``torch.std_mean`` (ddof = 1), ``conv1d`` / ``conv2d``
(padding="same") -- so that ``bench.py``'s ``cpu_baseline`` has a like-for-like "torch on the host cores" leg for
BASELINE.json configs[4] (the reference cannot travel to the GPU box).  Follows
/root/reference/noisereduce/torchgate/torchgate.py:73-124 (filter), :126-165 (stationary mask), :167-198
(non-stationary mask), :200-264 (forward) and torchgate/utils.py:5-23 (amp_to_db), :26-39 (temperature sigmoid),
:42-66 (linspace).  Only ``tests/`` and ``bench.py``'s cpu_baseline leg import it.

Parity status: PINNED -- tests/test_oracle_golden.py checks it against the golden vectors of the live reference
(tests/golden/T_*.npz) next to the numpy oracle.
"""
import torch
import torch.nn.functional as F


def _db(X, top_db=40.0, eps=torch.finfo(torch.float64).eps):
    # torchgate/utils.py:5-23: 20 log10(|X| + eps), floored at (max over the LAST axis) - top_db
    d = 20.0 * torch.log10(X.abs() + eps)
    return torch.maximum(d, d.max(dim=-1, keepdim=True).values - top_db)


def _ramp(m, dtype):
    # torchgate/utils.py:42-66 with endpoint=False, start 0, stop 1, m + 1 points, then [1:]: 1/(m+1) .. m/(m+1)
    return torch.arange(1, m + 1, dtype=dtype) / (m + 1)


def smoothing_filter(sr, n_fft, hop, freq_mask_smooth_hz, time_mask_smooth_ms, dtype=torch.float32):
    """torchgate.py:73-124: outer product of two triangles, unit sum; None when both widths collapse to 1."""
    if freq_mask_smooth_hz is None and time_mask_smooth_ms is None:
        return None
    nf = 1 if freq_mask_smooth_hz is None else int(freq_mask_smooth_hz / (sr / (n_fft / 2)))
    nt = 1 if time_mask_smooth_ms is None else int(time_mask_smooth_ms / ((hop / sr) * 1000))
    if nf < 1 or nt < 1:
        raise ValueError("mask smoothing width below one bin / one frame")
    if nf == 1 and nt == 1:
        return None

    def tri(m):
        up = _ramp(m, dtype)
        return torch.cat([up, torch.ones(1, dtype=dtype), up.flip(0)])
    k = torch.outer(tri(nf), tri(nt))
    return (k / k.sum())[None, None]


@torch.no_grad()
def _mask(X, xn, win, nonstationary, n_fft, W, H, n_std, n_thresh, temp, n_movemean):
    if nonstationary:
        # torchgate.py:167-198: boxcar mean along time (zero padded), slowness ratio, temperature sigmoid
        A = X.abs()
        box = torch.ones(1, 1, n_movemean, dtype=A.dtype)
        S = F.conv1d(A.reshape(-1, 1, A.shape[-1]), box, padding="same").view(A.shape) / n_movemean
        return torch.sigmoid(((A - S) / S - n_thresh) / temp)
    # torchgate.py:126-165
    X_db = _db(X)
    if xn is not None:
        XN = torch.stft(xn, n_fft=n_fft, hop_length=H, win_length=W, return_complex=True, pad_mode="constant",
                        center=True, window=win)
        N_db = _db(XN).to(X_db.dtype)
    else:
        N_db = X_db
    std, mean = torch.std_mean(N_db, dim=-1)
    return torch.gt(X_db, (mean + std * n_std).unsqueeze(2))


def torchgate_cpu(x, sr, xn=None, nonstationary=False, n_std_thresh_stationary=1.5, n_thresh_nonstationary=1.3,
                  temp_coeff_nonstationary=0.1, n_movemean_nonstationary=20, prop_decrease=1.0, n_fft=1024,
                  win_length=None, hop_length=None, freq_mask_smooth_hz=500, time_mask_smooth_ms=50):
    """x: (B, L) CPU tensor (float32 or float64) -> (B, hop * (L // hop)) of the same dtype."""
    assert x.ndim == 2 and x.device.type == "cpu"
    W = n_fft if win_length is None else win_length
    H = W // 4 if hop_length is None else hop_length
    if x.shape[-1] < 2 * W:
        raise Exception(f"x must be bigger than {2 * W}")
    filt = smoothing_filter(sr, n_fft, H, freq_mask_smooth_hz, time_mask_smooth_ms)
    win = torch.hann_window(W)
    X = torch.stft(x, n_fft=n_fft, hop_length=H, win_length=W, return_complex=True, pad_mode="constant",
                   center=True, window=win)
    m = _mask(X, xn, win, nonstationary, n_fft, W, H, n_std_thresh_stationary, n_thresh_nonstationary,
              temp_coeff_nonstationary, n_movemean_nonstationary)
    m = prop_decrease * (m * 1.0 - 1.0) + 1.0            # torchgate.py:241
    if filt is not None:
        m = F.conv2d(m.unsqueeze(1), filt.to(m.dtype), padding="same").squeeze(1)
    y = torch.istft(X * m, n_fft=n_fft, hop_length=H, win_length=W, center=True, window=win)
    return y.to(x.dtype)
