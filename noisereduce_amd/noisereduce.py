"""``reduce_noise`` -- same signature as /root/reference/noisereduce/noisereduce.py:13-36.

All work runs on the GPU named by ``device`` (default ``"cuda"``, i.e. the MI355X the
process sees); there is no CPU path.  ``y`` may be array-like (result: numpy array of the
input dtype and shape, like the reference) or a torch tensor (result: tensor on the GPU;
use this to keep long recordings resident in HBM).
"""
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary


def reduce_noise(y, sr, stationary=False, y_noise=None, prop_decrease=1.0, time_constant_s=2.0,
                 freq_mask_smooth_hz=500, time_mask_smooth_ms=50, thresh_n_mult_nonstationary=2,
                 sigmoid_slope_nonstationary=10, n_std_thresh_stationary=1.5, tmp_folder=None,
                 chunk_size=600000, padding=30000, n_fft=1024, win_length=None, hop_length=None,
                 clip_noise_stationary=True, use_tqdm=False, n_jobs=1, use_torch=False,
                 device="cuda", precision=None):
    """Reduce noise via spectral gating (see the reference docstring,
    noisereduce.py:37-109, for the meaning of every argument).

    ``precision`` (not in the reference, ``use_torch=False`` only): ``"float32"`` -- the fused float32 kernels (mask
    decisions exact, output within 2e-7 of peak of the reference's float64 arithmetic, whatever the container dtype);
    ``"float64"`` -- the float64 pipeline (the reference computes every dtype in float64, base.py:140: a float64
    recording then matches it to ~1e-13; about 14 x slower); ``None`` (default) -- ``"float64"`` if the environment
    holds ``NOISEREDUCE_AMD_EXACT=1``, else ``"float32"``.  Integer recordings are bit-exact either way.

    ``use_torch=False`` evaluates the numpy/scipy "spectralgate" algorithm,
    ``use_torch=True`` the "torchgate" algorithm (the two differ)."""
    if precision not in (None, "float32", "float64"):
        raise ValueError("precision must be None, 'float32' or 'float64'")
    if use_torch:
        if precision == "float64":
            # (the torchgate algorithm computes in float32 in the reference too, torchgate.py:200-264: asking for float64
            # arithmetic there must not be silently ignored)
            raise ValueError("precision='float64' applies to use_torch=False only (the torchgate algorithm is float32)")
        if n_jobs != 1:
            raise ValueError("n_jobs must be 1 when using torch version of spectral gating.")
        from noisereduce_amd.spectralgate.streamed_torch_gate import StreamedTorchGate
        sg = StreamedTorchGate(
            y=y, sr=sr, stationary=stationary, y_noise=y_noise, prop_decrease=prop_decrease,
            time_constant_s=time_constant_s, freq_mask_smooth_hz=freq_mask_smooth_hz,
            time_mask_smooth_ms=time_mask_smooth_ms,
            thresh_n_mult_nonstationary=thresh_n_mult_nonstationary,
            sigmoid_slope_nonstationary=sigmoid_slope_nonstationary, tmp_folder=tmp_folder,
            chunk_size=chunk_size, padding=padding, n_fft=n_fft, win_length=win_length,
            hop_length=hop_length, clip_noise_stationary=clip_noise_stationary,
            use_tqdm=use_tqdm, n_jobs=n_jobs, device=device)
    elif stationary:
        sg = SpectralGateStationary(
            y=y, sr=sr, y_noise=y_noise, prop_decrease=prop_decrease,
            n_std_thresh_stationary=n_std_thresh_stationary, chunk_size=chunk_size,
            clip_noise_stationary=clip_noise_stationary, padding=padding, n_fft=n_fft,
            win_length=win_length, hop_length=hop_length, time_constant_s=time_constant_s,
            freq_mask_smooth_hz=freq_mask_smooth_hz, time_mask_smooth_ms=time_mask_smooth_ms,
            tmp_folder=tmp_folder, use_tqdm=use_tqdm, n_jobs=n_jobs, device=device, precision=precision)
    else:
        sg = SpectralGateNonStationary(
            y=y, sr=sr, chunk_size=chunk_size, padding=padding, prop_decrease=prop_decrease,
            n_fft=n_fft, win_length=win_length, hop_length=hop_length,
            time_constant_s=time_constant_s, freq_mask_smooth_hz=freq_mask_smooth_hz,
            time_mask_smooth_ms=time_mask_smooth_ms,
            thresh_n_mult_nonstationary=thresh_n_mult_nonstationary,
            sigmoid_slope_nonstationary=sigmoid_slope_nonstationary, tmp_folder=tmp_folder,
            use_tqdm=use_tqdm, n_jobs=n_jobs, device=device, precision=precision)
    return sg.get_traces()
