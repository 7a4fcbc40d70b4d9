"""Non-stationary spectral gate, mirror of
/root/reference/noisereduce/spectralgate/nonstationary.py."""
import numpy as np

from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.base import SpectralGate


def iir_coefficient(time_constant_s, samplerate, hop_length):
    """One-pole coefficient of get_time_smoothed_representation (nonstationary.py:106-114)."""
    t_frames = time_constant_s * samplerate / float(hop_length)
    return (np.sqrt(1 + 4 * t_frames ** 2) - 1) / (2 * t_frames ** 2)


class SpectralGateNonStationary(SpectralGate):
    def __init__(self, y, sr, chunk_size, padding, n_fft, win_length, hop_length,
                 time_constant_s, freq_mask_smooth_hz, time_mask_smooth_ms,
                 thresh_n_mult_nonstationary, sigmoid_slope_nonstationary, tmp_folder,
                 prop_decrease, use_tqdm, n_jobs, device="cuda", precision=None):
        self._thresh_n_mult_nonstationary = thresh_n_mult_nonstationary
        self._sigmoid_slope_nonstationary = sigmoid_slope_nonstationary
        super().__init__(y=y, sr=sr, chunk_size=chunk_size, padding=padding, n_fft=n_fft,
                         win_length=win_length, hop_length=hop_length,
                         time_constant_s=time_constant_s,
                         freq_mask_smooth_hz=freq_mask_smooth_hz,
                         time_mask_smooth_ms=time_mask_smooth_ms, tmp_folder=tmp_folder,
                         prop_decrease=prop_decrease, use_tqdm=use_tqdm, n_jobs=n_jobs,
                         device=device, precision=precision)
        b = iir_coefficient(self._time_constant_s, self.sr, self._hop_length)
        self._gate = _ffi.cached_gate(self.device, stationary=False, iir_b=b,
                               nonstat_thresh=thresh_n_mult_nonstationary,
                               nonstat_slope=sigmoid_slope_nonstationary, **self._gate_kwargs())

    def spectral_gating_nonstationary(self, chunk):
        """(nonstationary.py:47-97)"""
        return self._do_filter(chunk)
