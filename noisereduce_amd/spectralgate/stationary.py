"""Stationary spectral gate, mirror of
/root/reference/noisereduce/spectralgate/stationary.py."""
import numpy as np
import torch

from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.base import SpectralGate


class SpectralGateStationary(SpectralGate):
    def __init__(self, y, sr, y_noise, n_std_thresh_stationary, chunk_size,
                 clip_noise_stationary, padding, n_fft, win_length, hop_length, time_constant_s,
                 freq_mask_smooth_hz, time_mask_smooth_ms, tmp_folder, prop_decrease, use_tqdm,
                 n_jobs, device="cuda", slot=0, precision=None):
        super().__init__(y=y, sr=sr, chunk_size=chunk_size, padding=padding, n_fft=n_fft,
                         win_length=win_length, hop_length=hop_length,
                         time_constant_s=time_constant_s,
                         freq_mask_smooth_hz=freq_mask_smooth_hz,
                         time_mask_smooth_ms=time_mask_smooth_ms, tmp_folder=tmp_folder,
                         prop_decrease=prop_decrease, use_tqdm=use_tqdm, n_jobs=n_jobs,
                         device=device, precision=precision)
        self.n_std_thresh_stationary = n_std_thresh_stationary

        # noise clip, (channels, frames)  (stationary.py:47-58)
        if y_noise is None:
            # a host recording that get_traces will stream in pieces: the statistics only need its first chunk_size samples,
            # which arrive with the first piece
            x_dev = self._pipeline_begin() if clip_noise_stationary and chunk_size is not None else None
            noise_dev = x_dev if x_dev is not None else self._device_y()
        else:
            if not isinstance(y_noise, torch.Tensor):
                y_noise = np.asarray(y_noise)
            if len(y_noise.shape) == 1:
                y_noise = y_noise[None, :]
            elif len(y_noise.shape) > 2:
                raise ValueError("Waveform must be in shape (# frames, # channels)")
            noise_dev = self._to_device(y_noise)
        if clip_noise_stationary and chunk_size is not None:
            noise_dev = noise_dev[:, :chunk_size]          # stationary.py:63-64

        self._gate = _ffi.cached_gate(self.device, slot=slot, stationary=True,
                                      n_std_thresh=n_std_thresh_stationary, top_db=80.0, ddof=0,
                                      **self._gate_kwargs())
        # channel mean -> STFT -> dB -> per-band mean/std -> threshold (stationary.py:61-81),
        # all on the device; the result stays there.  The engine handle is shared by every object with
        # the same settings, the threshold is NOT: like the reference (self.noise_thresh,
        # stationary.py:79-81) it belongs to this object.  It stays IN the handle while this object is the last one to
        # have put one there; whoever overwrites it first saves a float64 device copy into _thr_dev (Gate.evict_threshold),
        # and _bind() loads that copy back when this object gates again.
        self._token = object()   # identity of this object's threshold
        self._thr_dev = None
        with self._gate.lock:
            self._gate.noise_stats(noise_dev)
            self._gate.claim_threshold(self, self._token)
        self._noise_thresh = None

    def _bind(self):
        """Make the shared handle hold THIS object's threshold (caller holds self._gate.lock)."""
        if self._gate.thresh_owner is not self._token:
            if self._thr_dev is None:
                raise RuntimeError("the engine handle lost this object's noise threshold (it was overwritten without "
                                   "Gate.evict_threshold)")
            self._gate.set_noise_threshold_tensor(self._thr_dev)
            self._gate.claim_threshold(self, self._token)

    @property
    def noise_thresh(self):
        """Per-band threshold in dB (stationary.py:79-81), fetched from the device on demand."""
        if self._noise_thresh is None:
            with self._gate.lock:
                t = self._thr_dev if self._thr_dev is not None else self._gate.noise_threshold_tensor()
            self._noise_thresh = t.cpu().numpy()
        return self._noise_thresh

    def spectral_gating_stationary(self, chunk):
        """(stationary.py:83-127)"""
        return self._do_filter(chunk)
