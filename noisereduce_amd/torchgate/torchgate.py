"""``TorchGate`` -- nn.Module with the constructor, buffer and forward contract of
/root/reference/noisereduce/torchgate/torchgate.py:7-264, evaluated by the HIP engine
(sg_process_batch).  0 parameters, one buffer ``smoothing_filter`` of shape
(1, 1, 2*n_grad_freq+1, 2*n_grad_time+1) (or None), so state_dicts round-trip with the
reference's.  The mask is computed without gradient (torchgate.py:126,167) and the output
is differentiable w.r.t. ``x`` through STFT -> (x mask) -> ISTFT.
"""
from typing import Optional, Union

import torch

from noisereduce_amd import _ffi
from noisereduce_amd.torchgate.utils import linspace


class _GateFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, xn, module):
        gate = module._gate_for(x.device)
        if ctx.needs_input_grad[0]:
            y, mask = gate.process_batch(x.detach(), xn, save_mask=True)
            ctx.gate = gate
            ctx.L = x.shape[-1]
            ctx.save_for_backward(mask)
        else:
            y = gate.process_batch(x, xn)
        return y

    @staticmethod
    def backward(ctx, grad_out):
        (mask,) = ctx.saved_tensors
        gx = ctx.gate.process_batch_backward(grad_out.contiguous(), mask, ctx.L)
        return gx, None, None


class TorchGate(torch.nn.Module):
    """A PyTorch module that applies a spectral gate to an input signal (see the reference
    docstring, torchgate.py:8-29, for the arguments)."""

    @torch.no_grad()
    def __init__(self, sr: int, nonstationary: bool = False, n_std_thresh_stationary: float = 1.5,
                 n_thresh_nonstationary: float = 1.3, temp_coeff_nonstationary: float = 0.1,
                 n_movemean_nonstationary: int = 20, prop_decrease: float = 1.0, n_fft: int = 1024,
                 win_length: int = None, hop_length: int = None, freq_mask_smooth_hz: float = 500,
                 time_mask_smooth_ms: float = 50):
        super().__init__()
        self.sr = sr
        self.nonstationary = nonstationary
        assert 0.0 <= prop_decrease <= 1.0
        self.prop_decrease = prop_decrease
        self.n_fft = n_fft
        self.win_length = self.n_fft if win_length is None else win_length
        self.hop_length = self.win_length // 4 if hop_length is None else hop_length
        self.n_std_thresh_stationary = n_std_thresh_stationary
        self.temp_coeff_nonstationary = temp_coeff_nonstationary
        self.n_movemean_nonstationary = n_movemean_nonstationary
        self.n_thresh_nonstationary = n_thresh_nonstationary
        self.freq_mask_smooth_hz = freq_mask_smooth_hz
        self.time_mask_smooth_ms = time_mask_smooth_ms
        self._n_grad = (1, 1)
        self.register_buffer("smoothing_filter", self._generate_mask_smoothing_filter())
        self._gates = {}

    @torch.no_grad()
    def _generate_mask_smoothing_filter(self) -> Union[torch.Tensor, None]:
        """(torchgate.py:73-124).  The reference's error path for a too-small
        freq_mask_smooth_hz dereferences a missing attribute (torchgate.py:94); here it
        raises the ValueError it meant to."""
        if self.freq_mask_smooth_hz is None and self.time_mask_smooth_ms is None:
            return None
        n_grad_freq = (1 if self.freq_mask_smooth_hz is None
                       else int(self.freq_mask_smooth_hz / (self.sr / (self.n_fft / 2))))
        if n_grad_freq < 1:
            raise ValueError(
                f"freq_mask_smooth_hz needs to be at least {int((self.sr / (self.n_fft / 2)))} Hz")
        n_grad_time = (1 if self.time_mask_smooth_ms is None
                       else int(self.time_mask_smooth_ms / ((self.hop_length / self.sr) * 1000)))
        if n_grad_time < 1:
            raise ValueError(
                f"time_mask_smooth_ms needs to be at least {int((self.hop_length / self.sr) * 1000)} ms")
        if n_grad_time == 1 and n_grad_freq == 1:
            return None
        self._n_grad = (n_grad_freq, n_grad_time)
        v_f = torch.cat([linspace(0, 1, n_grad_freq + 1, endpoint=False),
                         linspace(1, 0, n_grad_freq + 2)])[1:-1]
        v_t = torch.cat([linspace(0, 1, n_grad_time + 1, endpoint=False),
                         linspace(1, 0, n_grad_time + 2)])[1:-1]
        smoothing_filter = torch.outer(v_f, v_t).unsqueeze(0).unsqueeze(0)
        return smoothing_filter / smoothing_filter.sum()

    def _gate_for(self, device):
        key = (device.type, device.index)
        g = self._gates.get(key)
        if g is None:
            # The reference builds its Hann window in float32 whatever the input dtype
            # (torchgate.py:150,231,261); hand the engine the same table.
            window = torch.hann_window(self.win_length).double().numpy()
            nf, nt = self._n_grad
            g = _ffi.cached_gate(device, variant=_ffi.SG_VARIANT_T, stationary=not self.nonstationary,
                          n_fft=self.n_fft, win_length=self.win_length, hop_length=self.hop_length,
                          n_grad_freq=nf, n_grad_time=nt,
                          smooth_mask=self.smoothing_filter is not None,
                          prop_decrease=self.prop_decrease,
                          n_std_thresh=self.n_std_thresh_stationary, top_db=40.0, ddof=1,
                          n_movemean=self.n_movemean_nonstationary,
                          nonstat_thresh=self.n_thresh_nonstationary,
                          nonstat_slope=1.0 / self.temp_coeff_nonstationary, window=window)
            self._gates[key] = g
        return g

    def forward(self, x: torch.Tensor, xn: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: (batch, signal_length); xn: optional noise signal(s) for the stationary
        statistics.  Returns (batch, hop*(signal_length//hop)) in x.dtype
        (torchgate.py:200-264)."""
        assert x.ndim == 2
        if x.shape[-1] < self.win_length * 2:
            raise Exception(f"x must be bigger than {self.win_length * 2}")
        assert xn is None or xn.ndim == 1 or xn.ndim == 2
        if xn is not None and xn.shape[-1] < self.win_length * 2:
            raise Exception(f"xn must be bigger than {self.win_length * 2}")
        if x.device.type != "cuda":
            raise RuntimeError("noisereduce_amd.TorchGate runs on the GPU only (no CPU fallback): "
                               "move the input with x.to('cuda')")
        dtype = x.dtype
        if dtype not in (torch.float32, torch.float64):
            x = x.float()
        if xn is not None:
            xn = xn.detach().to(device=x.device, dtype=x.dtype)
            if xn.ndim == 1:
                # the reference crashes on 1-D xn in stationary mode (torchgate.py:164); a
                # single noise row is the evident intent.
                xn = xn.unsqueeze(0)
            if self.nonstationary:
                xn = None  # unused by the non-stationary mask (torchgate.py:235-236)
        y = _GateFunction.apply(x, xn, self)
        return y.to(dtype=dtype)

    def __getstate__(self):
        st = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        st = dict(st)
        st["_gates"] = {}  # device handles are not picklable
        return st
