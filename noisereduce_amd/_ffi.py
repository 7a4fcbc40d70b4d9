"""ctypes binding of libmi355gate.so (C ABI: include/mi355gate.h; stage taps, development options and
per-kernel timing: include/mi355gate_debug.h).

PyTorch-ROCm tensors are only the I/O container: this module passes
``tensor.data_ptr()`` and the current HIP stream to the library.  There is no CPU
fallback: if the shared library is missing, or no MI355X/HIP device is visible, the
calls raise.
"""
import contextlib
import ctypes
import os
import threading
import weakref
from ctypes import (POINTER, Structure, byref, c_char_p, c_double, c_int, c_int32,
                    c_int64, c_void_p)

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355gate.so")

SG_F32, SG_F64, SG_I16, SG_I32 = 0, 1, 2, 3
SG_VARIANT_S, SG_VARIANT_T = 0, 1
SG_E_INVALID, SG_E_UNSUPPORTED, SG_E_HIP, SG_E_NOMEM, SG_E_STATE, SG_E_HANDOFF = -1, -2, -3, -4, -5, -6
SG_N_STAGES = 20
SG_OPT_FORCE_F64_DECIDE = 3
SG_OPT_FORCE_NOSEAM = 4
SG_OPT_FORCE_NOLEAN = 5
SG_OPT_FORCE_SPLIT = 6
SG_OPT_INJECT_HANDOFF_FAULT = 7
SG_OPT_FAST_INTEGER = 8
SG_OPT_FORCE_EXACT = 9
SG_OPT_FORCE_NOROWGATE = 10
SG_OPT_ROWGATE_TAP = 11
SG_OPT_ROWGATE_SHAPE = 12
SG_OPT_FLOOR_TEST = 13
SG_OPT_TILE_ORDER = 15
SG_OPT_EXACT_MATERIALISED = 16
SG_OPT_FORCE_UNFUSED = 1
SG_OPT_FORCE_NOFAST = 2

_TORCH_DTYPES = {torch.float32: SG_F32, torch.float64: SG_F64, torch.int16: SG_I16,
                 torch.int32: SG_I32}


class HandoffTimeout(RuntimeError):
    """SG_E_HANDOFF: a bounded inter-workgroup wait of a fused kernel timed out (the device was preempted for
    about a second); the outputs of the calls enqueued since the last check are invalid and must be re-run."""


class SgParams(Structure):
    """struct sg_params (include/mi355gate.h)."""
    _fields_ = [
        ("variant", c_int32), ("stationary", c_int32), ("n_fft", c_int32),
        ("win_length", c_int32), ("hop_length", c_int32), ("n_grad_freq", c_int32),
        ("n_grad_time", c_int32), ("smooth_mask", c_int32), ("chunk_size", c_int64),
        ("padding", c_int64), ("prop_decrease", c_double), ("n_std_thresh", c_double),
        ("top_db", c_double), ("ddof", c_int32), ("n_movemean", c_int32),
        ("nonstat_thresh", c_double), ("nonstat_slope", c_double), ("iir_b", c_double),
        ("max_workspace_bytes", c_int64),
    ]


# every symbol include/mi355gate.h and include/mi355gate_debug.h declare: name -> (restype, argtypes)
_PROTOTYPES = {
    "sg_version": (c_int, []),
    "sg_last_error": (c_char_p, [c_void_p]),
    "sg_create": (c_int, [POINTER(SgParams), POINTER(c_double), POINTER(c_void_p)]),
    "sg_destroy": (c_int, [c_void_p]),
    "sg_n_frames": (c_int, [c_void_p, c_int64, POINTER(c_int64)]),
    "sg_output_length": (c_int, [c_void_p, c_int64, POINTER(c_int64)]),
    "sg_workspace_bytes": (c_int, [c_void_p, c_int64, c_int64, c_int32, POINTER(c_int64)]),
    "sg_noise_stats": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p]),
    "sg_get_noise_threshold": (c_int, [c_void_p, POINTER(c_double), c_int32, c_void_p]),
    "sg_set_noise_threshold": (c_int, [c_void_p, POINTER(c_double), c_int32, c_void_p]),
    "sg_get_noise_threshold_dev": (c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    "sg_set_noise_threshold_dev": (c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    "sg_process_chunks": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64,
                                  c_int64, c_int64, c_int64, c_int64, c_int32, c_int64, c_int64,
                                  c_void_p]),
    "sg_filter_padded": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int64, c_int64,
                                 c_int64, c_int64, c_void_p]),
    "sg_process_batch": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p,
                                 c_int64, c_int64, c_int64, c_void_p, c_int, c_int64, c_void_p,
                                 c_void_p]),
    "sg_process_batch_backward": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64,
                                          c_void_p, c_void_p, c_int64, c_void_p]),
    "sg_stft": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "sg_set_option": (c_int, [c_void_p, c_int32, c_int64]),
    "sg_get_option": (c_int, [c_void_p, c_int32, POINTER(c_int64)]),
    "sg_check_errors": (c_int, [c_void_p, c_void_p]),
    "sg_profile_enable": (c_int, [c_void_p, c_int32]),
    "sg_profile_select": (c_int, [c_void_p, c_int64]),
    "sg_profile_read": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64), c_int32, c_int32]),
    "sg_stage_name": (c_char_p, [c_int32]),
    "sg_debug_dims": (c_int, [c_void_p, POINTER(c_int64)]),
    "sg_debug_range": (c_int, [c_void_p, POINTER(c_int64)]),
    "sg_debug_counter": (c_int, [c_void_p, c_int32, POINTER(c_int64), c_void_p]),
    "sg_debug_fetch": (c_int, [c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
}

_lib = None


def load_library():
    """dlopen libmi355gate.so and bind every declared symbol.  Loading needs the HIP
    runtime but no GPU; calling compute entry points does."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SG_LIB_PATH", LIB_PATH)   # development: A/B builds (tools/ablate.sh)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  noisereduce_amd has no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_PROTOTYPES)


def resolve_device(device="cuda"):
    """Return a torch.device on an AMD GPU or raise -- never falls back to CPU."""
    if not torch.cuda.is_available():
        raise RuntimeError(
            "noisereduce_amd needs a HIP device (MI355X / gfx950): torch.cuda.is_available() is "
            "False and there is no CPU fallback.")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"noisereduce_amd only runs on the GPU (got device={device!r})")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _sg_dtype(t):
    try:
        return _TORCH_DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported device sample dtype {t.dtype}") from None


def _rows(t):
    """(data_ptr, row_stride) of a 2-D tensor whose last dim is contiguous."""
    assert t.dim() == 2
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t, t.stride(0) if t.shape[0] > 1 else t.shape[1]


class Gate:
    """One sg_handle bound to one device (and used on that device's current stream)."""

    def __init__(self, device, *, variant, stationary, n_fft, win_length, hop_length,
                 n_grad_freq=1, n_grad_time=1, smooth_mask=False, chunk_size=600000,
                 padding=30000, prop_decrease=1.0, n_std_thresh=1.5, top_db=80.0, ddof=0,
                 n_movemean=20, nonstat_thresh=2.0, nonstat_slope=10.0, iir_b=0.0,
                 window=None, max_workspace_bytes=0, fast_integer=False, exact=False):
        self.lib = load_library()
        self.device = resolve_device(device)
        p = SgParams(variant=variant, stationary=int(bool(stationary)), n_fft=int(n_fft),
                     win_length=int(win_length), hop_length=int(hop_length),
                     n_grad_freq=int(n_grad_freq), n_grad_time=int(n_grad_time),
                     smooth_mask=int(bool(smooth_mask)), chunk_size=int(chunk_size or 1),
                     padding=int(padding or 0), prop_decrease=float(prop_decrease),
                     n_std_thresh=float(n_std_thresh), top_db=float(top_db), ddof=int(ddof),
                     n_movemean=int(n_movemean), nonstat_thresh=float(nonstat_thresh),
                     nonstat_slope=float(nonstat_slope), iir_b=float(iir_b),
                     max_workspace_bytes=int(max_workspace_bytes))
        self.params = p
        self.n_bins = int(n_fft) // 2 + 1
        # A handle carries per-call state (workspace, the stationary threshold h->thresh): callers that
        # share a cached handle serialise on `lock`, and `thresh_owner` names the object whose threshold
        # the handle currently holds (None: unknown) -- see SpectralGateStationary._bind().
        self.lock = threading.RLock()
        self.thresh_owner = None
        self._thresh_owner_ref = None    # weak reference to the owning object while its only copy is the handle's (claim_threshold)
        self._thr_stream = None          # stream the handle's current threshold was written on (noise_stats / set_noise_threshold*)
        wptr = None
        if window is not None:
            w = np.ascontiguousarray(np.asarray(window, dtype=np.float64))
            if w.shape != (int(win_length),):
                raise ValueError("window must have win_length entries")
            wptr = w.ctypes.data_as(POINTER(c_double))
        self._h = c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.sg_create(byref(p), wptr, byref(self._h))
        if rc != 0:
            msg = self.lib.sg_last_error(None).decode()
            self._h = c_void_p()
            self._raise(rc, msg)
        if fast_integer:   # integer outputs from the fused float32 kernels (<= 1 LSB off) instead of the float64 pipeline
            self.set_option(SG_OPT_FAST_INTEGER, 1)
        if exact:          # float64 pipeline for every sample type (precision="float64": float64-accurate results)
            self.set_option(SG_OPT_FORCE_EXACT, 1)

    # -- plumbing --------------------------------------------------------------------
    @staticmethod
    def _raise(rc, msg):
        if rc == SG_E_INVALID:
            raise ValueError(msg)
        if rc == SG_E_UNSUPPORTED:
            raise NotImplementedError(msg)
        if rc == SG_E_NOMEM:
            raise MemoryError(msg)
        if rc == SG_E_HANDOFF:
            raise HandoffTimeout(msg)
        raise RuntimeError(f"libmi355gate error {rc}: {msg}")

    def _check(self, rc):
        if rc != 0:
            self._raise(rc, self.lib.sg_last_error(self._h).decode())

    def _stream(self):
        return c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            with torch.cuda.device(self.device):
                self.lib.sg_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _on_device(self, t):
        if not isinstance(t, torch.Tensor) or t.device != self.device:
            raise ValueError(f"expected a tensor on {self.device}")

    # -- geometry --------------------------------------------------------------------
    def n_frames(self, L):
        out = c_int64()
        self._check(self.lib.sg_n_frames(self._h, int(L), byref(out)))
        return out.value

    def output_length(self, L):
        out = c_int64()
        self._check(self.lib.sg_output_length(self._h, int(L), byref(out)))
        return out.value

    def workspace_bytes(self, C, N, chunked=True):
        """HBM the handle will own after process_chunks on a (C, N) recording (host arithmetic only)."""
        out = c_int64()
        self._check(self.lib.sg_workspace_bytes(self._h, int(C), int(N), int(bool(chunked)), byref(out)))
        return out.value

    # -- variant S -------------------------------------------------------------------
    # The threshold belongs to the object that computed it (stationary.py:79-81), the handle is shared.  An object leaves its
    # threshold IN the handle and takes a device copy only when somebody else is about to overwrite it (evict_threshold):
    # the usual life -- one object, statistics, gate, gone -- never copies (a 4 KB device copy is a 4.8 us slot in the
    # stream of a 0.29 ms call).
    def claim_threshold(self, obj, token):
        """`obj` (with attributes _token, _thr_dev) owns the threshold the handle holds now.  Caller holds the lock."""
        self.thresh_owner = token
        self._thresh_owner_ref = weakref.ref(obj)

    def evict_threshold(self):
        """Before the handle's threshold is overwritten: its owner, if still alive and without a copy, saves one."""
        with self.lock:
            ref, self._thresh_owner_ref = self._thresh_owner_ref, None
            obj = ref() if ref is not None else None
            if obj is not None and obj._thr_dev is None and self.thresh_owner is obj._token:
                # The copy runs on the EVICTING caller's stream; the threshold was written on the owner's.  One stream (the
                # usual case): ordered by the stream, nothing to do.  Two: the copy waits for the writer's stream, the
                # owner's stream -- which will read the copy later -- waits for the copy, and the caching allocator is told
                # (costs nothing on the single-stream path)
                with torch.cuda.device(self.device):
                    cur, src = torch.cuda.current_stream(self.device), self._thr_stream
                    if src is not None and src != cur:
                        cur.wait_stream(src)
                    obj._thr_dev = self.noise_threshold_tensor()
                    if src is not None and src != cur:
                        src.wait_stream(cur)
                        obj._thr_dev.record_stream(src)
            self.thresh_owner = None

    def noise_stats(self, noise):
        self.evict_threshold()
        self._on_device(noise)
        noise, stride = _rows(noise)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_noise_stats(self._h, noise.data_ptr(), _sg_dtype(noise),
                                                noise.shape[0], noise.shape[1], stride,
                                                self._stream()))
            self._thr_stream = torch.cuda.current_stream(self.device)

    def get_noise_threshold(self):
        out = np.empty(self.n_bins, dtype=np.float64)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_get_noise_threshold(
                self._h, out.ctypes.data_as(POINTER(c_double)), self.n_bins, self._stream()))
        return out

    def set_noise_threshold(self, thresh):
        self.evict_threshold()
        t = np.ascontiguousarray(np.asarray(thresh, dtype=np.float64))
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_set_noise_threshold(
                self._h, t.ctypes.data_as(POINTER(c_double)), int(t.shape[0]), self._stream()))
            self._thr_stream = torch.cuda.current_stream(self.device)

    def noise_threshold_tensor(self):
        """The threshold as a float64 device tensor (no host synchronisation)."""
        t = torch.empty(self.n_bins, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_get_noise_threshold_dev(self._h, t.data_ptr(), self.n_bins,
                                                            self._stream()))
        return t

    def set_noise_threshold_tensor(self, t):
        self.evict_threshold()
        self._on_device(t)
        t = t.to(torch.float64).contiguous()
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_set_noise_threshold_dev(self._h, t.data_ptr(), int(t.numel()),
                                                            self._stream()))
            self._thr_stream = torch.cuda.current_stream(self.device)

    def process_chunks(self, x, out_dtype=None, start_frame=0, end_frame=None, chunked=True,
                       out=None, halo_left=0, halo_right=0):
        """x: (C, N) tensor, or -- with halos -- a (C, halo_left + N + halo_right) tensor whose
        columns [halo_left, halo_left + N) are this shard's samples."""
        self._on_device(x)
        x, stride = _rows(x)
        C = x.shape[0]
        N = x.shape[1] - halo_left - halo_right
        x_ptr = x.data_ptr() + halo_left * x.element_size()
        end_frame = N if end_frame is None else int(end_frame)
        n_out = end_frame if not chunked else end_frame - int(start_frame)
        if out is None:
            out = torch.empty((C, n_out), dtype=out_dtype or x.dtype, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_process_chunks(
                self._h, x_ptr, _sg_dtype(x), out.data_ptr(), _sg_dtype(out), C, N, stride,
                out.stride(0) if C > 1 else n_out, int(start_frame), end_frame, int(bool(chunked)),
                int(halo_left), int(halo_right), self._stream()))
        return out

    def filter_padded(self, chunk, out_dtype=None):
        self._on_device(chunk)
        chunk, stride = _rows(chunk)
        C, Lp = chunk.shape
        out = torch.empty((C, Lp), dtype=out_dtype or chunk.dtype, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_filter_padded(self._h, chunk.data_ptr(), _sg_dtype(chunk),
                                                  out.data_ptr(), _sg_dtype(out), C, Lp, stride, Lp,
                                                  self._stream()))
        return out

    # -- variant T -------------------------------------------------------------------
    def process_batch(self, x, xn=None, out_dtype=None, save_mask=False):
        """TorchGate.forward.  With save_mask=True also returns the final mask
        (B, T, FS) float32 needed by process_batch_backward."""
        self._on_device(x)
        x, xs = _rows(x)
        B, L = x.shape
        Lout = self.output_length(L)
        out = torch.empty((B, Lout), dtype=out_dtype or x.dtype, device=self.device)
        mask = None
        if save_mask:
            FS = (self.n_bins + 15) // 16 * 16
            mask = torch.empty((B, self.n_frames(L), FS), dtype=torch.float32, device=self.device)
        if xn is not None:
            self._on_device(xn)
            if xn.dtype != x.dtype:
                xn = xn.to(x.dtype)
            xn, xns = _rows(xn)
            xn_ptr, Bn, Ln = xn.data_ptr(), xn.shape[0], xn.shape[1]
        else:
            xn_ptr, Bn, Ln, xns = None, 0, 0, 0
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_process_batch(
                self._h, x.data_ptr(), _sg_dtype(x), B, L, xs, xn_ptr, Bn, Ln, xns, out.data_ptr(),
                _sg_dtype(out), Lout, None if mask is None else mask.data_ptr(), self._stream()))
        return (out, mask) if save_mask else out

    def process_batch_backward(self, grad_out, mask, L):
        """Adjoint of process_batch with the mask fixed: (B, Lout) -> (B, L)."""
        self._on_device(grad_out)
        grad_out, gs = _rows(grad_out)
        B = grad_out.shape[0]
        gx = torch.empty((B, L), dtype=grad_out.dtype, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_process_batch_backward(
                self._h, grad_out.data_ptr(), _sg_dtype(grad_out), B, L, gs, mask.data_ptr(),
                gx.data_ptr(), L, self._stream()))
        return gx

    def set_option(self, option, value):
        self._check(self.lib.sg_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        """The option's current value on this (shared, cached) handle, read from the handle itself (sg_get_option):
        the library's default if it was never set -- not every default is 0 (SG_OPT_ROWGATE_SHAPE: 16)."""
        v = c_int64(0)
        rc = self.lib.sg_get_option(self._h, int(option), byref(v))
        if rc != 0:   # (the handle is const in sg_get_option: the library leaves no message of its own)
            raise ValueError(f"sg_get_option: unknown option {int(option)} (rc {rc})")
        return int(v.value)

    @contextlib.contextmanager
    def with_options(self, pairs):
        """`with g.with_options([(SG_OPT_X, 1), ...]):` -- sets the options and RESTORES their previous values (the
        handle is shared between objects: a user's or a test's setting must survive somebody else's retry)."""
        prev = [(o, self.get_option(o)) for o, _ in pairs]
        try:
            for o, v in pairs:
                self.set_option(o, v)
            yield self
        finally:
            for o, v in prev:
                self.set_option(o, v)

    def check_errors(self):
        """Synchronise the current stream and raise HandoffTimeout if a launch enqueued on this handle since the
        last check lost an in-launch hand-off (include/mi355gate.h: sg_check_errors)."""
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_check_errors(self._h, self._stream()))

    def run_checked(self, fn):
        """fn() -> device tensor, then check_errors(); a call that lost a hand-off is re-run ONCE on the kernels
        without in-launch hand-offs (three-kernel gate, apply + seam kernel).  For callers that synchronise
        anyway (host arrays out); holds the handle's lock."""
        with self.lock:
            # an error left behind by an EARLIER unchecked (tensor-in / tensor-out) call on this shared handle belongs to
            # that call: report it as such instead of "fixing" it with a retry of this one
            try:
                self.check_errors()
            except HandoffTimeout as e:
                raise HandoffTimeout("an earlier, unchecked call on this engine handle lost a tile hand-off (its output "
                                     "is invalid); this call has not run: " + str(e)) from None
            try:
                out = fn()
                self.check_errors()
                return out
            except HandoffTimeout:
                with self.with_options([(SG_OPT_FORCE_SPLIT, 1), (SG_OPT_FORCE_NOLEAN, 1)]):
                    out = fn()
                    self.check_errors()
                    return out

    # -- per-kernel timing -----------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.sg_profile_enable(self._h, int(bool(on))))

    def profile_select(self, stage_names=None):
        """Time only the named stages (names as returned by profile_read); None = all."""
        mask = 0
        if stage_names:
            names = [self.lib.sg_stage_name(i).decode() for i in range(SG_N_STAGES)]
            for n in stage_names:
                mask |= 1 << names.index(n)
        self._check(self.lib.sg_profile_select(self._h, mask))

    def profile_read(self, reset=True):
        """{stage name: (total ms, launches)} accumulated since the last reset (synchronises)."""
        ms = (c_double * SG_N_STAGES)()
        cnt = (c_int64 * SG_N_STAGES)()
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_profile_read(self._h, ms, cnt, SG_N_STAGES, int(bool(reset))))
        return {self.lib.sg_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(SG_N_STAGES)
                if cnt[i]}

    # -- stage taps ------------------------------------------------------------------
    def stft(self, x):
        """(B, L) -> complex128 (B, T, F), scaled like the variant's reference STFT."""
        self._on_device(x)
        x, xs = _rows(x)
        B, L = x.shape
        T = self.n_frames(L)
        z = torch.empty((B, T, self.n_bins, 2), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_stft(self._h, x.data_ptr(), _sg_dtype(x), B, L, xs, z.data_ptr(),
                                         self._stream()))
        return torch.view_as_complex(z)

    def debug_range(self):
        r = (c_int64 * 2)()
        self._check(self.lib.sg_debug_range(self._h, r))
        return int(r[0]), int(r[1])

    def debug_counter(self, which=0, value=0):
        """0: (row, band) pairs the row gate re-evaluated in float64 since the handle was created; 1 / 2: batches of the
        one-pass gate that took the in-kernel / the a-priori floor test (SG_OPT_FLOOR_TEST); 3: launch epoch of the last gate
        call in which a chunk's floor test fired."""
        v = c_int64(int(value))   # (in: an argument of the development counters; 0 for the documented ones)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_debug_counter(self._h, int(which), ctypes.byref(v), self._stream()))
        return int(v.value)

    def debug_field(self, what):
        """0: raw mask, 1: final mask (float32); 2: power (float64) of the last unit batch,
        as (units, T, F) numpy arrays."""
        dims = (c_int64 * 3)()
        self._check(self.lib.sg_debug_dims(self._h, dims))
        units, T, FS = dims[0], dims[1], dims[2]
        if what == 4:  # row gate: float32 power tile (4 |X|^2) of pass 1, (units, 64, 528); rows >= T are scratch
            host = np.empty((units, 64, 528), dtype=np.float32)
            with torch.cuda.device(self.device):
                self._check(self.lib.sg_debug_fetch(self._h, 4, host.ctypes.data_as(c_void_p), host.nbytes,
                                                    self._stream()))
            return host[:, :T, :self.n_bins]
        if what == 3:  # bit field -> boolean (units, T, F)
            wpr = (self.n_bins + 63) // 64
            words = np.empty((units, T, wpr), dtype=np.uint64)
            with torch.cuda.device(self.device):
                self._check(self.lib.sg_debug_fetch(self._h, 3, words.ctypes.data_as(c_void_p),
                                                    words.nbytes, self._stream()))
            b = np.unpackbits(words.view(np.uint8), axis=-1, bitorder="little")
            return b[:, :, :self.n_bins].astype(bool)
        dt = np.float64 if what == 2 else np.float32
        host = np.empty((units, T, FS), dtype=dt)
        with torch.cuda.device(self.device):
            self._check(self.lib.sg_debug_fetch(self._h, int(what), host.ctypes.data_as(c_void_p),
                                                host.nbytes, self._stream()))
        return host[:, :, :self.n_bins]


# Handles are cached per (device, parameters): a handle owns its twiddle/window tables and a
# grown-on-demand workspace, so repeated reduce_noise()/TorchGate calls with the same settings
# reuse them instead of paying hipMalloc/hipFree per call.  The cache itself is guarded by
# _CACHE_LOCK; a handle is NOT re-entrant: users hold Gate.lock around a call sequence that depends on
# handle state (noise statistics -> filter).  The stationary threshold is per OBJECT (reference
# stationary.py:79-81), never per handle: objects re-load theirs when Gate.thresh_owner is not them.
_GATE_CACHE = {}
_CACHE_LOCK = threading.Lock()


def cached_gate(device, slot=0, **kw):
    """`slot` distinguishes handles with identical parameters: calls that are in flight at the same time
    on different streams must not share a handle (its workspace is per call)."""
    dev = resolve_device(device)
    # integer recordings: the reference truncates a float64 result (base.py:217-226) -> float64 pipeline by default;
    # NOISEREDUCE_AMD_FAST_INT=1 keeps the fused float32 kernels (<= 1 LSB off on ~1 % of the samples)
    kw.setdefault("fast_integer", os.environ.get("NOISEREDUCE_AMD_FAST_INT", "0") == "1")
    # precision="float64" / NOISEREDUCE_AMD_EXACT=1: the float64 pipeline for every sample type -- a float64 recording then
    # gets float64-accurate results like the reference's (base.py:140 computes every dtype in float64) instead of
    # float32-accurate ones in a float64 container (the default: 14 x faster, 2e-7 of peak off)
    if kw.get("exact") is None:
        kw["exact"] = os.environ.get("NOISEREDUCE_AMD_EXACT", "0") == "1"

    def norm(v):
        if isinstance(v, np.ndarray):
            return ("nd", v.shape, v.tobytes())
        if isinstance(v, (bool, np.bool_)):
            return bool(v)
        if isinstance(v, (int, np.integer)):
            return int(v)
        if isinstance(v, (float, np.floating)):
            return float(v)
        return v
    key = (dev.index, int(slot)) + tuple(sorted((k, norm(v)) for k, v in kw.items()))
    with _CACHE_LOCK:
        g = _GATE_CACHE.get(key)
        if g is None:
            if len(_GATE_CACHE) >= 8:  # bound the number of cached workspaces
                # dropped, not closed: an object that still holds the evicted gate keeps it alive, and
                # Gate.__del__ frees the handle with the last reference
                _GATE_CACHE.pop(next(iter(_GATE_CACHE)))
            g = Gate(dev, **kw)
            _GATE_CACHE[key] = g
        return g


def clear_gate_cache():
    with _CACHE_LOCK:
        while _GATE_CACHE:
            _GATE_CACHE.popitem()[1].close()
