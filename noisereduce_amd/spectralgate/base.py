"""Host-side mirror of the reference chunk streamer
(/root/reference/noisereduce/spectralgate/base.py).

Same constructor arguments, attributes and method names as the reference's
``SpectralGate``; the difference is where the work happens: the recording is uploaded
once and the whole chunk grid of ``get_traces`` (base.py:167-226) -- every
(channel, chunk) unit -- is evaluated on the GPU by ``sg_process_chunks``; there is no
joblib pool and no memmap tempfile (``n_jobs``, ``tmp_folder`` and ``use_tqdm`` are
accepted and ignored).  ``_do_filter`` stays the operator seam (base.py:158-160).
"""
import numpy as np
import torch

from noisereduce_amd import _ffi, _hostbuf


def _triangle(m):
    # base.py:17-27 one axis: [1..m, m+1, m..1] / (m+1)
    up = np.arange(1, m + 2, dtype=np.float64)
    return np.concatenate([up, up[-2::-1]]) / (m + 1)


def _smoothing_filter(n_grad_freq, n_grad_time):
    """Mask smoothing filter (base.py:7-29): outer product of two triangles, unit sum."""
    f = np.outer(_triangle(n_grad_freq), _triangle(n_grad_time))
    return f / np.sum(f)


_PIPE_STREAMS = {}


def _pipe_stream(dev):
    """The upload side stream of a device, created once."""
    key = (dev.type, dev.index)
    st = _PIPE_STREAMS.get(key)
    if st is None:
        st = _PIPE_STREAMS[key] = torch.cuda.Stream(dev)
    return st


_DEVICE_DTYPES = {
    np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
    np.dtype(np.int16): torch.int16, np.dtype(np.int32): torch.int32,
}


class SpectralGate:
    def __init__(self, y, sr, prop_decrease, chunk_size, padding, n_fft, win_length,
                 hop_length, time_constant_s, freq_mask_smooth_hz, time_mask_smooth_ms,
                 tmp_folder, use_tqdm, n_jobs, device="cuda", precision=None):
        self.sr = sr
        if precision not in (None, "float32", "float64"):
            raise ValueError('precision must be None, "float32" or "float64"')
        # None: NOISEREDUCE_AMD_EXACT decides (default: the fused float32 kernels); "float64": the float64 pipeline
        self._exact = None if precision is None else precision == "float64"
        self.flat = False
        self._tensor_io = isinstance(y, torch.Tensor)
        if not self._tensor_io:
            y = np.asarray(y)  # read-only here: no need for the reference's defensive copy (base.py:54)
        # reshape data to (#channels, #frames)  (base.py:54-62)
        if len(y.shape) == 1:
            self.y = y[None, :]
            self.flat = True
        elif len(y.shape) > 2:
            raise ValueError("Waveform must be in shape (# frames, # channels)")
        else:
            self.y = y
        self._dtype = y.dtype
        self.n_channels, self.n_frames = self.y.shape
        self._chunk_size = chunk_size
        self.padding = padding
        self.n_jobs = n_jobs
        self.use_tqdm = use_tqdm
        self._tmp_folder = tmp_folder

        # STFT parameters (base.py:77-86)
        self._n_fft = n_fft
        self._win_length = self._n_fft if win_length is None else win_length
        self._hop_length = self._win_length // 4 if hop_length is None else hop_length
        self._time_constant_s = time_constant_s
        self._prop_decrease = prop_decrease

        self._n_grad_freq = 1
        self._n_grad_time = 1
        if (freq_mask_smooth_hz is None) & (time_mask_smooth_ms is None):
            self.smooth_mask = False
        else:
            self._generate_mask_smoothing_filter(freq_mask_smooth_hz, time_mask_smooth_ms)

        self.device = _ffi.resolve_device(device)
        self._y_dev = None
        self._pipe = None      # state of a pipelined upload in progress (_pipeline_begin)
        self._gate = None

    # -- filter design (base.py:99-128) ---------------------------------------------------
    def _generate_mask_smoothing_filter(self, freq_mask_smooth_hz, time_mask_smooth_ms):
        if freq_mask_smooth_hz is None:
            n_grad_freq = 1
        else:
            n_grad_freq = int(freq_mask_smooth_hz / (self.sr / (self._n_fft / 2)))
            if n_grad_freq < 1:
                raise ValueError("freq_mask_smooth_hz needs to be at least {}Hz".format(
                    int((self.sr / (self._n_fft / 2)))))
        if time_mask_smooth_ms is None:
            n_grad_time = 1
        else:
            n_grad_time = int(time_mask_smooth_ms / ((self._hop_length / self.sr) * 1000))
            if n_grad_time < 1:
                raise ValueError("time_mask_smooth_ms needs to be at least {}ms".format(
                    int((self._hop_length / self.sr) * 1000)))
        if (n_grad_time == 1) & (n_grad_freq == 1):
            self.smooth_mask = False
        else:
            self.smooth_mask = True
            self._n_grad_freq, self._n_grad_time = n_grad_freq, n_grad_time
            self._smoothing_filter = _smoothing_filter(n_grad_freq, n_grad_time)

    # -- device plumbing ---------------------------------------------------------------------
    def _gate_kwargs(self):
        return dict(variant=_ffi.SG_VARIANT_S, n_fft=self._n_fft, win_length=self._win_length,
                    hop_length=self._hop_length, n_grad_freq=self._n_grad_freq,
                    n_grad_time=self._n_grad_time, smooth_mask=self.smooth_mask,
                    chunk_size=self._chunk_size, padding=self.padding,
                    prop_decrease=self._prop_decrease, exact=self._exact)

    def _to_device(self, a):
        """Upload a host array (or pass a tensor) as one of the dtypes the kernels read
        natively; anything else goes through float64 like the reference's chunk copy
        (base.py:140)."""
        if isinstance(a, torch.Tensor):
            t = a.to(self.device)
            if t.dtype not in _ffi._TORCH_DTYPES:
                t = t.to(torch.float64)
            return t
        a = np.asarray(a)
        if a.dtype not in _DEVICE_DTYPES:
            a = a.astype(np.float64)
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def _device_y(self):
        if self._pipe is not None:
            self._pipeline_finish()
        if self._y_dev is None:
            self._y_dev = self._to_device(self.y)
        return self._y_dev

    def _finish(self, out_dev):
        """Device result -> what the reference returns: input dtype, 1-D if the input was
        (base.py:217-226)."""
        if self._tensor_io:
            out = out_dev if out_dev.dtype == self._dtype else out_dev.to(self._dtype)
            return out.flatten() if self.flat else out
        if out_dev.dtype in _ffi._TORCH_DTYPES and _DEVICE_DTYPES.get(np.dtype(self._dtype)) == out_dev.dtype:
            # straight DMA into the array we return (no staging copy); the array is backed by a pooled
            # page-locked buffer when one is available (_hostbuf.py)
            out, out_t = _hostbuf.result_array(tuple(out_dev.shape), self._dtype)
            out_t.copy_(out_dev)
        else:
            out = out_dev.cpu().numpy().astype(self._dtype, copy=False)
        return out.reshape(-1) if self.flat else out

    # -- the reference's host-side chunk helpers (kept for API compatibility) -----------------
    def _host_y(self):
        return self.y.cpu().numpy() if self._tensor_io else self.y

    def _read_chunk(self, i1, i2):
        """read chunk and pad with zeros (base.py:130-142)"""
        i1b = max(i1, 0)
        i2b = min(i2, self.n_frames)
        chunk = np.zeros((self.n_channels, i2 - i1))
        chunk[:, i1b - i1: i2b - i1] = self._host_y()[:, i1b:i2b]
        return chunk

    def filter_chunk(self, start_frame, end_frame):
        """Pad and perform filtering (base.py:144-150)"""
        i1 = start_frame - self.padding
        i2 = end_frame + self.padding
        padded_chunk = self._read_chunk(i1, i2)
        filtered_padded_chunk = self._do_filter(padded_chunk)
        return filtered_padded_chunk[:, start_frame - i1: end_frame - i1]

    def _get_filtered_chunk(self, ind):
        """Grabs a single chunk (base.py:152-156)"""
        start0 = ind * self._chunk_size
        end0 = (ind + 1) * self._chunk_size
        return self.filter_chunk(start_frame=start0, end_frame=end0)

    def _do_filter(self, chunk):
        """The operator seam (base.py:158-160): float64 (C, Lp) padded chunk in, filtered
        chunk of the same shape out -- computed by sg_filter_padded on the GPU."""
        if self._gate is None:
            raise NotImplementedError
        is_np = isinstance(chunk, np.ndarray)
        dev = self._to_device(chunk)
        def run():
            self._bind()
            return self._gate.filter_padded(dev, out_dtype=dev.dtype if dev.dtype.is_floating_point
                                            else torch.float64)
        if is_np:   # host array out: the call synchronises anyway -> deferred device errors are checked (and re-run)
            return self._gate.run_checked(run).cpu().numpy()
        with self._gate.lock:
            return run()

    def _bind(self):
        """Load per-object state into the shared engine handle (stationary gate: its threshold)."""


    # -- host arrays: upload / compute / download in pieces, overlapped ----------
    # The reference streams chunks through a memmap (base.py:180-216).  Here a host recording crosses PCIe twice, and done one
    # after the other -- upload everything, gate, download everything -- a ten-minute float32 recording takes 4.5 ms of which
    # 0.3 is compute.  PCIe is full duplex, so the recording goes up in chunk-aligned pieces on a side stream while, on the
    # caller's stream, the previous piece is gated (a start_frame / end_frame range of the one device copy: the chunk grid is
    # the reference's, every output bit that of the one-upload path) with the kernels STORING STRAIGHT INTO THE PAGE-LOCKED
    # RESULT ARRAY.  Measured (tools/ubench/pcie_*.py, profiles/r05_pcie_overlap.txt): a runtime device-to-host copy next to
    # an upload halves the upload's rate whatever the streams, piece size or synchronisation (the two copies serialise);
    # stores issued by a kernel do overlap it.  The first and last pieces are short: they are the two that run alone.
    _PIPE_PIECE_BYTES = 24 << 20
    _PIPE_MIN_PIECES = 3

    def _pipeline_pieces(self):
        """[(p0, p1), ...] chunk-aligned pieces of a host recording worth pipelining, else None."""
        import os
        if self._tensor_io or os.environ.get("NOISEREDUCE_AMD_PIPELINE", "1") == "0":
            return None
        if self._chunk_size is None or np.dtype(self._dtype) not in _DEVICE_DTYPES or not isinstance(self.y, np.ndarray):
            return None
        # torch.from_numpy (the per-piece upload) refuses views with a negative stride (y[::-1], np.flip) -- and there is
        # nothing to pipeline in a recording without channels or samples: those take the one-upload path, which goes
        # through np.ascontiguousarray
        # (the stride of a length-1 axis is arbitrary -- np.expand_dims leaves 0 or anything else there -- and never used)
        if self.y.ndim != 2 or self.n_channels <= 0 or self.n_frames <= 0 or \
                any(st <= 0 for st, n in zip(self.y.strides, self.y.shape) if n > 1):
            return None
        cs, N = int(self._chunk_size), int(self.n_frames)
        if cs <= 0:
            return None
        n_chunks = -(-N // cs)
        piece_bytes = int(os.environ.get("NOISEREDUCE_AMD_PIPELINE_PIECE_BYTES", self._PIPE_PIECE_BYTES))
        k = max(1, piece_bytes // (self.n_channels * cs * np.dtype(self._dtype).itemsize))
        short = max(1, k // 4)
        sizes, left = [short], n_chunks - short
        while left > k + short:
            sizes.append(k)
            left -= k
        if left > short:
            sizes.append(left - short)
            left = short
        if left > 0:
            sizes.append(left)
        if len(sizes) < self._PIPE_MIN_PIECES:
            return None
        pieces, p0 = [], 0
        for n in sizes:
            pieces.append((p0, min(p0 + n * cs, N)))
            p0 += n * cs
        return pieces

    def _pipeline_begin(self):
        """Allocate the device copy and send up the first piece (with its right halo); the stationary gate takes its noise
        statistics from it.  None when the recording is not worth pipelining."""
        pieces = self._pipeline_pieces()
        if pieces is None:
            return None
        dev, N, C = self.device, int(self.n_frames), int(self.n_channels)
        with torch.cuda.device(dev):
            cur, up = torch.cuda.current_stream(dev), _pipe_stream(dev)
            x_dev = torch.empty((C, N), dtype=_DEVICE_DTYPES[np.dtype(self._dtype)], device=dev)
            up.wait_stream(cur)     # (the allocator may have handed back memory that work queued on `cur` still reads)
        self._pipe = dict(pieces=pieces, x_dev=x_dev, sent=0)
        self._pipeline_upload(0)
        return x_dev

    def _pipeline_upload(self, i):
        """Piece i's samples and right halo (its left halo came with piece i - 1).  Blocks this thread until they are up."""
        st = self._pipe
        N, pad = int(self.n_frames), int(self.padding)
        b = min(N, st["pieces"][i][1] + pad)
        a = st["sent"]
        if b > a:
            up = _pipe_stream(self.device)
            with torch.cuda.device(self.device), torch.cuda.stream(up):
                st["x_dev"][:, a:b].copy_(torch.from_numpy(self.y[:, a:b]), non_blocking=True)
                done = torch.cuda.Event()
                done.record(up)
            done.synchronize()    # THIS upload only: the side stream is shared by every thread of the process
            st["sent"] = b

    def _get_traces_pipelined(self):
        """None if the result array could not be page-locked (then the one-upload path is the faster one)."""
        st = self._pipe
        pieces, x_dev = st["pieces"], st["x_dev"]
        out, out_t = _hostbuf.result_array((int(self.n_channels), int(self.n_frames)), self._dtype)
        if not out_t.is_pinned():
            return None
        with torch.cuda.device(self.device), self._gate.lock:
            try:    # an error left by an earlier unchecked call on this shared handle is that call's (Gate.run_checked)
                self._gate.check_errors()
            except _ffi.HandoffTimeout as e:
                raise RuntimeError("an earlier, unchecked call on this engine handle lost a tile hand-off (its output "
                                   "is invalid); this call has not run: " + str(e)) from None
            for i, (p0, p1) in enumerate(pieces):
                self._bind()
                # out: the page-locked host array itself (device-visible at the same address)
                self._gate.process_chunks(x_dev, out=out_t[:, p0:p1], start_frame=p0, end_frame=p1, chunked=True)
                if i + 1 < len(pieces):
                    self._pipeline_upload(i + 1)       # piece i is being gated and stored to the host meanwhile
            self._gate.check_errors()   # synchronises; HandoffTimeout -> the caller re-runs on the plain path
        return out.reshape(-1) if self.flat else out

    def _pipeline_finish(self):
        """Whatever happened, leave the object with a complete device copy (later calls and the plain path use it)."""
        st = self._pipe
        if st is not None:
            self._pipeline_upload(len(st["pieces"]) - 1)     # sends whatever has not gone up yet
            self._pipe = None
            self._y_dev = st["x_dev"]

    def get_traces(self, start_frame=None, end_frame=None):
        """Grab filtered data iterating over chunks (base.py:167-226) -- on the device."""
        if self._gate is None:
            raise NotImplementedError
        if self._pipe is None and self._y_dev is None and start_frame is None and end_frame is None:
            self._pipeline_begin()
        if self._pipe is not None:
            try:
                if start_frame is None and end_frame is None:
                    try:
                        out = self._get_traces_pipelined()
                        if out is not None:
                            return out
                    except _ffi.HandoffTimeout:
                        pass    # (rare: a fused kernel's bounded wait timed out) the plain path below checks and re-runs
            finally:
                self._pipeline_finish()
        if start_frame is None:
            start_frame = 0
        if end_frame is None:
            end_frame = self.n_frames
        ydev = self._device_y()
        chunked = self._chunk_size is not None and end_frame - start_frame > self._chunk_size
        def run():
            self._bind()
            return self._gate.process_chunks(ydev, out_dtype=ydev.dtype, start_frame=start_frame,
                                             end_frame=end_frame, chunked=chunked)
        if self._tensor_io:   # device tensor out: asynchronous, no host synchronisation (Gate.check_errors is the
            with self._gate.lock:   # caller's to use; an unchecked failure surfaces at the next call on the handle)
                out = run()
        else:
            out = self._gate.run_checked(run)
        return self._finish(out)
