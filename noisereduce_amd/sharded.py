"""Multi-GPU sharding of the variant-S chunk grid: one process per GPU
(``torch.distributed``, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Every (channel, chunk) unit of the reference's chunk grid is filtered independently
(/root/reference/noisereduce/spectralgate/base.py:144-156), so the path shards with no
data-path collective.  Two small exchanges remain:

* time sharding of a long recording (rank r holds samples [r*S, (r+1)*S), S a multiple of
  chunk_size): chunk windows reach ``padding`` samples into the neighbouring shard, so the
  ranks all-gather their first/last ``padding`` samples per channel (2*pad*C samples per
  rank -- the "seam" exchange) and pass them to the kernels as halos.  Outputs need no
  exchange: the reference discards the padded part of every chunk (base.py:150).
* the stationary threshold is a property of the whole recording (stationary.py:47-81):
  the rank that owns the noise clip computes it; its n_fft/2+1 doubles ride in the SAME
  all-gather as the seams (one collective per call: small-message collectives over xGMI are
  latency-bound, ~tens of microseconds each against a ~0.6 ms step).  With channel sharding the
  channel mean of the clip is an all-reduce(sum) of one clip-length vector.

``filter_fn`` makes the compute step pluggable so the partition/exchange logic is testable on
CPU (world_size 2, gloo) against the oracle; the product path uses the HIP engine.
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_total, chunk_size, world_size, rank):
    """Chunk-aligned time shard [s0, s1) of rank `rank`: chunks are dealt out in contiguous
    runs, the first (n_chunks % world_size) ranks get one more."""
    n_chunks = -(-n_total // chunk_size)
    q, r = divmod(n_chunks, world_size)
    c0 = rank * q + min(rank, r)
    c1 = c0 + q + (1 if rank < r else 0)
    return min(c0 * chunk_size, n_total), min(c1 * chunk_size, n_total)


def channel_bounds(c_total, world_size, rank):
    """Channels [c0, c1) of rank `rank` when C channels are dealt to the ranks in contiguous runs (the first
    c_total % world_size ranks hold one more; with fewer channels than ranks the last ranks hold none -- they still take
    part in the all-reduce of ChannelShardedStationary)."""
    q, r = divmod(int(c_total), int(world_size))
    c0 = rank * q + min(rank, r)
    return c0, c0 + q + (1 if rank < r else 0)


def exchange_seams(y_local, padding, group=None):
    """All-gather the seam samples and return (left_halo, right_halo), each (C, padding):
    the previous rank's last / the next rank's first `padding` samples, zeros at the ends of
    the recording (== the reference's zero padding, base.py:139-141)."""
    ws = dist.get_world_size(group)
    rank = dist.get_rank(group)
    C, S = y_local.shape
    if S >= padding:
        seams = torch.stack([y_local[:, :padding], y_local[:, S - padding:]]).contiguous()
    else:
        # short last shard: zero-filled seams (== the zero padding beyond the recording's end); raising
        # here would strand the other ranks in the all-gather
        seams = y_local.new_zeros((2, C, padding))
        if S:
            seams[0][:, :S] = y_local
            seams[1][:, padding - S:] = y_local
    if ws == 1:
        z = torch.zeros_like(seams[0])
        return z, z.clone()
    gathered = [torch.empty_like(seams) for _ in range(ws)]
    dist.all_gather(gathered, seams, group=group)
    left = gathered[rank - 1][1] if rank > 0 else torch.zeros_like(seams[0])
    right = gathered[rank + 1][0] if rank < ws - 1 else torch.zeros_like(seams[0])
    return left, right


def with_halos(y_local, padding, group=None, ext=None):
    """(C, padding + S + padding) buffer: [left halo | shard | right halo].  If `ext` is given it
    must be that buffer with the shard already in its middle (see alloc_shard): only the two halo
    strips are written, no copy of the shard."""
    if padding == 0:
        return y_local
    left, right = exchange_seams(y_local, padding, group)
    if ext is None:
        return torch.cat([left, y_local, right], dim=1)
    S = y_local.shape[1]
    ext[:, :padding].copy_(left)
    ext[:, padding + S:].copy_(right)
    return ext


def alloc_shard(channels, shard_len, padding, dtype, device):
    """Allocate a rank's shard inside a halo-extended buffer.  Returns (ext, shard) where shard is
    the (channels, shard_len) view to fill with this rank's samples."""
    ext = torch.zeros((channels, shard_len + 2 * padding), dtype=dtype, device=device)
    return ext, ext[:, padding:padding + shard_len]


class HipStationaryBackend:
    """Compute steps of the sharded stationary gate on this rank's MI355X."""

    def __init__(self, sr, device, slot=0, **kw):
        self.sr, self.device, self.slot = sr, device, slot
        self.kw = dict(y_noise=None, n_std_thresh_stationary=1.5, chunk_size=600000,
                       clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None,
                       hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
                       time_mask_smooth_ms=50, tmp_folder=None, prop_decrease=1.0,
                       use_tqdm=False, n_jobs=1)
        # precision (reduce_noise's extension): "float64" selects the float64 pipeline on this rank's handle too -- a sharded
        # call must not quietly come back float32-accurate; None defers to NOISEREDUCE_AMD_EXACT like cached_gate
        precision = kw.pop("precision", None)
        if precision not in (None, "float32", "float64"):
            raise ValueError('precision must be None, "float32" or "float64"')
        self.exact = None if precision is None else precision == "float64"
        self.kw.update(kw)
        self.chunk_size, self.padding = self.kw["chunk_size"], self.kw["padding"]
        self._g = None

    def _gate(self, y_local=None, with_stats=False):
        """The engine handle for these settings (built once per backend, cached per device by _ffi)."""
        if self._g is None:
            from noisereduce_amd import _ffi
            from noisereduce_amd.spectralgate.base import SpectralGate
            k = self.kw
            W = k["n_fft"] if k["win_length"] is None else k["win_length"]
            H = W // 4 if k["hop_length"] is None else k["hop_length"]
            probe = SpectralGate.__new__(SpectralGate)  # only to reuse the filter-design arithmetic
            probe.sr, probe._n_fft, probe._hop_length = self.sr, k["n_fft"], H
            probe._n_grad_freq = probe._n_grad_time = 1
            probe.smooth_mask = False
            if not (k["freq_mask_smooth_hz"] is None and k["time_mask_smooth_ms"] is None):
                probe._generate_mask_smoothing_filter(k["freq_mask_smooth_hz"], k["time_mask_smooth_ms"])
            self._g = _ffi.cached_gate(self.device, slot=self.slot, variant=_ffi.SG_VARIANT_S, stationary=True,
                                       n_fft=k["n_fft"], win_length=W, hop_length=H,
                                       n_grad_freq=probe._n_grad_freq, n_grad_time=probe._n_grad_time,
                                       smooth_mask=probe.smooth_mask, chunk_size=k["chunk_size"],
                                       padding=k["padding"], prop_decrease=k["prop_decrease"],
                                       n_std_thresh=k["n_std_thresh_stationary"], top_db=80.0, ddof=0,
                                       exact=self.exact)
        return self._g

    def lock(self):
        """The engine handle's lock (callers hold it across stats -> filter)."""
        return self._gate().lock

    def stats(self, y_local):
        """Noise statistics from this rank's data, left on the device (owning rank only):
        y_noise=None means the recording itself, clipped to chunk_size (stationary.py:47-64)."""
        g = self._gate()
        noise = y_local
        if self.kw["y_noise"] is not None:
            noise = self.kw["y_noise"]
            if noise.dim() == 1:
                noise = noise[None, :]
        if self.kw["clip_noise_stationary"] and self.chunk_size is not None:
            noise = noise[:, :self.chunk_size]
        g.noise_stats(noise)
        g.thresh_owner = None   # the handle no longer holds any SpectralGateStationary object's threshold
        return g

    def threshold(self, y_local):
        """Per-band threshold (dB) as a device tensor, for the broadcast (no host sync)."""
        return self.stats(y_local).noise_threshold_tensor()

    def filter(self, y_local, ext, halo, thresh, owner):
        """Filter the shard.  `ext` is the halo-extended buffer (or y_local when halo == 0)."""
        g = self._gate()
        if not owner:
            g.set_noise_threshold_tensor(thresh)
            g.thresh_owner = None
        S = y_local.shape[1]
        if halo == 0 and ext is y_local:
            return g.process_chunks(y_local, chunked=S > self.chunk_size)
        return g.process_chunks(ext, out_dtype=y_local.dtype, chunked=True, halo_left=halo,
                                halo_right=halo)


def _check_shard_lengths(lens, chunk_size, padding):
    """Validate the time shards of ALL ranks (identical verdict on every rank: called with the gathered
    lengths, so every rank raises -- or none does).  Non-empty shards must be contiguous from rank 0;
    every shard but the last non-empty one must be a positive multiple of chunk_size and at least
    `padding` long (its seams are real samples); the last one may be short (its missing seam samples lie
    beyond the end of the recording == the reference's zero padding, base.py:139-141)."""
    if any(n < 0 for n in lens):    # a rank could not produce its contribution (rank 0: the noise statistics)
        bad = [i for i, n in enumerate(lens) if n < 0]
        raise ValueError(f"rank(s) {bad} failed before the seam exchange (noise statistics of the first chunk); "
                         f"shard lengths {[n if n >= 0 else -n - 1 for n in lens]}")
    nonempty = [i for i, n in enumerate(lens) if n > 0]
    if not nonempty:
        return
    last = nonempty[-1]
    for r in range(last):
        n = lens[r]
        if n <= 0:
            raise ValueError(f"time shard of rank {r} is empty but rank {last} holds samples "
                             f"(shard lengths {list(lens)})")
        if chunk_size and n % chunk_size != 0:
            raise ValueError(f"time shard of rank {r} ({n} samples) is not chunk-aligned "
                             f"(chunk_size {chunk_size}; shard lengths {list(lens)})")
        if n < padding:
            raise ValueError(f"time shard of rank {r} ({n} samples) is shorter than the chunk padding "
                             f"({padding})")


def flush_pending(bufs):
    """Deferred mode: raise the verdict on the LAST call's gathered shard lengths (every rank alike).  Call it after the
    last `exchange_seams_and_threshold(..., defer=True)` of a loop; a no-op when nothing is pending."""
    if bufs is not None and bufs.get("pending") is not None:
        host, ev, pcs, ppad = bufs.pop("pending")
        cause = bufs.pop("pending_cause", None)   # rank 0: the exception its statistics raised in the deferred call
        ev.synchronize()
        try:
            _check_shard_lengths(host.tolist(), pcs, ppad)
        except ValueError as e:
            if cause is not None:
                raise e from cause
            raise


def exchange_seams_and_threshold(y_local, padding, thr, n_bins, group=None, bufs=None, chunk_size=None, defer=False):
    """ONE all-gather carrying every rank's shard length, seam samples and rank 0's threshold.
    Each rank contributes [int64 shard length | first `padding` | last `padding` samples of every channel
    | n_bins float64] as raw bytes (only rank 0's threshold slot is meaningful).  Returns
    (left_halo, right_halo, thr): halos (C, padding) in y_local's dtype, zeros at the ends of the
    recording; thr float64 (n_bins,).
    No rank raises BEFORE the collective (the others would block in it until the RCCL timeout): a shard
    shorter than `padding` sends zero-filled seams, `thr=None` on rank 0 (its statistics failed) travels as a
    negative length AND a NaN threshold (a NaN threshold gates everything: no rank can filter with a stale one), and
    the shard layout is validated from the GATHERED lengths.  The verdict is SYMMETRIC: every rank takes the same
    branch, which depends on `defer` only (never on what a single rank knows about its own shard):
      * defer=False (default): every rank reads the gathered lengths back (one small device-to-host copy) and
        validates them in THIS call -- every rank raises in the same call, or none does;
      * defer=True (hot loops; needs `bufs`): the lengths go to page-locked memory asynchronously and are validated by
        every rank at the START of its next call, before that call's collective, or by `flush_pending(bufs)` after the
        last one -- every rank raises one call late, together.  The call with the bad layout has already produced its
        (wrong) output by then: callers that use `defer` treat the result of call k as valid once call k + 1 or the
        flush has returned.
    `bufs`: optional dict that keeps the send/receive buffers between calls."""
    ws, rank = dist.get_world_size(group), dist.get_rank(group)
    C, S = y_local.shape
    es = y_local.element_size()
    HDR = 8
    seam_bytes = 2 * C * padding * es
    sb = HDR + (seam_bytes + 7) // 8 * 8                # threshold slot 8-byte aligned
    total = sb + n_bins * 8
    key = (total, ws, y_local.device)
    flush_pending(bufs)     # verdict on the previous deferred call (arrived long ago: no stall); every rank alike
    if bufs is not None and bufs.get("key") == key:
        send, recv = bufs["send"], bufs["recv"]
    else:
        send = torch.zeros(total, dtype=torch.uint8, device=y_local.device)
        recv = torch.empty(ws * total, dtype=torch.uint8, device=y_local.device)
        if bufs is not None:
            bufs.update(key=key, send=send, recv=recv, len_dev=None)
    S_hdr = S if (rank != 0 or thr is not None) else -S - 1      # rank 0 without a threshold: failure marker
    if bufs is None or bufs.get("len_dev") != S_hdr:
        send[:HDR].view(torch.int64).fill_(S_hdr)
        if bufs is not None:
            bufs["len_dev"] = S_hdr
    if padding:
        seams = send[HDR:HDR + seam_bytes].view(y_local.dtype).view(2, C, padding)
        if S >= padding:
            seams[0].copy_(y_local[:, :padding])
            seams[1].copy_(y_local[:, S - padding:])
        else:   # short (last) shard: what lies beyond it is the zero padding of the recording's end
            seams.zero_()
            if S:
                seams[0][:, :S].copy_(y_local)
                seams[1][:, padding - S:].copy_(y_local)
    if rank == 0:
        if thr is not None:
            send[sb:].view(torch.float64).copy_(thr)
        else:
            send[sb:].view(torch.float64).fill_(float("nan"))    # never a stale threshold from an earlier call
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(ws, total)
    hdr = recv[:, :HDR].contiguous().view(torch.int64).flatten()
    if defer and bufs is not None and hdr.is_cuda:
        host = bufs.get("hdr_host")
        if host is None or host.numel() != ws:
            host = bufs["hdr_host"] = torch.empty(ws, dtype=torch.int64).pin_memory()
        host.copy_(hdr, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        bufs["pending"] = (host, ev, chunk_size, padding)
    else:
        _check_shard_lengths(hdr.cpu().tolist(), chunk_size, padding)   # every rank, this call
    thr_out = recv[0, sb:].view(torch.float64)

    def seam_of(r, which):
        return recv[r, HDR:HDR + seam_bytes].view(y_local.dtype).view(2, C, padding)[which]

    zero = torch.zeros((C, padding), dtype=y_local.dtype, device=y_local.device)
    left = seam_of(rank - 1, 1) if (rank > 0 and padding) else zero
    right = seam_of(rank + 1, 0) if (rank < ws - 1 and padding) else zero
    return left, right, thr_out


class TimeShardedStationary:
    """reduce_noise(stationary=True, y_noise=None) of a recording that is time-sharded over the
    ranks of `group` (rank r holds the r-th chunk-aligned slice, already on its device).
    `backend` supplies the two compute steps (HipStationaryBackend in production)."""

    def __init__(self, backend, n_bins, group=None):
        self.backend = backend
        self.n_bins = n_bins
        self.group = group
        self.ws = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # send/receive buffers of the exchange live with the backend (one per process), not with this
        # per-call object
        self._bufs = backend.__dict__.setdefault("_xchg_bufs", {})

    def finish(self):
        """After a loop of run(..., defer_check=True): the verdict on the last call's shard layout (every rank)."""
        flush_pending(self._bufs)

    def run(self, y_local, ext=None, defer_check=False, timing=None):
        """y_local: this rank's (C, S) shard.  ext: optional halo-extended buffer that already
        holds the shard in its middle (alloc_shard) -- avoids copying the shard every call.
        defer_check: validate the gathered shard layout one call late (no host synchronisation in this call; see
        exchange_seams_and_threshold) -- for hot loops that end with finish().
        timing: optional list; a (start, end) pair of CUDA events around the exchange is appended per call (measurement
        of the time a rank's stream spends in / waiting for the collective: rank 0 enters it after its statistics, the
        other ranks at once -- their time in it is the exposed wait for rank 0)."""
        if y_local.dim() == 1:
            y_local = y_local[None, :]
        pad, cs = self.backend.padding, self.backend.chunk_size
        S = y_local.shape[1]
        # (shard layout -- chunk alignment, lengths vs padding -- is validated collectively inside the
        # exchange, from the gathered lengths: a rank-local raise here would strand the other ranks in
        # the all-gather)
        # threshold: y_noise=None means "the first chunk_size samples of the recording"
        # (stationary.py:47-64); they live on rank 0, which broadcasts n_bins doubles.
        # the statistics -> filter sequence depends on handle state: one call at a time per handle
        lock = getattr(self.backend, "lock", None)
        with (lock() if lock is not None else contextlib.nullcontext()):
            if self.ws == 1:
                self.backend.stats(y_local)
                return self.backend.filter(y_local, y_local, 0, None, owner=True)
            thr, thr_err = None, None
            if self.rank == 0:
                try:        # never raise before the collective: a failure travels to every rank in the header
                    thr = self.backend.threshold(y_local)
                except Exception as e:      # noqa: BLE001 -- re-raised below, after the exchange
                    thr_err = e
            ev = None
            if timing is not None and y_local.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            try:
                left, right, thr = exchange_seams_and_threshold(y_local, pad, thr, self.n_bins, self.group,
                                                                self._bufs, chunk_size=cs, defer=defer_check)
            except ValueError:
                if thr_err is not None:
                    raise thr_err
                raise
            if thr_err is not None and defer_check:
                # deferred verdict: this call returns output gated with the NaN threshold (everything gated) and every
                # rank raises at its next call / finish(); rank 0 keeps the original exception to chain it there
                self._bufs["pending_cause"] = thr_err
            if ev is not None:
                ev[1].record()
                timing.append(ev)
            if S == 0:      # more ranks than chunks: this rank took part in the exchange and has nothing to filter
                return y_local.new_empty((y_local.shape[0], 0))
            if pad == 0:
                ext = y_local
            elif ext is None:
                ext = torch.cat([left, y_local, right], dim=1)
            else:
                ext[:, :pad].copy_(left)
                ext[:, pad + S:].copy_(right)
            # (rank 0 after a failed statistics pass does not "own" a threshold: it loads the NaN one it sent, like the others)
            return self.backend.filter(y_local, ext, pad, thr, owner=self.rank == 0 and thr_err is None)


class TimeShardedNonStationary:
    """reduce_noise(stationary=False) of a recording that is time-sharded over the ranks of `group`.  The gate has no
    state shared between chunks (nonstationary.py:47-97: floor, sigmoid mask and smoothing are per chunk window), so
    the ONLY exchange is the seam all-gather that lets chunk windows reach `padding` samples into the neighbouring
    shards (base.py:144-156) -- no threshold, nothing else.  `filter_fn(ext, halo, out_dtype) -> (C, S)` is the
    compute step: the HIP engine's `process_chunks` in production (hip_nonstationary_filter), the oracle in the CPU
    tests.  The shard layout is validated like the stationary gate's (same header, same verdict on every rank)."""

    def __init__(self, filter_fn, chunk_size, padding, group=None):
        self.filter_fn = filter_fn
        self.chunk_size, self.padding = chunk_size, padding
        self.group = group
        self.ws = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._bufs = {}

    def finish(self):
        flush_pending(self._bufs)

    def run(self, y_local, ext=None, defer_check=False):
        if y_local.dim() == 1:
            y_local = y_local[None, :]
        pad, S = self.padding, y_local.shape[1]
        if self.ws == 1:
            return self.filter_fn(y_local, 0, y_local.dtype)
        # the threshold slot of the shared exchange stays empty (0 bins); rank 0 "has" its (non-existent) threshold
        left, right, _ = exchange_seams_and_threshold(y_local, pad, torch.empty(0, dtype=torch.float64, device=y_local.device),
                                                      0, self.group, self._bufs, chunk_size=self.chunk_size,
                                                      defer=defer_check)
        if S == 0:
            return y_local.new_empty((y_local.shape[0], 0))
        if pad == 0:
            return self.filter_fn(y_local, 0, y_local.dtype)
        if ext is None:
            ext = torch.cat([left, y_local, right], dim=1)
        else:
            ext[:, :pad].copy_(left)
            ext[:, pad + S:].copy_(right)
        return self.filter_fn(ext, pad, y_local.dtype)


def hip_nonstationary_filter(gate, chunk_size):
    """filter_fn of TimeShardedNonStationary on the HIP engine: `gate` is the _ffi.Gate of a SpectralGateNonStationary
    (its chunk grid reads the halos as real neighbour samples)."""
    def fn(ext, halo, out_dtype):
        if halo == 0:
            return gate.process_chunks(ext, chunked=chunk_size is not None and ext.shape[1] > chunk_size)
        return gate.process_chunks(ext, out_dtype=out_dtype, chunked=True, halo_left=halo, halo_right=halo)
    return fn


class ChannelShardedStationary:
    """reduce_noise(stationary=True, y_noise=None) of a (C_total, N) recording whose CHANNELS are
    dealt to the ranks (BASELINE.json configs[3]: 64 channels, 8 per GPU).  Every rank holds the full
    timeline of its channels, so chunk windows need no halo; the only exchange is the channel mean
    of the noise clip (stationary.py:61-64): an all-reduce(sum) of one clip-length float64 vector,
    after which every rank computes the identical threshold locally."""

    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.ws = dist.get_world_size(group) if dist.is_initialized() else 1
        self._last_count = None   # all-reduced channel count of the last run() (device tensor), for check_channel_total

    def run(self, y_local, c_total=None, timing=None):
        """y_local: this rank's (C_local, N) channels -- C_local may differ between ranks and may be 0 (more ranks than
        channels): the channel COUNT rides in the same all-reduce as the clip sum (one extra element), so the mean is
        over the true total whatever the split.  `c_total`, when given, must equal that total (checked on every rank
        alike, after the collective).  `timing`: list that receives a (start, end) CUDA-event pair around the
        all-reduce (bench.py: time a rank's stream spends in the exchange)."""
        if y_local.dim() == 1:
            y_local = y_local[None, :]
        C_local, N = y_local.shape
        kw = getattr(self.backend, "kw", {})
        y_noise = kw.get("y_noise")
        if y_noise is not None:
            # an explicit noise clip is the same on every rank (stationary.py:47-58): no exchange at all
            if C_local == 0:
                return y_local.new_empty((0, N))
            noise = y_noise if y_noise.dim() == 2 else y_noise[None, :]
            return self.backend.filter_with_noise(y_local, noise, clip=kw.get("clip_noise_stationary", True))
        # y_noise=None: the recording itself, clipped to chunk_size unless clip_noise_stationary=False
        # or chunk_size=None (stationary.py:61-64)
        cs = self.backend.chunk_size
        n_clip = min(N, cs) if (kw.get("clip_noise_stationary", True) and cs is not None) else N
        buf = torch.empty(n_clip + 1, dtype=torch.float64, device=y_local.device)
        buf[:n_clip] = y_local[:, :n_clip].to(torch.float64).sum(dim=0)
        buf[n_clip] = float(C_local)
        if self.ws > 1:
            ev = None
            if timing is not None and y_local.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            if ev is not None:
                ev[1].record()
                timing.append(ev)
        if c_total is None:
            clip_mean = (buf[:n_clip] / buf[n_clip]).unsqueeze(0)     # (1, n_clip): its own "channel mean"; no host sync
        else:
            # the caller's count is used for the mean (identical on every rank by contract); the all-reduced count is
            # NOT read back here (a device-to-host sync per call) -- check_channel_total() does that on demand
            clip_mean = (buf[:n_clip] / float(c_total)).unsqueeze(0)
        self._last_count = buf[n_clip:]
        if C_local == 0:
            return y_local.new_empty((0, N))
        return self.backend.filter_with_noise(y_local, clip_mean)

    def check_channel_total(self, c_total):
        """After run(): the all-reduced channel count of the last call equals `c_total` (synchronises; every rank reads
        the same number, so every rank raises -- or none does)."""
        if self._last_count is None:
            raise RuntimeError("check_channel_total: no run() with y_noise=None has taken place on this object yet")
        got = int(round(float(self._last_count.item())))
        if got != int(c_total):
            raise ValueError(f"channel-sharded gate: the ranks hold {got} channels in total, caller said {c_total}")


def _hip_filter_with_noise(self, y_local, noise, clip=False):
    """HipStationaryBackend: statistics from an explicit noise clip, then the chunk grid.
    clip=False: `noise` already IS the clip the statistics are taken from (the all-reduced channel mean)."""
    from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
    kw = dict(self.kw)
    kw["y_noise"] = noise
    kw["clip_noise_stationary"] = clip
    sg = SpectralGateStationary(y=y_local, sr=self.sr, device=self.device, slot=self.slot, **kw)
    return sg.get_traces()


HipStationaryBackend.filter_with_noise = _hip_filter_with_noise
