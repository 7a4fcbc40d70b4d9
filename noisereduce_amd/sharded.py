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
  the rank that owns the noise clip computes it and broadcasts n_fft/2+1 doubles; with
  channel sharding the channel mean of the clip is an all-reduce(sum) of one clip-length
  vector.

``filter_fn`` makes the compute step pluggable so the partition/exchange logic is testable on
CPU (world_size 2, gloo) against the oracle; the product path uses the HIP engine.
"""
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


def exchange_seams(y_local, padding, group=None):
    """All-gather the seam samples and return (left_halo, right_halo), each (C, padding):
    the previous rank's last / the next rank's first `padding` samples, zeros at the ends of
    the recording (== the reference's zero padding, base.py:139-141)."""
    ws = dist.get_world_size(group)
    rank = dist.get_rank(group)
    C, S = y_local.shape
    if S < padding:
        raise ValueError("time shard shorter than the chunk padding")
    seams = torch.stack([y_local[:, :padding], y_local[:, S - padding:]]).contiguous()
    if ws == 1:
        z = torch.zeros_like(seams[0])
        return z, z.clone()
    gathered = [torch.empty_like(seams) for _ in range(ws)]
    dist.all_gather(gathered, seams, group=group)
    left = gathered[rank - 1][1] if rank > 0 else torch.zeros_like(seams[0])
    right = gathered[rank + 1][0] if rank < ws - 1 else torch.zeros_like(seams[0])
    return left, right


def with_halos(y_local, padding, group=None):
    """(C, padding + S + padding) buffer: [left halo | shard | right halo]."""
    if padding == 0:
        return y_local
    left, right = exchange_seams(y_local, padding, group)
    return torch.cat([left, y_local, right], dim=1)


class ShardedStationaryGate:
    """reduce_noise(stationary=True) of a time-sharded (C, n_total) recording.  Each rank
    constructs this with ITS shard (already on its GPU) and calls run()."""

    def __init__(self, y_local, sr, n_total=None, group=None, device=None, **kw):
        from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
        self.group = group
        self.ws = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if y_local.dim() == 1:
            y_local = y_local[None, :]
        self.y_local = y_local
        self.chunk_size = kw.get("chunk_size", 600000)
        self.padding = kw.get("padding", 30000)
        S = y_local.shape[1]
        if self.ws > 1 and S % self.chunk_size != 0 and self.rank != self.ws - 1:
            raise ValueError("time shards must be chunk-aligned")
        # The gate object of this rank: statistics from the local data on rank 0 (y_noise=None
        # means "the first chunk_size samples of the recording", stationary.py:47-64, which
        # live on rank 0); other ranks build theirs on a stand-in clip and get the threshold
        # by broadcast.
        defaults = dict(y_noise=None, n_std_thresh_stationary=1.5, chunk_size=600000,
                        clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None,
                        hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
                        time_mask_smooth_ms=50, tmp_folder=None, prop_decrease=1.0,
                        use_tqdm=False, n_jobs=1)
        defaults.update(kw)
        self.sg = SpectralGateStationary(y=y_local, sr=sr,
                                         device=device or y_local.device, **defaults)
        if self.ws > 1:
            thr = torch.from_numpy(self.sg.noise_thresh).to(y_local.device)
            dist.broadcast(thr, src=0, group=group)
            if self.rank != 0:
                self.sg._gate.set_noise_threshold(thr.cpu().numpy())

    def run(self):
        """Filter this rank's shard; returns (C, S) on the rank's device."""
        S = self.y_local.shape[1]
        pad = self.padding
        if self.ws == 1:
            return self.sg._gate.process_chunks(self.y_local, chunked=S > self.chunk_size)
        ext = with_halos(self.y_local, pad, self.group)
        # the sharded recording is always "chunked" (it is longer than one chunk)
        return self.sg._gate.process_chunks(ext, out_dtype=self.y_local.dtype, chunked=True,
                                            halo_left=pad, halo_right=pad)


def reduce_noise_time_sharded(y_local, sr, filter_fn, chunk_size=600000, padding=30000, group=None):
    """Backend-agnostic form used by the CPU tests: `filter_fn(ext, halo)` filters the chunks
    of a shard given its halo-extended buffer and returns (C, S)."""
    ext = with_halos(y_local, padding, group)
    return filter_fn(ext, padding)
