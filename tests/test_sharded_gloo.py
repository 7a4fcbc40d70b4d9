"""world_size-2 test of the time-sharded gate's partition + exchange logic on CPU (gloo).
The compute steps are supplied by the oracle here (the product backend is the HIP engine);
the result of the two ranks, concatenated, must equal the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import spectralgate_oracle as O

SR, CS, PAD, NFFT = 48000, 20000, 3000, 1024


class OracleBackend:
    """Same interface as noisereduce_amd.sharded.HipStationaryBackend, computed by the oracle."""
    chunk_size, padding = CS, PAD

    def stats(self, y_local):
        return self

    def threshold(self, y_local):
        thr, _, _ = O.noise_threshold_S(y_local.numpy().astype(np.float64), NFFT, NFFT, NFFT // 4,
                                        1.5, CS)
        return torch.from_numpy(thr)

    def filter(self, y_local, ext, halo, thresh, owner):
        filt = O.smoothing_filter(5, 9)
        C, S = y_local.shape
        e = ext.numpy().astype(np.float64)
        if halo == 0:
            e = np.pad(e, ((0, 0), (PAD, PAD)))
        out = np.zeros((C, S))
        for i in range(-(-S // CS)):
            win = np.zeros((C, CS + 2 * PAD))
            seg = e[:, i * CS:i * CS + CS + 2 * PAD]
            win[:, :seg.shape[1]] = seg
            res = O.gate_stationary_S(win, thresh.numpy(), NFFT, NFFT, NFFT // 4, 1.0, filt)
            n = min(CS, S - i * CS)
            out[:, i * CS:i * CS + n] = res[:, PAD:PAD + n]
        return torch.from_numpy(out)


def _worker_channels(rank, world, port, y, want, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import ChannelShardedStationary

    class Backend:
        chunk_size, padding = CS, PAD

        def filter_with_noise(self, y_local, noise):
            out = O.reduce_noise_S(y_local.numpy().astype(np.float64), SR, stationary=True,
                                   y_noise=noise.numpy(), chunk_size=CS, padding=PAD, n_fft=NFFT)
            return torch.from_numpy(out)

    C = y.shape[0] // world
    y_local = y[rank * C:(rank + 1) * C].contiguous()
    out = ChannelShardedStationary(Backend()).run(y_local)
    ret[rank] = float((out - want[rank * C:(rank + 1) * C]).abs().max() / want.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_channel_sharded_two_ranks_matches_single_process():
    n = 3 * CS + 777
    y = np.stack([O.synth_signal(n, seed=60 + c, tone_hz=150.0 * (c + 1)).astype(np.float64) for c in range(4)])
    want = O.reduce_noise_S(y, SR, stationary=True, chunk_size=CS, padding=PAD, n_fft=NFFT)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_channels, args=(2, _free_port(), torch.from_numpy(y), torch.from_numpy(want), ret),
             nprocs=2, join=True)
    # the all-reduced channel mean can differ from numpy's sequential mean in the last bit
    assert ret[0] < 1e-9 and ret[1] < 1e-9


def test_channel_sharded_eight_ranks_matches_single_process():
    # configs[3]'s layout on the scaling node: 8 ranks, channels dealt 2 per rank (64 channels / 8 GPUs there); the only
    # exchange is the all-reduce of the noise clip's channel sum
    n = 2 * CS + 333
    y = np.stack([O.synth_signal(n, seed=80 + c, tone_hz=120.0 * (c + 1)).astype(np.float64) for c in range(16)])
    want = O.reduce_noise_S(y, SR, stationary=True, chunk_size=CS, padding=PAD, n_fft=NFFT)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_channels, args=(8, _free_port(), torch.from_numpy(y), torch.from_numpy(want), ret),
             nprocs=8, join=True)
    assert all(ret[r] < 1e-9 for r in range(8)), dict(ret)


def _worker_channels_uneven(rank, world, port, y, want, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import ChannelShardedStationary, channel_bounds

    class Backend:
        chunk_size, padding = CS, PAD

        def filter_with_noise(self, y_local, noise):
            out = O.reduce_noise_S(y_local.numpy().astype(np.float64), SR, stationary=True,
                                   y_noise=noise.numpy(), chunk_size=CS, padding=PAD, n_fft=NFFT)
            return torch.from_numpy(out)

    c0, c1 = channel_bounds(y.shape[0], world, rank)
    y_local = y[c0:c1].contiguous()
    cs = ChannelShardedStationary(Backend())
    out = cs.run(y_local)                      # the channel count rides in the all-reduce: no c_total needed
    assert out.shape == (c1 - c0, y.shape[1])
    err = float((out - want[c0:c1]).abs().max() / want.abs().max()) if c1 > c0 else 0.0
    cs.check_channel_total(y.shape[0])
    out2 = cs.run(y_local, c_total=y.shape[0])  # an explicit (correct) total gives the same result
    same = bool(torch.equal(out, out2))
    try:
        cs.check_channel_total(y.shape[0] + 1)
        wrong = "no error"
    except ValueError:
        wrong = "ValueError"
    ret[rank] = (c1 - c0, err, same, wrong)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("channels,world", [(5, 2), (3, 2), (2, 3), (7, 3)])
def test_channel_sharded_uneven_channel_counts(channels, world):
    """C not divisible by the world size: channels are dealt in contiguous runs, the first C % world
    ranks hold one more, ranks beyond the channel count hold none and still take part in the all-reduce; the channel mean
    of the noise clip is over the TRUE total (stationary.py:61-64)."""
    from noisereduce_amd.sharded import channel_bounds
    bounds = [channel_bounds(channels, world, r) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == channels and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    assert max(b - a for a, b in bounds) - min(b - a for a, b in bounds) <= 1
    n = CS + 555
    y = np.stack([O.synth_signal(n, seed=90 + c, tone_hz=170.0 * (c + 1)).astype(np.float64) for c in range(channels)])
    want = O.reduce_noise_S(y, SR, stationary=True, chunk_size=CS, padding=PAD, n_fft=NFFT)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_channels_uneven, args=(world, _free_port(), torch.from_numpy(y), torch.from_numpy(want), ret),
             nprocs=world, join=True)
    assert sum(ret[r][0] for r in range(world)) == channels
    for r in range(world):
        assert ret[r][1] < 1e-9 and ret[r][2] and ret[r][3] == "ValueError", dict(ret)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, y, want, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import TimeShardedStationary, exchange_seams, shard_bounds
    s0, s1 = shard_bounds(y.shape[1], CS, world, rank)
    y_local = y[:, s0:s1].contiguous()
    left, right = exchange_seams(y_local, PAD)
    ok = True
    ok &= bool(torch.equal(left, y[:, s0 - PAD:s0]) if rank > 0 else torch.all(left == 0))
    ok &= bool(torch.equal(right, y[:, s1:s1 + PAD]) if rank < world - 1 else torch.all(right == 0))
    out = TimeShardedStationary(OracleBackend(), NFFT // 2 + 1).run(y_local)
    err = float((out - want[:, s0:s1]).abs().max() / want.abs().max())
    ret[rank] = (ok, err, (s0, s1))
    dist.barrier()
    dist.destroy_process_group()


def test_time_sharded_two_ranks_matches_single_process():
    n = 7 * CS + 1234          # 8 chunks, the last one partial: rank 0 gets 4, rank 1 gets 4
    y = np.stack([O.synth_signal(n, seed=5).astype(np.float64),
                  O.synth_signal(n, seed=6, tone_hz=300.0).astype(np.float64)])
    want = O.reduce_noise_S(y, SR, stationary=True, chunk_size=CS, padding=PAD, n_fft=NFFT)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), torch.from_numpy(y), torch.from_numpy(want), ret),
             nprocs=2, join=True)
    assert ret[0][2] == (0, 4 * CS) and ret[1][2] == (4 * CS, n)
    for r in (0, 1):
        ok, err, _ = ret[r]
        assert ok, "seam exchange returned wrong halos"
        assert err < 1e-12, err


def test_shard_bounds_cover_and_align():
    from noisereduce_amd.sharded import shard_bounds
    for n, cs, ws in [(28_800_000, 600_000, 8), (1_000_001, 600_000, 2), (90_000, 25_000, 3),
                      (600_000, 600_000, 4)]:
        edges = [shard_bounds(n, cs, ws, r) for r in range(ws)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
            assert a1 == b0 and (a1 % cs == 0 or a1 == n)


# ---- edge layouts: short last shard, more ranks than chunks, collective validation ----
def _worker_layout(rank, world, port, y, want, bounds, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import TimeShardedStationary
    s0, s1 = bounds[rank]
    y_local = y[:, s0:s1].contiguous()
    try:
        out = TimeShardedStationary(OracleBackend(), NFFT // 2 + 1).run(y_local)
        err = float((out - want[:, s0:s1]).abs().max() / want.abs().max()) if s1 > s0 else 0.0
        ret[rank] = ("ok", err, tuple(out.shape))
    except ValueError as e:          # must be raised by EVERY rank (after the collective), never by one
        ret[rank] = ("ValueError", str(e), None)
    dist.barrier()
    dist.destroy_process_group()


def _run_layout(n, world, bounds=None):
    from noisereduce_amd.sharded import shard_bounds
    y = np.stack([O.synth_signal(n, seed=15).astype(np.float64)])
    want = O.reduce_noise_S(y, SR, stationary=True, chunk_size=CS, padding=PAD, n_fft=NFFT)
    if bounds is None:
        bounds = [shard_bounds(n, CS, world, r) for r in range(world)]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_layout, args=(world, _free_port(), torch.from_numpy(y), torch.from_numpy(want), bounds, ret),
             nprocs=world, join=True)
    return [ret[r] for r in range(world)], bounds


def test_time_sharded_last_shard_shorter_than_padding():
    # 1 full chunk + 1000 samples (< PAD = 3000): rank 1's shard is shorter than the seam it must send
    res, bounds = _run_layout(CS + 1000, 2)
    assert bounds == [(0, CS), (CS, CS + 1000)]
    for kind, err, _ in res:
        assert kind == "ok" and err < 1e-12, (kind, err)


def test_time_sharded_more_ranks_than_chunks():
    # 2 chunks on 3 ranks: rank 2 holds nothing, takes part in the exchange and returns an empty result
    res, bounds = _run_layout(2 * CS - 5, 3)
    assert bounds[2][0] == bounds[2][1]
    assert res[2][0] == "ok" and res[2][2] == (1, 0)
    for kind, err, _ in res[:2]:
        assert kind == "ok" and err < 1e-12, (kind, err)


def test_time_sharded_eight_ranks():
    # the node the scaling runs use: 8 ranks, 21 chunks (the last one partial) -> shards of 3, 3, 3, 3, 3, 2, 2, 2 chunks;
    # every rank's result against the single-process oracle, seams exchanged in ONE all-gather
    n = 20 * CS + 777
    res, bounds = _run_layout(n, 8)
    assert bounds[0][0] == 0 and bounds[-1][1] == n
    assert all(bounds[r][1] == bounds[r + 1][0] and bounds[r][1] % CS == 0 for r in range(7))
    for kind, err, shape in res:
        assert kind == "ok" and err < 1e-12, (kind, err)


def test_time_sharded_bad_layout_raises_on_every_rank():
    # rank 0's shard is not chunk-aligned: every rank must raise (none may hang in the all-gather)
    n = 2 * CS
    res, _ = _run_layout(n, 2, bounds=[(0, CS + 17), (CS + 17, n)])
    assert [r[0] for r in res] == ["ValueError", "ValueError"], res
    assert "not chunk-aligned" in res[0][1] and res[0][1] == res[1][1]


# ---- the verdict on a shard layout must be the same on every rank on EVERY call, and a rank that fails
# before the collective must not strand the others ----
def _worker_relayout(rank, world, port, y, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import TimeShardedStationary
    be = OracleBackend()                       # ONE backend: the exchange buffers and the validation memo persist
    log = []
    # call 1: good layout; call 2: rank 0's shard loses its chunk alignment while rank 1's length stays the same
    for b in ([(0, CS), (CS, 2 * CS)], [(0, CS - 40), (CS, 2 * CS)]):
        s0, s1 = b[rank]
        try:
            TimeShardedStationary(be, NFFT // 2 + 1).run(y[:, s0:s1].contiguous())
            log.append("ok")
        except ValueError as e:
            log.append("ValueError")
    ret[rank] = log
    dist.barrier()
    dist.destroy_process_group()


def test_layout_change_on_one_rank_raises_on_every_rank():
    y = torch.from_numpy(np.stack([O.synth_signal(2 * CS, seed=16).astype(np.float64)]))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_relayout, args=(2, _free_port(), y, ret), nprocs=2, join=True)
    assert ret[0] == ["ok", "ValueError"] and ret[1] == ["ok", "ValueError"], dict(ret)


def _worker_rank0_fails(rank, world, port, y, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import TimeShardedStationary

    class Failing(OracleBackend):
        def threshold(self, y_local):
            raise RuntimeError("noise statistics failed on rank 0")
    try:
        TimeShardedStationary(Failing(), NFFT // 2 + 1).run(y[:, rank * CS:(rank + 1) * CS].contiguous())
        ret[rank] = "ok"
    except RuntimeError as e:
        ret[rank] = "RuntimeError"       # rank 0: its own error, re-raised AFTER the collective
    except ValueError as e:
        ret[rank] = "ValueError"         # the others: told through the gathered header
    dist.barrier()
    dist.destroy_process_group()


def test_rank0_statistics_failure_reaches_every_rank():
    y = torch.from_numpy(np.stack([O.synth_signal(2 * CS, seed=17).astype(np.float64)]))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_rank0_fails, args=(2, _free_port(), y, ret), nprocs=2, join=True)
    assert ret[0] == "RuntimeError" and ret[1] == "ValueError", dict(ret)


# ---- the verdict must be SYMMETRIC (every rank raises in the same call); a NaN threshold instead of a stale
# one when rank 0's statistics fail; VERDICT r3 item 6: a sharded wrapper for the non-stationary gate ----
def _worker_recover(rank, world, port, y, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import TimeShardedStationary
    be = OracleBackend()
    log = []
    # good, bad on rank 0 only, good again: both ranks raise in call 2 and BOTH are back in step for call 3 (a rank that
    # raised one call later than the other would enter call 3's all-gather alone and hang)
    for b in ([(0, CS), (CS, 2 * CS)], [(0, CS - 40), (CS, 2 * CS)], [(0, CS), (CS, 2 * CS)]):
        s0, s1 = b[rank]
        try:
            out = TimeShardedStationary(be, NFFT // 2 + 1).run(y[:, s0:s1].contiguous())
            log.append("ok %d" % out.shape[1])
        except ValueError:
            log.append("ValueError")
    ret[rank] = log
    dist.barrier()
    dist.destroy_process_group()


def test_bad_layout_then_recovery_keeps_the_ranks_in_step():
    y = torch.from_numpy(np.stack([O.synth_signal(2 * CS, seed=18).astype(np.float64)]))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_recover, args=(2, _free_port(), y, ret), nprocs=2, join=True)
    assert ret[0] == ["ok %d" % CS, "ValueError", "ok %d" % CS] and ret[1] == ret[0], dict(ret)


def _worker_nan_threshold(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import exchange_seams_and_threshold
    bufs = {}
    y = torch.ones((1, CS), dtype=torch.float64)
    thr = torch.full((5,), 7.0, dtype=torch.float64)
    _, _, t1 = exchange_seams_and_threshold(y, PAD, thr if rank == 0 else None, 5, None, bufs, chunk_size=CS)
    first = t1.clone()
    try:
        exchange_seams_and_threshold(y, PAD, None, 5, None, bufs, chunk_size=CS)   # rank 0 lost its threshold
        second = "no error"
    except ValueError:
        second = "ValueError"
    # what the receive buffer holds for the threshold now: NaN, not the 7.0 of the call before
    stale = bufs["recv"].view(world, -1)[0, -40:].view(torch.float64)
    ret[rank] = (first.tolist(), second, bool(torch.isnan(stale).all()))
    dist.barrier()
    dist.destroy_process_group()


def test_failed_statistics_never_leave_a_stale_threshold():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_nan_threshold, args=(2, _free_port(), ret), nprocs=2, join=True)
    for r in (0, 1):
        assert ret[r] == ([7.0] * 5, "ValueError", True), dict(ret)


def _oracle_nonstat_filter(ext, halo, out_dtype):
    from noisereduce_amd.spectralgate.nonstationary import iir_coefficient
    filt = O.smoothing_filter(5, 9)
    b = iir_coefficient(2.0, SR, NFFT // 4)
    C = ext.shape[0]
    S = ext.shape[1] - 2 * halo
    e = ext.numpy().astype(np.float64)
    if halo == 0:
        e = np.pad(e, ((0, 0), (PAD, PAD)))
    out = np.zeros((C, S))
    for i in range(-(-S // CS)):
        win = np.zeros((C, CS + 2 * PAD))
        seg = e[:, i * CS:i * CS + CS + 2 * PAD]
        win[:, :seg.shape[1]] = seg
        res = O.gate_nonstationary_S(win, NFFT, NFFT, NFFT // 4, 1.0, filt, b, 2, 10)
        n = min(CS, S - i * CS)
        out[:, i * CS:i * CS + n] = res[:, PAD:PAD + n]
    return torch.from_numpy(out)


def _worker_nonstat(rank, world, port, y, want, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from noisereduce_amd.sharded import TimeShardedNonStationary, shard_bounds
    s0, s1 = shard_bounds(y.shape[1], CS, world, rank)
    out = TimeShardedNonStationary(_oracle_nonstat_filter, CS, PAD).run(y[:, s0:s1].contiguous())
    ret[rank] = (float((out - want[:, s0:s1]).abs().max() / want.abs().max()) if s1 > s0 else 0.0, out.shape[1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_time_sharded_nonstationary_matches_single_process(world):
    """Non-stationary gate, time-sharded: the seam exchange alone (no threshold) -- every rank's output equals the
    single-process oracle on its slice, including the chunks whose windows reach into the neighbouring shard."""
    n = 5 * CS + 1234
    y = torch.from_numpy(np.stack([O.synth_signal(n, seed=21 + c).astype(np.float64) for c in range(2)]))
    want = torch.from_numpy(O.reduce_noise_S(y.numpy(), SR, stationary=False, chunk_size=CS, padding=PAD, n_fft=NFFT))
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_nonstat, args=(world, _free_port(), y, want, ret), nprocs=world, join=True)
    assert sum(v[1] for v in ret.values()) == n
    assert all(v[0] < 1e-12 for v in ret.values()), dict(ret)
