"""Two ranks, HIP engine: the time- and channel-sharded gates against the single-process oracle.
The ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device) -- the exchange
code and the engine calls are exactly those of the N-GPU bench (halo-extended shards, one
all-gather carrying seams + threshold)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import spectralgate_oracle as O

pytestmark = pytest.mark.gpu
SR, CS, PAD, NFFT = 48000, 50000, 6000, 1024
TOL = 1e-4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    return torch.device("cuda", 0)


def _worker_time(rank, world, port, y, want, ret):
    dev = _init(rank, world, port)
    from noisereduce_amd.sharded import (HipStationaryBackend, TimeShardedStationary, alloc_shard,
                                         shard_bounds)
    s0, s1 = shard_bounds(y.shape[1], CS, world, rank)
    backend = HipStationaryBackend(SR, dev, chunk_size=CS, padding=PAD, n_fft=NFFT)
    errs = []
    for dtype in (torch.float64, torch.float32):
        ext, shard = alloc_shard(y.shape[0], s1 - s0, PAD, dtype, dev)
        shard.copy_(y[:, s0:s1].to(dtype))
        gate = TimeShardedStationary(backend, NFFT // 2 + 1)
        for use_ext in (True, False):          # halos written into the extended buffer / concatenated
            out = gate.run(shard, ext=ext if use_ext else None)
            assert out.shape == shard.shape and out.dtype == dtype
            errs.append(float((out.double().cpu() - want[:, s0:s1]).abs().max() / want.abs().max()))
    ret[rank] = max(errs)
    dist.barrier()
    dist.destroy_process_group()


def test_time_sharded_hip_two_ranks():
    n = 5 * CS + 4321                          # 6 chunks, the last one partial: 3 + 3
    y = np.stack([O.synth_signal(n, seed=41).astype(np.float64),
                  O.synth_signal(n, seed=42, tone_hz=2500.0).astype(np.float64)])
    want = O.reduce_noise_S(y, SR, stationary=True, chunk_size=CS, padding=PAD, n_fft=NFFT)
    ret = mp.Manager().dict()
    mp.spawn(_worker_time, args=(2, _free_port(), torch.from_numpy(y), torch.from_numpy(want), ret),
             nprocs=2, join=True)
    assert ret[0] < TOL and ret[1] < TOL, dict(ret)


def _worker_channels(rank, world, port, y, want, ret):
    dev = _init(rank, world, port)
    from noisereduce_amd.sharded import ChannelShardedStationary, HipStationaryBackend
    C = y.shape[0] // world
    y_local = y[rank * C:(rank + 1) * C].to(torch.float32).to(dev)
    backend = HipStationaryBackend(SR, dev, chunk_size=CS, padding=PAD, n_fft=NFFT)
    out = ChannelShardedStationary(backend).run(y_local)
    ret[rank] = float((out.double().cpu() - want[rank * C:(rank + 1) * C]).abs().max() / want.abs().max())
    dist.barrier()
    dist.destroy_process_group()


def test_channel_sharded_hip_two_ranks():
    n = 2 * CS + 999
    y = np.stack([O.synth_signal(n, seed=70 + c, tone_hz=180.0 * (c + 1)).astype(np.float64) for c in range(4)])
    want = O.reduce_noise_S(y, SR, stationary=True, chunk_size=CS, padding=PAD, n_fft=NFFT)
    ret = mp.Manager().dict()
    mp.spawn(_worker_channels, args=(2, _free_port(), torch.from_numpy(y), torch.from_numpy(want), ret),
             nprocs=2, join=True)
    assert ret[0] < TOL and ret[1] < TOL, dict(ret)


def test_rccl_single_rank_collectives():
    """The collectives of the sharded gate on the real "nccl" (= RCCL) backend: a world_size-1 group in a
    subprocess (one GPU is all a test box has) -- uint8 all_gather_into_tensor with seams + threshold,
    barrier, float64 all_reduce."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "nccl_single_rank.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "collectives ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


def _worker_deferred_failure(rank, world, port, y, ret):
    dev = _init(rank, world, port)
    from noisereduce_amd.sharded import HipStationaryBackend, TimeShardedStationary, alloc_shard, shard_bounds

    class Failing(HipStationaryBackend):
        def threshold(self, y_local):
            raise RuntimeError("noise statistics failed on rank 0")
    s0, s1 = shard_bounds(y.shape[1], CS, world, rank)
    ext, shard = alloc_shard(1, s1 - s0, PAD, torch.float32, dev)
    shard.copy_(y[:, s0:s1].float())
    gate = TimeShardedStationary(Failing(SR, dev, chunk_size=CS, padding=PAD, n_fft=NFFT), NFFT // 2 + 1)
    out = gate.run(shard, ext=ext, defer_check=True)      # deferred verdict: no rank raises in the failing call
    gated = bool((out == 0).all() or torch.isnan(out).all())   # filtered with the NaN threshold: nothing passes
    try:
        gate.finish()
        res = ("no error", None)
    except ValueError as e:
        res = ("ValueError", type(e.__cause__).__name__ if e.__cause__ is not None else None)
    ret[rank] = (gated,) + res
    dist.barrier()
    dist.destroy_process_group()


def test_deferred_statistics_failure_keeps_its_cause():
    """with defer_check=True a failure of rank 0's noise statistics surfaces one call late on EVERY rank (a
    ValueError from the gathered header) -- and on rank 0 the original exception is chained to it instead of being lost."""
    y = torch.from_numpy(np.stack([O.synth_signal(4 * CS, seed=19).astype(np.float64)]))
    ret = mp.Manager().dict()
    mp.spawn(_worker_deferred_failure, args=(2, _free_port(), y, ret), nprocs=2, join=True)
    assert ret[0][1:] == ("ValueError", "RuntimeError"), dict(ret)
    assert ret[1][1:] == ("ValueError", None), dict(ret)
    assert ret[0][0] and ret[1][0], dict(ret)


def _worker_channels_uneven(rank, world, port, y, want, ret):
    dev = _init(rank, world, port)
    from noisereduce_amd.sharded import ChannelShardedStationary, HipStationaryBackend, channel_bounds
    c0, c1 = channel_bounds(y.shape[0], world, rank)
    y_local = y[c0:c1].to(torch.float32).to(dev)
    cs = ChannelShardedStationary(HipStationaryBackend(SR, dev, chunk_size=CS, padding=PAD, n_fft=NFFT))
    ev = []
    out = cs.run(y_local, timing=ev)
    cs.check_channel_total(y.shape[0])
    ret[rank] = (float((out.double().cpu() - want[c0:c1]).abs().max() / want.abs().max()), len(ev))
    dist.barrier()
    dist.destroy_process_group()


def test_channel_sharded_hip_uneven_channels():
    """3 channels on 2 ranks (2 + 1): the channel mean of the noise clip is over all 3 (the count rides in the all-reduce)."""
    n = 2 * CS + 999
    y = np.stack([O.synth_signal(n, seed=75 + c, tone_hz=210.0 * (c + 1)).astype(np.float64) for c in range(3)])
    want = O.reduce_noise_S(y, SR, stationary=True, chunk_size=CS, padding=PAD, n_fft=NFFT)
    ret = mp.Manager().dict()
    mp.spawn(_worker_channels_uneven, args=(2, _free_port(), torch.from_numpy(y), torch.from_numpy(want), ret),
             nprocs=2, join=True)
    assert ret[0][0] < TOL and ret[1][0] < TOL and ret[0][1] == 1 and ret[1][1] == 1, dict(ret)


@pytest.mark.parametrize("workload", ["config2", "config3", "config4"])
def test_bench_two_ranks_end_to_end_over_gloo(workload):
    """`python bench.py --gpus 2` exactly as the driver starts an N-GPU run (bare: it launches its own ranks through
    torch.distributed.run on 127.0.0.1), with BENCH_BACKEND=gloo so that both ranks may share this box's one GPU: launcher,
    rendezvous, sharded step, max-over-ranks timing, the weak-scaling line with its distributed parity verdict -- everything
    but RCCL itself, so that the first 8-GPU lease cannot fail on plumbing."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    steps = "2"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", steps, "--warmup", "1",
           "--no-cpu-baseline", "--workload", workload]
    pr = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "Msamples/s"
    assert d["value"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None
    per_gpu = d["config"]["samples_per_gpu"]
    # whole-job aggregate: the samples of BOTH ranks over the slowest rank's time
    assert abs(d["value"] - 2 * per_gpu / (d["ms_per_step"] * 1e-3) / 1e6) <= 0.01 * d["value"]
    dd = d["distributed"]
    assert dd["world_size"] == 2 and dd["parity_ok"] is True and len(dd["per_rank"]) == 2
    assert len(dd["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in dd["ms_per_step_per_rank"])
    if workload == "config4":
        sg = dd["single_gpu_same_share"]
        assert sg["ms_per_step"] > 0 and sg["throughput_factor_vs_it"] > 0
        assert all(g["rel_err_unit"] < 1e-4 for g in dd["per_rank"])
    else:
        assert all(g["halo_ok"] and g["rel_err_chunk0"] < 1e-4 for g in dd["per_rank"])
        assert "roofline" in d and d["roofline"]["frac"] > 0
