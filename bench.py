#!/usr/bin/env python3
"""bench.py -- Msamples/s of reduce_noise on MI355X (BASELINE.json metric).

Workload (configs[1]): synthetic 48 kHz mono, 10 min (28.8 M samples) of white noise + 1 kHz
tone, float32, stationary reduce_noise, n_fft=1024, hop=256, chunk_size=600000,
padding=30000.  One "step" = one whole reduce_noise pass over the recording, input and
output resident in HBM (noise statistics + the full chunk grid: every kernel of the path).

N GPUs (weak scaling): the recording is N x 10 min, time-sharded on chunk boundaries, one
process per GPU; per step ONE all-gather carries every rank's seam samples (2*padding per rank)
and rank 0's per-band threshold -- the only collective of the path.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the step
(HIP events around every launch, on the launch stream, inside the timed region);
`cpu_baseline` times the numpy oracle (a port of the reference's CPU path) on a bounded
sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 48000
SECONDS = 600
N_PER_GPU = SR * SECONDS            # 28.8 M samples
CHUNK, PAD, NFFT, HOP = 600000, 30000, 1024, 256
ALGO_BYTES_PER_SAMPLE = 8           # 4 B float32 read + 4 B float32 written
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec


def synth_on_device(n, seed, device, tone_hz=1000.0, offset=0):
    """0.1*N(0,1) + 0.5*sin(2 pi f t) as float32, generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    noise = torch.randn(n, generator=g, device=device, dtype=torch.float32) * 0.1
    t = (torch.arange(n, device=device, dtype=torch.float64) + offset) / SR
    return (noise + 0.5 * torch.sin(2 * np.pi * tone_hz * t).float()).contiguous()


def cpu_baseline():
    """numpy oracle ("port") on the host: stationary reduce_noise of the first 60 s of the
    workload (2.88 M samples = 5 chunks), 1 warm-up + median of 3."""
    from oracle import spectralgate_oracle as O
    n = SR * 60
    y = O.synth_signal(n, dtype=np.float32).astype(np.float64)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        O.reduce_noise_S(y, SR, stationary=True, n_fft=NFFT, chunk_size=CHUNK, padding=PAD)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[1:]))
    return {"value": round(n / med / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "first 60 s (2.88 M samples, 5 chunks) of the workload, stationary, "
                      "oracle/spectralgate_oracle.py reduce_noise_S, float64, numpy single thread, "
                      "median of 3 after 1 warm-up; os.cpu_count()=%d" % (os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--nonstationary", action="store_true", help="configs[2] instead of configs[1]")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent calls in flight on separate HIP streams (serving mode; default 1)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # BENCH_BACKEND=gloo lets the multi-rank code path be exercised on a box with fewer GPUs than
    # ranks (ranks share devices; RCCL itself refuses that) -- a functional check, not a measurement.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from noisereduce_amd.sharded import HipStationaryBackend, TimeShardedStationary, alloc_shard, with_halos
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary

    # this rank's time shard of the (world x 10 min) recording
    # (allocated inside a halo-extended buffer so that the seam exchange writes 2*padding samples
    # per step instead of re-copying the shard)
    y_ext, y2d = alloc_shard(1, N_PER_GPU, PAD, torch.float32, device)
    y2d[0].copy_(synth_on_device(N_PER_GPU, 1234 + rank, device, offset=rank * N_PER_GPU))
    y = y2d[0]
    stationary = not args.nonstationary

    backend = HipStationaryBackend(SR, device, chunk_size=CHUNK, padding=PAD, n_fft=NFFT)
    # --streams S > 1 (serving mode, not the default): S independent calls in flight, each on its own
    # HIP stream with its own engine handle -- the latency-bound statistics chain and the kernel tails
    # of one call hide under the kernels of another.  The default times one call at a time.
    n_streams = max(1, args.streams) if stationary else 1
    backends = [backend] + [HipStationaryBackend(SR, device, slot=i, chunk_size=CHUNK, padding=PAD, n_fft=NFFT)
                            for i in range(1, n_streams)]
    streams = [torch.cuda.current_stream(device)] + [torch.cuda.Stream(device) for _ in range(1, n_streams)]
    step_no = [0]

    def make_gate():
        if stationary:
            return TimeShardedStationary(backend, NFFT // 2 + 1)
        return SpectralGateNonStationary(
            y=y, sr=SR, chunk_size=CHUNK, padding=PAD, n_fft=NFFT, win_length=None, hop_length=None,
            time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
            thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None,
            prop_decrease=1.0, use_tqdm=False, n_jobs=1, device=device)

    def gate_of(sg):
        return sg.backend._gate() if stationary else sg._gate

    def step():
        # one whole reduce_noise: statistics + (seam, threshold) all-gather + chunk grid.
        # The engine handle (tables + workspace) is cached across calls by noisereduce_amd._ffi.
        if stationary and n_streams > 1:
            i = step_no[0] % n_streams
            step_no[0] += 1
            with torch.cuda.stream(streams[i]):
                return TimeShardedStationary(backends[i], NFFT // 2 + 1).run(y2d, ext=y_ext if world > 1 else None)
        sg = make_gate()
        if stationary:
            out = sg.run(y2d, ext=y_ext if world > 1 else None)
        else:
            gate = sg._gate
            ext = with_halos(y2d, PAD, ext=y_ext) if world > 1 else None
            if ext is None:
                out = gate.process_chunks(y2d, chunked=True)
            else:
                out = gate.process_chunks(ext, out_dtype=y.dtype, chunked=True, halo_left=PAD,
                                          halo_right=PAD)
        return out

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    gate = gate_of(make_gate())
    # untimed survey pass: every kernel bracketed by HIP events -> per-kernel table + dominant kernel
    gate.profile_read(reset=True)
    gate.profile_select(None)
    gate.profile_enable(True)
    survey_steps = 3
    for _ in range(survey_steps):
        step()
    survey = gate.profile_read(reset=True)
    dom = max(survey, key=lambda k: survey[k][0])
    # timed region: HIP events (on the launch stream) only around the dominant kernel
    gate.profile_select([dom])
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    profs = [gate.profile_read(reset=True)]
    gate.profile_enable(False)
    gate.profile_select(None)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        total = float(N_PER_GPU) * world * args.steps
        value = total / elapsed / 1e6
        # dominant kernel over the timed region
        agg = {}
        for p in profs:
            for k, (ms, cnt) in p.items():
                a = agg.setdefault(k, [0.0, 0])
                a[0] += ms
                a[1] += cnt
        avg_ms = agg[dom][0] / agg[dom][1]
        # the event pairs live in the handle of stream 0: it ran every n_streams-th step
        steps_profiled = (args.steps + n_streams - 1) // n_streams
        launches_per_step = agg[dom][1] / steps_profiled
        algo_bytes = ALGO_BYTES_PER_SAMPLE * N_PER_GPU / launches_per_step
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(dom)
            except Exception:
                traffic = None
        line = {
            "metric": "Msamples/s reduce_noise (48 kHz mono, n_fft=1024)",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("configs[1]" if stationary else "configs[2]") +
                       ": synthetic 48 kHz mono 10 min per GPU, %s reduce_noise, n_fft=1024 hop=256, "
                       "chunk_size=600000 padding=30000, float32 in/out resident in HBM"
                       % ("stationary" if stationary else "non-stationary"),
                       "samples_per_gpu": N_PER_GPU, "sharding": "time (chunk-aligned), seam all-gather",
                       "calls_in_flight": n_streams},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(algo_bytes),
                         "whole_step_frac": round(value * 1e6 / world * ALGO_BYTES_PER_SAMPLE / 1e9
                                                  / HBM_PEAK_GBS, 5)},
            "kernel_ms_per_step": {k: round(v[0] / survey_steps, 4) for k, v in
                                   sorted(survey.items(), key=lambda kv: -kv[1][0])},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
