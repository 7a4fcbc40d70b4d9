#!/usr/bin/env python3
"""bench.py -- Msamples/s of reduce_noise on MI355X (BASELINE.json metric).

Default workload (configs[1]): synthetic 48 kHz mono, 10 min (28.8 M samples) of white noise + 1 kHz
tone, float32, stationary reduce_noise, n_fft=1024, hop=256, chunk_size=600000, padding=30000.  One
"step" = one whole reduce_noise pass over the recording, input and output resident in HBM (noise
statistics + the full chunk grid: every kernel of the path).

N GPUs (weak scaling), one process per GPU:
  --workload config2 (default): the recording is N x 10 min, time-sharded on chunk boundaries; per step
      ONE all-gather (RCCL) carries every rank's seam samples (2*padding per rank) and rank 0's per-band
      threshold -- the only collective of the path.
  --workload config4: BASELINE.json configs[3] -- 8 channels x 30 min per GPU (64 channels on 8 GPUs),
      channel-sharded; per step ONE all-reduce of the noise clip's channel sum (stationary.py:61-64).
  --workload config3: configs[2] (non-stationary), time-sharded like config2 (seam all-gather only).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the step (HIP events around
every launch, on the launch stream, inside the timed region); `cpu_baseline` times the numpy oracle (a
port of the reference's CPU path) on the host cores: 1 core and a process pool over chunks (the
reference's n_jobs semantics, base.py:206-216).  Extra keys (outside the timed region, N = 1 only):
`other_configs` (configs[2], configs[4] forward / forward+backward, the PCIe-inclusive numpy->numpy
rate), `parity` (the engine's result of this run against the oracle on one chunk), and at N > 1
`distributed` (world size, backend, per-step collective time, per-rank oracle spot check, halo check).
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

# CPython garbage-collector pauses on the enqueueing thread: a generation-2 collection of a process that has torch
# imported takes ~35 ms (tools/stall_probe.py), during which the GPU idles -- the "40 ms repetition" of BENCH_r02's
# configs[2] leg.  The collections are logged, and the objects that exist once the workload is set up are frozen
# (gc.freeze) so that a full collection inside a timed leg only has the leg's own garbage to look at.
import gc as _gc
GC_PAUSES = []


def _gc_cb(phase, info, _t=[0.0]):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        GC_PAUSES.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3))


_gc.callbacks.append(_gc_cb)

SR = 48000
SECONDS = 600
N_PER_GPU = SR * SECONDS            # 28.8 M samples
CHUNK, PAD, NFFT, HOP = 600000, 30000, 1024, 256
ALGO_BYTES_PER_SAMPLE = 8           # 4 B float32 read + 4 B float32 written
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3            # MI355X_MICROARCH.md: float32 vector peak (2.4 GHz)
ALGO_FLOPS_PER_SAMPLE = 450         # two 1024-point real transforms per 256-sample hop, smoothing, log / compare / window / overlap-add
TRAFFIC_DETAIL = "profiles/r06_v7_traffic_detail.json"   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/gpu_traffic.sh), committed
C4_CHANNELS, C4_SAMPLES = 8, SR * 1800   # configs[3]: one GPU's share


class GpuState:
    """sclk / mclk / socket power / power cap / temperature of one GPU, read IN-PROCESS through librocm_smi64 (ctypes: a read
    is tens of microseconds of host time and no queue traffic, so it can sit directly before and after a timed region;
    `rocm-smi` as a subprocess would take ~0.5 s and idle the GPU).  Every field is None where the library, sysfs or the
    sensor is missing -- the reason is kept in `error`.  Purpose: a bench line that can say WHY two boxes
    differ (round 4: 0.533 ms here, 0.653 ms on the driver's box for the same call)."""

    def __init__(self, index=0):
        import ctypes
        self.ct = ctypes
        self.idx = index
        self.lib = None
        self.error = None
        try:
            lib = ctypes.CDLL("/opt/rocm/lib/librocm_smi64.so")
            rc = lib.rsmi_init(ctypes.c_uint64(0))
            if rc != 0:
                raise OSError("rsmi_init -> %d" % rc)
            n = ctypes.c_uint32(0)
            if lib.rsmi_num_monitor_devices(ctypes.byref(n)) != 0 or n.value <= index:
                raise OSError("rsmi sees %d device(s)" % n.value)

            class Freqs(ctypes.Structure):
                _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32),
                            ("current", ctypes.c_uint32), ("frequency", ctypes.c_uint64 * 33)]
            self.Freqs = Freqs
            self.lib = lib
        except Exception as e:  # noqa: BLE001 -- measurement metadata only
            self.error = repr(e)

    def _clk(self, kind):
        f = self.Freqs()
        if self.lib.rsmi_dev_gpu_clk_freq_get(self.ct.c_uint32(self.idx), self.ct.c_int(kind), self.ct.byref(f)) != 0:
            return None, None
        if f.num_supported == 0 or f.current >= 33:
            return None, None
        top = max(f.frequency[i] for i in range(min(f.num_supported, 33)))
        return round(f.frequency[f.current] / 1e6), round(top / 1e6)

    def read(self):
        """{"sclk_mhz", "sclk_max_mhz", "mclk_mhz", "power_w", "power_cap_w", "temp_c", "busy_pct"} (None where unreadable)."""
        if self.lib is None:
            return {"error": self.error}
        ct, lib, i = self.ct, self.lib, self.ct.c_uint32(self.idx)
        out = {}
        try:
            out["sclk_mhz"], out["sclk_max_mhz"] = self._clk(0)      # RSMI_CLK_TYPE_SYS
            out["mclk_mhz"], _ = self._clk(4)                        # RSMI_CLK_TYPE_MEM
            pw, ty = ct.c_uint64(0), ct.c_int(0)
            out["power_w"] = round(pw.value / 1e6, 1) if lib.rsmi_dev_power_get(i, ct.byref(pw), ct.byref(ty)) == 0 else None
            cap = ct.c_uint64(0)
            out["power_cap_w"] = round(cap.value / 1e6) if lib.rsmi_dev_power_cap_get(i, ct.c_uint32(0), ct.byref(cap)) == 0 else None
            tp = ct.c_int64(0)
            out["temp_c"] = round(tp.value / 1e3, 1) if lib.rsmi_dev_temp_metric_get(i, ct.c_uint32(1), ct.c_int(0), ct.byref(tp)) == 0 else None
            bz = ct.c_uint32(0)
            out["busy_pct"] = int(bz.value) if lib.rsmi_dev_busy_percent_get(i, ct.byref(bz)) == 0 else None
        except Exception as e:  # noqa: BLE001
            out["error"] = repr(e)
        return out


GPU_STATE = None


def gpu_state():
    """State of this process's GPU now (see GpuState); {} before main() has picked the device."""
    return GPU_STATE.read() if GPU_STATE is not None else {}


def event_pair_overhead_ms(n=200):
    """Elapsed time between the two events of an EMPTY pair on the current stream (median of n): what an event-bracketed
    kernel time carries on top of the kernel.  Per-kernel tables subtract it (`*_corrected`), so that the kernels of a step
    do not add up to more than the step (0.346 ms of kernels in a 0.319 ms step)."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def per_stage_ms(gate, fn, reps=4, ev_oh=0.0):
    """{stage: (ms per call of fn, launches per call)} with ONE stage bracketed at a time: a first pass with events around
    every launch only lists the stages of a call; then, per stage, `reps` calls with sg_profile_select([stage]) -- a single
    event pair in the queue per launch of that stage, as in the timed region.  (Events around EVERY launch make each
    bracket absorb the release / cache write-back the previous event forces: round 4's table added up to 0.346 ms for a
    0.319 ms step.)  The empty-pair elapsed time `ev_oh` is subtracted per launch."""
    gate.profile_read(reset=True)
    gate.profile_select(None)
    gate.profile_enable(True)
    fn()
    names = list(gate.profile_read(reset=True))
    out = {}
    for nm in names:
        gate.profile_select([nm])
        gate.profile_enable(True)
        for _ in range(reps):
            fn()
        pr = gate.profile_read(reset=True)
        if nm in pr and pr[nm][1]:
            ms, cnt = pr[nm]
            out[nm] = (max(0.0, ms - ev_oh * cnt) / reps, cnt / reps, ms / reps)
    gate.profile_enable(False)
    gate.profile_select(None)
    return out


def synth_on_device(n, seed, device, tone_hz=1000.0, offset=0):
    """0.1*N(0,1) + 0.5*sin(2 pi f t) as float32, generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    noise = torch.randn(n, generator=g, device=device, dtype=torch.float32) * 0.1
    t = (torch.arange(n, device=device, dtype=torch.float64) + offset) / SR
    return (noise + 0.5 * torch.sin(2 * np.pi * tone_hz * t).float()).contiguous()


# ---------------------------------------------------------------------------------------------
# CPU baseline (oracle = port of the reference's numpy/scipy path)
# ---------------------------------------------------------------------------------------------
def _pool_chunk(args):
    """One (chunk) unit of the reference's chunk grid on one host core (base.py:144-156)."""
    seed_n, ich, thr, stationary = args
    from oracle import spectralgate_oracle as O
    y = _pool_chunk.cache.get(seed_n)
    if y is None:
        y = O.synth_signal(seed_n, dtype=np.float32).astype(np.float64)
        _pool_chunk.cache[seed_n] = y
    chunk = O.read_chunk(y[None, :], ich * CHUNK - PAD, (ich + 1) * CHUNK + PAD)
    filt = O.smoothing_filter(5, 9)
    if stationary:
        out = O.gate_stationary_S(chunk, thr, NFFT, NFFT, HOP, 1.0, filt)
    else:
        out = O.gate_nonstationary_S(chunk, NFFT, NFFT, HOP, 1.0, filt, O.iir_coefficient(2.0, SR, HOP), 2, 10)
    return float(out[0, PAD]), ich


_pool_chunk.cache = {}


def cpu_baseline_reference(budget_s=40.0, timeout_s=240.0):
    """The UNMODIFIED reference timed on this host's cores (oracle/time_reference.py in a subprocess: no HIP context,
    none of this process's state), when oracle/make_ref.sh has staged it under the git-ignored oracle/_ref/ (it
    travels with the gpurun snapshot like the built .so).  Returns None when it is not staged or fails."""
    import subprocess
    script = os.path.join(ROOT, "oracle", "time_reference.py")
    if not os.path.isdir(os.path.join(ROOT, "oracle", "_ref", "noisereduce")):
        return None
    env = {k: v for k, v in os.environ.items() if k not in ("OMP_NUM_THREADS", "MKL_NUM_THREADS")}
    try:
        pr = subprocess.run([sys.executable, script, "--budget", str(budget_s)], capture_output=True, text=True,
                            timeout=timeout_s, env=env, cwd=ROOT)
        lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
        res = json.loads(lines[-1])
        if "value" not in res:
            return None
        return res
    except Exception as e:
        print("bench.py: reference cpu_baseline failed (%r): falling back to the port" % (e,), file=sys.stderr)
        return None


def cpu_baseline(budget_s=25.0):
    """`cpu_baseline` of the JSON line.  Preferred: "kind": "reference" -- the reference's own numpy/scipy path
    (reduce_noise(use_torch=False), n_jobs=1 and n_jobs=os.cpu_count(), plus its TorchGate on CPU tensors), staged by
    oracle/make_ref.sh, 1 warm-up + median of 5 (oracle/time_reference.py).  The numpy oracle ("port") is timed on the
    same single-core sample beside it.  Without the staged reference: the port alone ("kind": "port")."""
    ref = cpu_baseline_reference()
    if ref is not None:
        try:
            from oracle import spectralgate_oracle as O
            n = SR * 60
            y = O.synth_signal(n, dtype=np.float32).astype(np.float64)
            ts = []
            for i in range(3):
                t0 = time.perf_counter()
                O.reduce_noise_S(y, SR, stationary=True, n_fft=NFFT, chunk_size=CHUNK, padding=PAD)
                ts.append(time.perf_counter() - t0)
            ref["port_same_sample"] = {"value": round(n / float(np.median(ts[1:])) / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                                       "kind": "port", "sample": "oracle/spectralgate_oracle.py reduce_noise_S on the same "
                                       "60 s, median of 2 after 1 warm-up (the checker's own speed, for the record)"}
        except Exception as e:
            ref["port_same_sample"] = {"error": repr(e)}
        return ref
    return cpu_baseline_port(budget_s)


def cpu_baseline_port(budget_s=25.0):
    """(a) 1 core: stationary reduce_noise of the first 60 s of the workload (5 chunks), 1 warm-up +
    median of 3.  (b) all cores: the 48 chunks of the full 10-min workload dealt to a process pool
    (one chunk per task = the reference's joblib n_jobs semantics, base.py:206-216), median of 3 after
    one warm-up pass (which also generates each worker's copy of the recording)."""
    import multiprocessing as mp
    from oracle import spectralgate_oracle as O
    n = SR * 60
    y = O.synth_signal(n, dtype=np.float32).astype(np.float64)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        O.reduce_noise_S(y, SR, stationary=True, n_fft=NFFT, chunk_size=CHUNK, padding=PAD)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times[1:]))
    res = {"value": round(n / med / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "first 60 s (2.88 M samples, 5 chunks) of the workload, stationary, "
                     "oracle/spectralgate_oracle.py reduce_noise_S, float64, numpy single thread, "
                     "median of 3 after 1 warm-up; os.cpu_count()=%d" % (os.cpu_count() or 0)}
    # configs[2] on one core: the same 60 s, non-stationary (filtfilt one-pole floor + sigmoid mask)
    ts = []
    for i in range(3):
        t0 = time.perf_counter()
        O.reduce_noise_S(y, SR, stationary=False, n_fft=NFFT, chunk_size=CHUNK, padding=PAD)
        ts.append(time.perf_counter() - t0)
    res["nonstationary"] = {"value": round(n / float(np.median(ts[1:])) / 1e6, 3), "unit": "Msamples/s", "cores": 1,
                            "kind": "port", "sample": "configs[2]: the same 60 s, stationary=False, median of 2 after 1 warm-up"}
    # configs[4] on the host cores: TorchGate restated with the torch primitives the reference calls, on CPU tensors
    try:
        from oracle.torchgate_torch_port import torchgate_cpu
        torch.manual_seed(0)
        tt = torch.arange(16000, dtype=torch.float64) / 16000
        xc = (0.1 * torch.randn(256, 16000) + 0.5 * torch.sin(2 * np.pi * 440 * tt).float()).float()
        ts = []
        for i in range(4):
            t0 = time.perf_counter()
            torchgate_cpu(xc, 16000)
            ts.append(time.perf_counter() - t0)
        res["torchgate_cpu"] = {"value": round(xc.numel() / float(np.median(ts[1:])) / 1e6, 3), "unit": "Msamples/s",
                                "cores": torch.get_num_threads(), "kind": "port",
                                "sample": "configs[4]: 256 x 16000 float32 forward, oracle/torchgate_torch_port.py (torch.stft / "
                                          "conv2d / istft on CPU tensors, torch's own thread pool), median of 3 after 1 warm-up"}
    except Exception as e:
        res["torchgate_cpu"] = {"error": repr(e)}
    # how the port's speed relates to the LIVE reference (which cannot travel to the GPU box): measured side by side in
    # the build container by tools/ref_vs_port.py, committed
    try:
        rv = json.load(open(os.path.join(ROOT, "profiles", "r03_ref_vs_port.json")))
        res["port_vs_reference"] = {
            "stationary_port_over_reference_speed": rv["stationary"]["port_over_reference_speed"],
            "nonstationary_port_over_reference_speed": rv["nonstationary"]["port_over_reference_speed"],
            "torchgate_port_over_reference_speed": rv["torchgate_256x16000_f32"]["torch_port_over_reference_speed"],
            "source": "profiles/r03_ref_vs_port.json: live reference and port timed side by side on %d build-container cores "
                      "(tools/ref_vs_port.py); > 1 means the port is FASTER than the reference it stands in for -- NOT measured "
                      "in this run" % rv["host"]["cpu_count"]}
    except Exception as e:
        res["port_vs_reference"] = {"error": repr(e)}
    # multi-core leg
    try:
        n_chunks = N_PER_GPU // CHUNK
        workers = max(1, min(n_chunks, (os.cpu_count() or 1)))
        yfull = O.synth_signal(N_PER_GPU, dtype=np.float32).astype(np.float64)
        thr, _, _ = O.noise_threshold_S(yfull[None, :CHUNK], NFFT, NFFT, HOP, 1.5, CHUNK)
        del yfull
        ctx = mp.get_context("spawn")     # never fork a process that holds a HIP context
        tasks = [(N_PER_GPU, i, thr, True) for i in range(n_chunks)]
        t_start = time.perf_counter()
        with ctx.Pool(workers) as pool:
            ts = []
            for rep in range(4):
                t0 = time.perf_counter()
                pool.map(_pool_chunk, tasks, chunksize=1)
                ts.append(time.perf_counter() - t0)
                if time.perf_counter() - t_start > budget_s and rep >= 1:
                    break
        medp = float(np.median(ts[1:])) if len(ts) > 1 else ts[0]
        res["multicore"] = {"value": round(N_PER_GPU / medp / 1e6, 2), "unit": "Msamples/s", "cores": workers,
                            "kind": "port", "sample": "all %d chunks of the 10-min workload, one chunk per task in a "
                            "%d-process pool (n_jobs semantics of base.py:206-216), threshold precomputed, "
                            "median of %d passes after 1 warm-up" % (n_chunks, workers, max(1, len(ts) - 1))}
    except Exception as e:  # the single-core number stands on its own
        res["multicore"] = {"error": repr(e)}
    return res


def launcher_command(n, argv=None, port=None):
    """The command `python bench.py --gpus N` re-executes itself as when it was started without a launcher."""
    import socket
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
           list(sys.argv[1:] if argv is None else argv)


def self_launch(n):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    if torch.cuda.is_available() and torch.cuda.device_count() < n and "BENCH_BACKEND" not in env:
        print(f"bench.py: --gpus {n} but {torch.cuda.device_count()} device(s) visible: RCCL refuses ranks that share a "
              "device; set BENCH_BACKEND=gloo for a functional run", file=sys.stderr)
    rc = subprocess.call(launcher_command(n), env=env)
    if rc:
        raise SystemExit(rc)


def dry_nccl(world, rank, device, backend):
    """`bench.py --gpus N --dry-nccl`: the collectives of the sharded gates (uint8 all_gather_into_tensor of seams + threshold,
    float64 all_reduce of the clip sum, barrier) on N ranks, each on its own device -- seconds, no workload.  Rank 0 prints
    {"dry_nccl": "ok", ...}; any failure is a non-zero exit with the rank and the reason on stderr."""
    t0 = time.perf_counter()
    info = {"rank": rank, "device": str(device), "name": torch.cuda.get_device_name(device)}
    try:
        if world > 1:
            send = torch.full((1024,), rank, dtype=torch.uint8, device=device)
            recv = torch.empty(world * 1024, dtype=torch.uint8, device=device)
            dist.all_gather_into_tensor(recv, send)
            want = torch.arange(world, device=device, dtype=torch.uint8).repeat_interleave(1024)
            assert torch.equal(recv, want), "all_gather_into_tensor returned other ranks' data in the wrong order"
            v = torch.full((CHUNK + 1,), float(rank + 1), dtype=torch.float64, device=device)
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            assert float(v[0].item()) == world * (world + 1) / 2, "all_reduce(sum) wrong"
            dist.barrier()
        torch.cuda.synchronize(device)
    except Exception as e:  # noqa: BLE001
        print("bench.py --dry-nccl: rank %d on %s FAILED: %r" % (rank, device, e), file=sys.stderr)
        raise SystemExit(3)
    gathered = [info]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, info)
    if rank == 0:
        devs = sorted(g["device"] for g in gathered)
        ok = len(set(devs)) == world or backend != "nccl"
        print(json.dumps({"dry_nccl": "ok" if ok else "ranks share a device", "n_gpus": world, "backend": backend,
                          "ranks": gathered, "seconds": round(time.perf_counter() - t0, 2), "gpu_state": gpu_state()}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ---------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-every", type=int, default=4,
                    help="HIP events around the dominant kernel (and a step-boundary event) on every E-th step of the timed "
                         "region: each event pair is ~6 us of queue time around a 0.25 ms kernel, 4 %% of a step when every "
                         "step carries them (1 = every step)")
    ap.add_argument("--no-extras", action="store_true", help="skip other_configs / parity legs")
    ap.add_argument("--workload", choices=["config2", "config3", "config4"], default="config2")
    ap.add_argument("--nonstationary", action="store_true", help="alias of --workload config3")
    ap.add_argument("--dry-nccl", action="store_true",
                    help="self-check of the N-rank launch only: every rank must see its own device, then one all_reduce + one "
                         "all_gather_into_tensor + barrier over RCCL; prints one JSON line and exits (no workload)")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent calls in flight on separate HIP streams (serving mode; default 1)")
    args = ap.parse_args()
    if args.nonstationary:
        args.workload = "config3"

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started bare (`python bench.py --gpus N`): become the launcher -- one rank per GPU through
        # torch.distributed.run on a free local port; rank 0 prints the one JSON line on our stdout
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` bare (it "
                         "launches its own ranks) or through torch.distributed.run with --nproc-per-node N")
    # BENCH_BACKEND=gloo lets the multi-rank code path be exercised on a box with fewer GPUs than
    # ranks (ranks share devices; RCCL itself refuses that) -- a functional check, not a measurement.
    pg_backend = os.environ.get("BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if pg_backend == "nccl" and world > 1 and ndev < world:
        # fail fast and in words: RCCL would otherwise die inside init / the first collective with "invalid device ordinal"
        # or hang until its timeout
        raise SystemExit(f"bench.py: --gpus {world} over RCCL needs {world} visible devices, this node shows {ndev} "
                         "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?).  BENCH_BACKEND=gloo runs the same code path with "
                         "ranks sharing devices (functional check, not a measurement).")
    dev_index = local_rank if pg_backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if pg_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(pg_backend)

    global GPU_STATE
    GPU_STATE = GpuState(dev_index)
    if args.dry_nccl:
        return dry_nccl(world, rank, device, pg_backend)
    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from noisereduce_amd.sharded import (ChannelShardedStationary, HipStationaryBackend, TimeShardedNonStationary,
                                         TimeShardedStationary, alloc_shard, hip_nonstationary_filter, with_halos)
    from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary, iir_coefficient
    from oracle import spectralgate_oracle as O   # checker only, never inside the timed region

    wl = args.workload
    stationary = wl != "config3"
    backend = HipStationaryBackend(SR, device, chunk_size=CHUNK, padding=PAD, n_fft=NFFT)
    n_streams = max(1, args.streams) if wl == "config2" else 1

    if wl == "config4":
        # this rank's 8 channels of the (8 * world)-channel recording, generated on the device
        y2d = torch.empty((C4_CHANNELS, C4_SAMPLES), dtype=torch.float32, device=device)
        for c in range(C4_CHANNELS):
            gc = rank * C4_CHANNELS + c
            y2d[c] = synth_on_device(C4_SAMPLES, 1234 + gc, device, tone_hz=200.0 * (gc + 1))
        y_ext = None
        samples_per_gpu = C4_CHANNELS * C4_SAMPLES
        csg = ChannelShardedStationary(backend)
    else:
        # this rank's time shard of the (world x 10 min) recording (allocated inside a halo-extended buffer
        # so that the seam exchange writes 2*padding samples per step instead of re-copying the shard)
        y_ext, y2d = alloc_shard(1, N_PER_GPU, PAD, torch.float32, device)
        y2d[0].copy_(synth_on_device(N_PER_GPU, 1234 + rank, device, offset=rank * N_PER_GPU))
        samples_per_gpu = N_PER_GPU
    y = y2d[0]
    torch.cuda.synchronize(device)

    # --streams S > 1 (serving mode, not the default): S independent calls in flight, each on its own
    # HIP stream with its own engine handle and (N > 1) its own halo-extended buffer.
    backends = [backend] + [HipStationaryBackend(SR, device, slot=i, chunk_size=CHUNK, padding=PAD, n_fft=NFFT)
                            for i in range(1, n_streams)]
    streams = [torch.cuda.current_stream(device)] + [torch.cuda.Stream(device) for _ in range(1, n_streams)]
    exts = [y_ext]
    for s in streams[1:]:
        s.wait_stream(torch.cuda.current_stream(device))      # the shard was synthesised on the default stream
        if world > 1:
            e, sh = alloc_shard(1, N_PER_GPU, PAD, torch.float32, device)
            sh.copy_(y2d)
            exts.append(e)
        else:
            exts.append(None)
    torch.cuda.synchronize(device)
    step_no = [0]

    nonstat_gate = None
    if wl == "config3":
        nonstat_gate = SpectralGateNonStationary(
            y=y, sr=SR, chunk_size=CHUNK, padding=PAD, n_fft=NFFT, win_length=None, hop_length=None,
            time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
            thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None,
            prop_decrease=1.0, use_tqdm=False, n_jobs=1, device=device)._gate

    tsn = TimeShardedNonStationary(hip_nonstationary_filter(nonstat_gate, CHUNK), CHUNK, PAD) if wl == "config3" else None
    ts_objs = [TimeShardedStationary(b, NFFT // 2 + 1) for b in backends] if wl == "config2" else []

    def engine_gate():
        return nonstat_gate if wl == "config3" else backend._gate()

    if n_streams > 1 and wl == "config2":
        # serving mode: the one-tile-per-workgroup form of the gate (SG_OPT_TILE_ORDER 2).  Persistent workgroups (the default, 1 %
        # faster for one call at a time) hold every workgroup slot of the GPU until their launch ends, so the kernels of the other
        # calls in flight cannot slip in between: measured 97.2 / 100.8 Gsamples/s at S = 2 / 3 against 111.5 / 113.5 with this form
        from noisereduce_amd import _ffi as _ffi_mod
        for b in backends:
            b._gate().set_option(_ffi_mod.SG_OPT_TILE_ORDER, 2)

    xchg_events = []

    def step(timing=None):
        # one whole reduce_noise: statistics + exchange + chunk grid.  The engine handle (tables +
        # workspace) is cached across calls by noisereduce_amd._ffi.
        if wl == "config4":
            return csg.run(y2d, c_total=C4_CHANNELS * world, timing=timing)
        if wl == "config2":
            i = step_no[0] % n_streams
            step_no[0] += 1
            # (defer_check: the gathered shard layout of step k is validated at the start of step k + 1 -- by every rank --
            # and after the last step by finish(): no host synchronisation inside a step)
            if n_streams > 1:
                with torch.cuda.stream(streams[i]):
                    return ts_objs[i].run(y2d if i == 0 or world == 1 else exts[i][:, PAD:PAD + N_PER_GPU],
                                          ext=exts[i] if world > 1 else None, defer_check=True)
            return ts_objs[0].run(y2d, ext=y_ext if world > 1 else None, defer_check=True)
        # configs[2]: the non-stationary gate, time-sharded: the seam all-gather is its only exchange
        return tsn.run(y2d, ext=y_ext if world > 1 else None, defer_check=True)

    def sync():
        if world > 1:
            dist.barrier()
        # busy-wait for the queue to drain, THEN torch.cuda.synchronize (which returns at once): a thread that slept
        # inside the blocking synchronize enqueues its next launches 2-5 x slower for a moment, and right after the
        # opening barrier the GPU queue is empty, so that host time is part of the first timed step
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(device)
    _gc.collect()
    _gc.freeze()
    GC_PAUSES.clear()
    gate = engine_gate()
    ev_oh = event_pair_overhead_ms()
    # untimed survey passes: one stage bracketed by HIP events at a time -> per-kernel table + dominant kernel
    # (serving mode: steps rotate over the S handles; one survey call = S steps, of which exactly one runs on the profiled handle)
    survey_fn = step if n_streams == 1 else (lambda: [step() for _ in range(n_streams)])
    survey = per_stage_ms(gate, survey_fn, reps=4, ev_oh=ev_oh)
    dom = max(survey, key=lambda k: survey[k][0])
    # untimed settle loop on top of the W warm-up steps, DIRECTLY before the timed region: ~0.1 s of back-to-back steps
    # so that clock / power-state transitions of a GPU that was idle a moment ago happen BEFORE the timed steps.  (Up to
    # round 4 the survey pass and a gc.collect() sat between this loop and the timed region: the queue ran dry for
    # milliseconds and the first timed steps ran 10 % slower than the last -- `ms_per_step_all` fell monotonically.
    # One default run in ~20 of round 2 showed a single 8 ms step right after the box had been idle.)
    # (a FIXED count: every rank must run the same number of collectives)
    # Round 5: at N = 1 the loop is ADAPTIVE -- blocks of steps between HIP events until two consecutive blocks agree within
    # 1 % (at most ~1 s): a box whose clocks take longer to settle than a fixed count (round 4's driver run: first timed
    # block 0.339 ms against 0.31 after it) keeps settling, and the line records how long it took.  N > 1 keeps the fixed count.
    settle_blocks = []
    state_before_settle = gpu_state()
    if world == 1:
        blk = 5 if wl == "config4" else 200      # ~60 ms of steps per block: a slow clock ramp shows between blocks
        settle_steps = 0
        t_set = time.perf_counter()
        prev = None
        while True:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(blk):
                step()
            e1.record()
            e1.synchronize()
            cur = e0.elapsed_time(e1) / blk
            settle_steps += blk
            settle_blocks.append(round(cur, 4))
            if prev is not None and abs(cur - prev) <= 0.005 * prev:
                break
            if time.perf_counter() - t_set > 1.2:
                break
            prev = cur
        # (e1.synchronize() left the queue empty for a moment: refill it before the opening barrier of the timed region)
        for _ in range(50):
            step()
        settle_steps += 50
    else:
        settle_steps = 20 if wl == "config4" else 200
        for _ in range(settle_steps):
            step()
    # timed region: HIP events (on the launch stream) only around the dominant kernel, plus one event per
    # step boundary (median of the per-step times next to the mean)
    gate.profile_select([dom])
    gate.profile_enable(False)
    # (events on every E-th step only: the kernel's average duration over those launches is the roofline's; the steps
    # between run as a caller's would, without instrumentation in the queue)
    E = max(1, args.event_every) * n_streams
    sampled = [i for i in range(args.steps) if i % E == 0]
    mark_at = sampled + [args.steps]
    marks = {i: torch.cuda.Event(enable_timing=True) for i in mark_at} if n_streams == 1 else None
    if marks:
        # torch creates the HIP event at the first record(): do that here, not inside the timed region (right after the
        # opening barrier the GPU queue is empty, so host time of the first step is exposed: one run showed 0.43 ms)
        for m in marks.values():
            m.record()
    step_no[0] = 0   # (serving mode: the sampled steps i % E == 0 must land on the profiled handle, stream 0)
    sync()
    state_before = gpu_state()
    t0 = time.perf_counter()
    host_ms = []
    for i in range(args.steps):
        on = i % E == 0
        if on and marks:
            marks[i].record()
        gate.profile_enable(on)          # a host flag of the handle: no queue traffic
        th0 = time.perf_counter()
        out = step()
        host_ms.append((time.perf_counter() - th0) * 1e3)
    if marks:
        marks[args.steps].record()
    sync()
    elapsed = time.perf_counter() - t0
    state_after = gpu_state()
    elapsed_local = elapsed
    for o in ts_objs + ([tsn] if tsn is not None else []):
        o.finish()          # the deferred verdict on the last step's shard layout (outside the timed region)
    profs = [gate.profile_read(reset=True)]
    gate.profile_enable(False)
    gate.profile_select(None)
    # per-step times between consecutive step-boundary events (E steps apart: their mean)
    per_step = None
    if marks:
        per_step = []
        for a, b in zip(mark_at, mark_at[1:]):
            per_step += [marks[a].elapsed_time(marks[b]) / (b - a)] * (b - a)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- outside the timed region: distributed checks (every rank), then rank-0 extras -------------
    distributed = None
    if world > 1:
        info = {"ms_per_step": round(elapsed_local / args.steps * 1e3, 4), "gpu_state_after": state_after}
        if wl != "config4":
            # (1) the halo this rank received equals the neighbour's samples (regenerated here)
            ok = True
            if rank > 0:
                left = synth_on_device(N_PER_GPU, 1234 + rank - 1, device, offset=(rank - 1) * N_PER_GPU)[-PAD:]
                ok &= bool(torch.equal(y_ext[0, :PAD], left))
            if rank < world - 1:
                right = synth_on_device(N_PER_GPU, 1234 + rank + 1, device, offset=(rank + 1) * N_PER_GPU)[:PAD]
                ok &= bool(torch.equal(y_ext[0, PAD + N_PER_GPU:], right))
            info["halo_ok"] = ok
            # (2) oracle on this rank's FIRST chunk (its window reaches into the left neighbour's shard)
            eh = y_ext[0, :CHUNK + 2 * PAD].cpu().numpy().astype(np.float64)[None, :]
            filt = O.smoothing_filter(5, 9)
            if stationary:
                thr = gate.get_noise_threshold()
                ref = O.gate_stationary_S(eh, thr, NFFT, NFFT, HOP, 1.0, filt)[0, PAD:PAD + CHUNK]
            else:
                ref = O.gate_nonstationary_S(eh, NFFT, NFFT, HOP, 1.0, filt, iir_coefficient(2.0, SR, HOP), 2, 10)[0, PAD:PAD + CHUNK]
            info["rel_err_chunk0"] = O.rel_err(out[0, :CHUNK].cpu().numpy(), ref)
        else:
            # oracle on (local channel 3, chunk 1) with the threshold the engine derived from the all-reduced clip
            from noisereduce_amd.spectralgate.stationary import SpectralGateStationary  # noqa: F401
            thr_t = engine_gate().noise_threshold_tensor().cpu().numpy()
            c, ich = 3, 1
            s0 = ich * CHUNK
            chunk = y2d[c, s0 - PAD:s0 + CHUNK + PAD].cpu().numpy().astype(np.float64)[None, :]
            ref = O.gate_stationary_S(chunk, thr_t, NFFT, NFFT, HOP, 1.0, O.smoothing_filter(5, 9))[0, PAD:PAD + CHUNK]
            info["rel_err_unit"] = O.rel_err(out[c, s0:s0 + CHUNK].cpu().numpy(), ref)
        # (2b) configs[1]: how long a rank's stream sits in the exchange INSIDE a step (CUDA events around it, 20 steps):
        # rank 0 enters after its noise statistics, the others at once -- their time in it is the exposed wait
        if wl == "config2":
            evs = []
            for _ in range(20):
                ts_objs[0].run(y2d, ext=y_ext, defer_check=True, timing=evs)
            torch.cuda.synchronize(device)
            ts_objs[0].finish()
            info["exchange_in_step_ms"] = float(np.median([a.elapsed_time(b) for a, b in evs])) if evs else None
        if wl == "config4":
            # the same for configs[3]: events around the all-reduce of the clip sum inside 10 steps
            evs = []
            for _ in range(10):
                step(timing=evs)
            torch.cuda.synchronize(device)
            info["exchange_in_step_ms"] = float(np.median([a.elapsed_time(b) for a, b in evs])) if evs else None
            # (2c) the SAME per-GPU share without the exchange (this rank's channels as a recording of their own: no
            # collective, threshold from the local channel mean) -- the N = 1 figure of this workload measured in the same
            # process, so that value / (N x this) is the scaling factor even when the driver only ran this N
            from noisereduce_amd.sharded import ChannelShardedStationary as _CS
            solo = _CS(backend)
            solo.ws = 1
            for _ in range(3):
                solo.run(y2d)
            dist.barrier()
            torch.cuda.synchronize(device)
            ts0 = time.perf_counter()
            for _ in range(args.steps):
                solo.run(y2d)
            torch.cuda.synchronize(device)
            info["solo_ms_per_step"] = round((time.perf_counter() - ts0) / args.steps * 1e3, 4)
        # (3) collective time per step: the exchange alone, same payload, 20 repetitions
        torch.cuda.synchronize(device)
        dist.barrier()
        tc0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            if wl == "config4":
                cs_ = y2d[:, :CHUNK].to(torch.float64).sum(dim=0)
                dist.all_reduce(cs_, op=dist.ReduceOp.SUM)
            else:
                with_halos(y2d, PAD, ext=y_ext)
        torch.cuda.synchronize(device)
        info["collective_ms"] = (time.perf_counter() - tc0) / reps * 1e3
        gathered = [None] * world
        dist.all_gather_object(gathered, info)
        distributed = {"world_size": world, "backend": pg_backend + (" (RCCL)" if pg_backend == "nccl" else ""),
                       "collective": ("all_reduce(sum) of the noise clip's channel sum, %d float64" % CHUNK)
                       if wl == "config4" else "all_gather of [shard length | 2*padding seam samples | threshold] per rank",
                       "collective_ms_per_step_max": round(max(g["collective_ms"] for g in gathered), 4),
                       # time ranks > 0 spend in the in-step exchange beyond rank 0's own: waiting for its statistics
                       "exposed_wait_ms": (round(max(g["exchange_in_step_ms"] for g in gathered[1:]) - gathered[0]["exchange_in_step_ms"], 4)
                                           if all(g.get("exchange_in_step_ms") is not None for g in gathered) and len(gathered) > 1 else None),
                       "ms_per_step_per_rank": [g.get("ms_per_step") for g in gathered],
                       "per_rank": gathered}
        if wl == "config4" and all(g.get("solo_ms_per_step") for g in gathered):
            solo_ms = max(g["solo_ms_per_step"] for g in gathered)
            distributed["single_gpu_same_share"] = {
                "ms_per_step": solo_ms, "Msamples_s": round(samples_per_gpu / (solo_ms * 1e-3) / 1e6, 1),
                "what": "one GPU's 8 channels x 30 min gated WITHOUT the exchange in the same process (slowest rank, %d steps): "
                        "the N = 1 point of this weak-scaling workload" % args.steps,
                "throughput_factor_vs_it": round(world * solo_ms / (elapsed / args.steps * 1e3), 3)}
        bad = [g for g in gathered if g.get("halo_ok") is False or max(g.get("rel_err_chunk0", 0), g.get("rel_err_unit", 0)) > 1e-4]
        if bad and rank == 0:
            print("bench.py: distributed parity check FAILED: %r" % (gathered,), file=sys.stderr)
        distributed["parity_ok"] = not bad

    if rank == 0:
        total = float(samples_per_gpu) * world * args.steps
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
        steps_profiled = len(sampled)
        launches_per_step = agg[dom][1] / steps_profiled
        algo_bytes = ALGO_BYTES_PER_SAMPLE * samples_per_gpu / launches_per_step
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        traffic_src = None
        if wl == "config2" and os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(dom)
                traffic_src = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                               "workload (FETCH_SIZE x2 corrected), committed -- NOT measured in this run")
            except Exception:
                traffic = None
        desc = {"config2": "configs[1]: synthetic 48 kHz mono 10 min per GPU, stationary reduce_noise",
                "config3": "configs[2]: synthetic 48 kHz mono 10 min per GPU, non-stationary reduce_noise",
                "config4": "configs[3]: synthetic 48 kHz, 8 channels x 30 min per GPU (64 channels on 8 GPUs), "
                           "stationary reduce_noise"}[wl]
        line = {
            "metric": "Msamples/s reduce_noise (48 kHz mono, n_fft=1024)" if wl != "config4"
                      else "Msamples/s reduce_noise (48 kHz 64-channel, n_fft=1024)",
            "value": round(value, 1), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc + ", n_fft=1024 hop=256, chunk_size=600000 padding=30000, "
                                          "float32 in/out resident in HBM",
                       "samples_per_gpu": samples_per_gpu,
                       "sharding": "channels (8 per GPU), all-reduce of the clip's channel sum" if wl == "config4"
                                   else "time (chunk-aligned), seam all-gather",
                       "calls_in_flight": n_streams,
                       **({"gate_form": "one ticket-drawn tile per workgroup (SG_OPT_TILE_ORDER 2: serving mode)"}
                          if n_streams > 1 and wl == "config2" else {})},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(algo_bytes),
                         "events": "HIP events around this kernel on every %d-th step of the timed region (%d launches)"
                                   % (E, agg[dom][1]),
                         "whole_step_frac": round(value * 1e6 / world * ALGO_BYTES_PER_SAMPLE / 1e9
                                                  / HBM_PEAK_GBS, 5),
                         # the binding resource of this kernel is the float32 vector pipe, not HBM:
                         "valu": {"achieved": round(ALGO_FLOPS_PER_SAMPLE * samples_per_gpu / launches_per_step / (avg_ms * 1e-3) / 1e12, 2),
                                  "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(ALGO_FLOPS_PER_SAMPLE * samples_per_gpu / launches_per_step / (avg_ms * 1e-3) / 1e12
                                                / VALU_PEAK_TFLOPS, 4),
                                  "basis": "algorithmic flops (%d per sample, SURVEY.md 8(d)) / dominant-kernel time"
                                           % ALGO_FLOPS_PER_SAMPLE}},
            # per-kernel table of the untimed SURVEY passes (per_stage_ms: one stage bracketed at a time, 4 steps each).
            # `raw` is what the events say; an empty event pair already measures `event_pair_overhead_ms`, which
            # `kernel_ms_per_step` has subtracted per launch -- so the kernels of a step add up to no more than the step
            "kernel_ms_per_step": {k: round(v[0], 4) for k, v in sorted(survey.items(), key=lambda kv: -kv[1][0])},
            "kernel_ms_per_step_raw": {k: round(v[2], 4) for k, v in sorted(survey.items(), key=lambda kv: -kv[1][0])},
            "kernel_launches_per_step": {k: v[1] for k, v in sorted(survey.items(), key=lambda kv: -kv[1][0])},
            "event_pair_overhead_ms": round(ev_oh, 5),
        }
        line["kernel_ms_sum"] = round(sum(line["kernel_ms_per_step"].values()), 4)
        line["kernel_ms_sum_le_step"] = bool(line["kernel_ms_sum"] <= line["ms_per_step"])
        line["kernel_ms_note"] = ("event-bracketed kernel times (one stage at a time, empty-pair time subtracted) still include the "
                                  "launch / drain bubble a kernel has when events fence it (~3-5 % of a 0.25 ms kernel: the gate reads "
                                  "0.262-0.265 ms here, 0.252 ms in rocprofv3's kernel trace, profiles/); back to back in a step the bubbles "
                                  "overlap, so the sum may exceed ms_per_step by a few percent.  The roofline uses the dominant kernel's "
                                  "time from the TIMED region; rocprofv3 averages are committed under profiles/")
        line["roofline"]["avg_launch_ms_corrected"] = round(max(0.0, avg_ms - ev_oh), 4)
        if wl == "config3":
            # four kernels of comparable weight share this call's work (k_mag_fast, k_iir_chain_par, k_iir_mask, k_apply_fast):
            # whole-call algorithmic bytes over ONE of them would flatter it 3-4 x.  `frac` / `achieved`
            # are the WHOLE CALL's; the dominant kernel's own launch time stays in avg_launch_ms
            rl = line["roofline"]
            rl["dominant_kernel_only"] = {"achieved": rl["achieved"], "frac": rl["frac"],
                                          "note": "whole-call bytes / this one kernel's time: NOT a roofline fraction of the call"}
            whole = value * 1e6 / world * ALGO_BYTES_PER_SAMPLE / 1e9
            rl["achieved"], rl["frac"] = round(whole, 2), round(whole / HBM_PEAK_GBS, 5)
            rl["basis"] = "whole call: algorithmic bytes of a step / ms_per_step (several kernels of comparable weight per call)"
            rl["valu"]["frac"] = round(ALGO_FLOPS_PER_SAMPLE * value * 1e6 / world / 1e12 / VALU_PEAK_TFLOPS, 4)
            rl["valu"]["achieved"] = round(ALGO_FLOPS_PER_SAMPLE * value * 1e6 / world / 1e12, 2)
            rl["valu"]["basis"] = "algorithmic flops (%d per sample) / whole step" % ALGO_FLOPS_PER_SAMPLE
        line["gpu_state"] = {"before_settle": state_before_settle, "before_timed": state_before, "after_timed": state_after,
                             "source": "librocm_smi64 in-process (sclk / mclk of the current DPM level, socket power, cap, "
                                       "junction temperature); read outside the timed region's events, microseconds each"}
        line["settle"] = {"steps": settle_steps, "ms_per_step_blocks": settle_blocks,
                          "rule": "N = 1: blocks of ~60 ms of steps until two consecutive blocks agree within 0.5 % (cap 1.2 s), then 50 "
                                  "steps to refill the queue; N > 1: fixed count (every rank must run the same collectives)"}
        line["value_basis"] = ("value = samples of all %d timed steps / wall time between the two barrier+synchronize pairs (the "
                               "contract); value_at_median_step = the same from the median step-boundary-event interval -- "
                               "higher when the first block after the opening synchronize is the outlier" % args.steps)
        if per_step:
            line["ms_per_step_median"] = round(float(np.median(per_step)), 4)
            line["ms_per_step_min"] = round(float(np.min(per_step)), 4)
            line["value_at_median_step"] = round(samples_per_gpu * world / (float(np.median(per_step)) * 1e-3) / 1e6, 1)
            line["ms_per_step_all"] = [round(float(t), 4) for t in per_step]
            line["host_enqueue_ms_per_step"] = [round(t, 4) for t in host_ms]
        if distributed:
            line["distributed"] = distributed
        if world == 1 and not args.no_extras:
            line.update(extras(device, wl, out, y2d, gate, O))
        line["settle_steps_untimed"] = settle_steps
        line["host_gc"] = {"pauses_over_1ms": [(g_, round(ms_, 2)) for g_, ms_ in GC_PAUSES if ms_ > 1.0],
                           "note": "CPython collector pauses on the enqueueing thread since the warm-up (generation, ms); "
                                   "objects alive after the warm-up are frozen (gc.freeze)"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _roofline_of(gate, fn, samples, ms_median, traffic_keys):
    """`roofline` of one of the other configs: algorithmic bytes / median call time against the HBM peak, the dominant
    kernel of the call (HIP events around every launch of 5 calls), the float32-vector fraction of that kernel, and the
    committed PMC traffic of the call's kernels (TRAFFIC_DETAIL: NOT measured in this run)."""
    ev_oh = event_pair_overhead_ms(100)
    st = per_stage_ms(gate, fn, reps=4, ev_oh=ev_oh)     # one stage bracketed at a time
    pr = {k: (v[0] * 5, v[1] * 5) for k, v in st.items()}
    raw = {k: v[2] for k, v in st.items()}
    dom = max(pr, key=lambda k: pr[k][0])
    dom_ms = pr[dom][0] / max(pr[dom][1], 1)
    algo = ALGO_BYTES_PER_SAMPLE * samples
    r = {"bound": "hbm", "achieved": round(algo / (ms_median * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(algo / (ms_median * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "basis": "whole call (median of the HIP-event-timed "
         "repetitions): this config runs several kernels per call", "algorithmic_bytes_per_call": int(algo),
         "kernel": dom, "kernel_avg_launch_ms": round(dom_ms, 4),
         "kernel_ms_per_call": {k: round(v[0] / 5, 4) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])},
         "kernel_ms_per_call_raw": {k: round(v, 4) for k, v in sorted(raw.items(), key=lambda kv: -kv[1])},
         "event_pair_overhead_ms": round(ev_oh, 5),
         "kernel_ms_sum_le_call": bool(sum(v[0] / 5 for v in pr.values()) <= ms_median),
         "valu": {"achieved": round(ALGO_FLOPS_PER_SAMPLE * samples / (ms_median * 1e-3) / 1e12, 2), "peak": VALU_PEAK_TFLOPS,
                  "unit": "TFLOP/s", "frac": round(ALGO_FLOPS_PER_SAMPLE * samples / (ms_median * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 4),
                  "basis": "algorithmic flops (%d per sample, SURVEY.md 8(d)) / whole call" % ALGO_FLOPS_PER_SAMPLE},
         "traffic": None, "traffic_source": None}
    try:
        td = json.load(open(os.path.join(ROOT, TRAFFIC_DETAIL)))["kernels"]
        tot, used = 0, []
        for key in traffic_keys:
            for kn, v in td.items():
                if kn.startswith(key):
                    tot += v["total_bytes"]
                    used.append(kn)
        if used:
            r["traffic"] = int(tot)
            r["traffic_over_algorithmic"] = round(tot / algo, 2)
            r["traffic_source"] = TRAFFIC_DETAIL + ": per-launch FETCH_SIZE (x2 corrected) + WRITE_SIZE of " + ", ".join(used) + \
                " -- committed rocprofv3 --pmc passes, NOT measured in this run"
    except Exception:
        pass
    return r


def _time_events(fn, warm, reps, settle_ms=80.0):
    """median / mean milliseconds per call of fn() over 5 blocks of reps / 5 back-to-back calls, HIP events between the
    blocks on the current stream, plus the GPU state before / after and the settle history.  After `warm` calls (timed as a
    block on the host to size the next loop) the leg SETTLES: blocks of back-to-back calls (~60 ms each) until two
    consecutive blocks agree within 0.5 % (cap ~1.2 s) -- these legs follow seconds of CPU work (the oracle), and a GPU that
    idled that long runs its first milliseconds at reduced clocks (round 4: the driver's box needed more than the fixed
    80 ms this function used to run)."""
    t0 = time.perf_counter()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(warm):       # (the first batch paid one-time setup: size the blocks from a second one)
        fn()
    torch.cuda.synchronize()
    per_call = max((time.perf_counter() - t0) / max(1, warm), 1e-5)
    st0 = gpu_state()
    blk = int(min(2000, max(4, settle_ms * 0.75e-3 / per_call)))    # ~60 ms per block
    hist, prev, t_set = [], None, time.perf_counter()
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(blk):
            fn()
        e1.record()
        e1.synchronize()
        cur = e0.elapsed_time(e1) / blk
        hist.append(round(cur, 4))
        if prev is not None and abs(cur - prev) <= 0.005 * prev:
            break
        if time.perf_counter() - t_set > 1.2:
            break
        prev = cur
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for e in ev:
        e.record()
    for _ in range(min(blk, 40)):      # refill the queue after the settle loop's last synchronize
        fn()
    # blocks of calls between consecutive events (an event per call costs queue time of its own: the non-stationary call
    # measured 0.569 ms with one, 0.529 ms back to back)
    blocks = 5 if reps >= 20 else 1
    per = reps // blocks
    for b_ in range(blocks):
        ev[b_].record()
        for _ in range(per):
            fn()
    ev[blocks].record()
    torch.cuda.synchronize()
    st1 = gpu_state()
    ts = [ev[i].elapsed_time(ev[i + 1]) / per for i in range(blocks)]
    _time_events.last = {"settle_ms_per_call_blocks": hist, "settle_calls_per_block": blk,
                         "gpu_state_before": st0, "gpu_state_after": st1}
    return float(np.median(ts)), float(np.mean(ts)), [round(t, 4) for t in ts]


_time_events.last = {}


def extras(device, wl, out, y2d, gate, O):
    """Legs the driver cannot otherwise see (all OUTSIDE the timed region): parity of THIS run's output
    against the oracle on one chunk, the other BASELINE configs (device-resident), the PCIe-inclusive rate."""
    import noisereduce_amd as nr
    from noisereduce_amd.spectralgate.nonstationary import iir_coefficient
    from noisereduce_amd.torchgate import TorchGate
    res = {}
    # -- parity of the run just timed: one whole chunk (chunk 1: both halos are real samples)
    filt = O.smoothing_filter(5, 9)
    ich = 1
    yh = y2d[0, ich * CHUNK - PAD:(ich + 1) * CHUNK + PAD].cpu().numpy().astype(np.float64)[None, :]
    if wl == "config3":
        ref = O.gate_nonstationary_S(yh, NFFT, NFFT, HOP, 1.0, filt, iir_coefficient(2.0, SR, HOP), 2, 10)
        thr_err = None
    else:
        if wl == "config4":
            clip = y2d[:, :CHUNK].cpu().numpy().astype(np.float64)
        else:
            clip = y2d[:1, :CHUNK].cpu().numpy().astype(np.float64)
        thr, _, _ = O.noise_threshold_S(clip, NFFT, NFFT, HOP, 1.5, CHUNK)
        thr_err = float(np.max(np.abs(gate.get_noise_threshold() - thr)))
        ref = O.gate_stationary_S(yh, thr, NFFT, NFFT, HOP, 1.0, filt)
    got = out[0, ich * CHUNK:(ich + 1) * CHUNK].cpu().numpy()
    res["parity"] = {"rel_err_vs_oracle": O.rel_err(got, ref[0, PAD:PAD + CHUNK]), "tolerance": 1e-4,
                     "checked": "channel 0, chunk 1 (600000 samples) of the last timed step, oracle float64",
                     "threshold_max_abs_err_db": thr_err}
    if wl != "config2":
        return res
    oc = {}
    y = y2d[0]
    med, mean, reps3 = _time_events(lambda: nr.reduce_noise(y=y, sr=SR, stationary=False), 5, 40)
    oc["config3_nonstationary"] = {"ms_median": round(med, 4), "ms_mean": round(mean, 4), "ms_per_call_blocks": reps3,
                                   "Msamples_s": round(y.numel() / (med * 1e-3) / 1e6, 1),
                                   "what": "configs[2]: same recording, stationary=False, device-resident; 5 blocks of 8 "
                                           "back-to-back calls, median of the blocks' per-call times", **_time_events.last}
    try:
        from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
        ns = SpectralGateNonStationary(y=y, sr=SR, chunk_size=CHUNK, padding=PAD, n_fft=NFFT, win_length=None, hop_length=None,
                                       time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
                                       thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None,
                                       prop_decrease=1.0, use_tqdm=False, n_jobs=1, device=device)
        oc["config3_nonstationary"]["roofline"] = _roofline_of(
            ns._gate, lambda: nr.reduce_noise(y=y, sr=SR, stationary=False), y.numel(), med,
            ["k_mag_fast", "k_iir_comb", "k_iir_chain", "k_iir_mask", "k_apply_fast<4, false, true"])
    except Exception as e:
        oc["config3_nonstationary"]["roofline"] = {"error": repr(e)}
    torch.manual_seed(0)
    t = torch.arange(16000, device=device, dtype=torch.float64) / 16000
    x = (0.1 * torch.randn(256, 16000, device=device) + 0.5 * torch.sin(2 * np.pi * 440 * t).float()).float()
    tg = TorchGate(sr=16000).to(device)
    med, mean, reps5 = _time_events(lambda: tg(x), 10, 50)
    oc["config5_torchgate_forward"] = {"ms_median": round(med, 4), "ms_mean": round(mean, 4),
                                       "ms_max": max(reps5),
                                       "Msamples_s": round(x.numel() / (med * 1e-3) / 1e6, 1),
                                       "what": "configs[4]: TorchGate(sr=16000) on 256 x 16000 float32; 5 blocks of 10 "
                                               "back-to-back calls, median of the blocks' per-call times", **_time_events.last}
    try:
        (tgate,) = list(tg._gates.values())
        rf = _roofline_of(tgate, lambda: tg(x), x.numel(), med, ["k_row_gate"])
        # the committed PMC pass of k_row_gate ran WITH the float mask output (forward under autograd); this leg is a forward
        # without grad, which does not write it (torchgate.py: save_mask only when the input requires grad): subtract the
        # mask field's bytes from the quoted traffic and say so
        if rf.get("traffic"):
            mask_bytes = 256 * 64 * 528 * 4
            rf["traffic_with_mask_output"] = rf["traffic"]
            rf["traffic"] = int(max(0, rf["traffic"] - mask_bytes))
            rf["traffic_over_algorithmic"] = round(rf["traffic"] / rf["algorithmic_bytes_per_call"], 2)
            rf["traffic_source"] += ("; minus the %d-byte float mask ([256][64][528] float32) that only a forward under autograd "
                                     "writes -- this leg runs without grad" % mask_bytes)
        oc["config5_torchgate_forward"]["roofline"] = rf
        oc["config5_torchgate_forward"]["exact_pairs_per_call"] = (lambda c0: (tg(x), tgate.debug_counter(0) - c0)[1])(tgate.debug_counter(0))
    except Exception as e:
        oc["config5_torchgate_forward"]["roofline"] = {"error": repr(e)}
    xg = x.clone().requires_grad_()

    def fb():
        xg.grad = None
        tg(xg).sum().backward()
    med, mean, reps5b = _time_events(fb, 10, 50)
    oc["config5_torchgate_forward_backward"] = {"ms_median": round(med, 4), "ms_mean": round(mean, 4),
                                                "ms_max": max(reps5b),
                                                "Msamples_s": round(x.numel() / (med * 1e-3) / 1e6, 1), **_time_events.last}
    # other STFT geometries, the first two minutes of the same recording, device-resident
    try:
        y2 = y[:SR * 120].contiguous()
        og = {}
        for n_fft in (256, 512, 2048):
            for stat in (True, False):
                med, mean, blocks = _time_events(lambda: nr.reduce_noise(y=y2, sr=SR, stationary=stat, n_fft=n_fft), 5, 20)
                og["n_fft=%d,%s" % (n_fft, "stationary" if stat else "non-stationary")] = {
                    "ms_median": round(med, 4), "ms_per_call_blocks": blocks,
                    "Msamples_s": round(y2.numel() / (med * 1e-3) / 1e6, 1),
                    "settle_ms_per_call_blocks": _time_events.last.get("settle_ms_per_call_blocks")}
        og["what"] = ("reduce_noise(n_fft=...) on 2 minutes (5.76 M samples, 10 chunks) of the benchmark recording: n_fft = 256 on "
                      "fast256.hpp (four frames per register transform), 512 / 2048 on fast512.hpp / fast2048.hpp; stationary: the one-pass gates of round 6 (onepass256 / 512 / 2048.hpp); "
                      "5 blocks of 4 back-to-back calls after the adaptive settle, median of the blocks")
        oc["other_geometries_2min"] = og
    except Exception as e:
        oc["other_geometries_2min"] = {"error": repr(e)}
    # the reference's OWN arithmetic and test input: it computes every dtype in float64 (base.py:140) and its
    # test recording is int16 (test_reduction.py:8).  Ten minutes, device-resident, same settle protocol as the other legs.
    try:
        dt = {}
        y16 = (y * 20000.0).to(torch.int16)
        y64 = y.double()
        for name, yy, prec in (("int16", y16, None), ("float64,precision=float64", y64, "float64")):
            for stat in (True, False):
                med, mean, blocks = _time_events(lambda: nr.reduce_noise(y=yy, sr=SR, stationary=stat, precision=prec), 3, 10)
                dt["%s,%s" % (name, "stationary" if stat else "non-stationary")] = {
                    "ms_median": round(med, 4), "ms_per_call_blocks": blocks,
                    "Msamples_s": round(yy.numel() / (med * 1e-3) / 1e6, 1),
                    "settle_ms_per_call_blocks": _time_events.last.get("settle_ms_per_call_blocks")}
        del y16, y64
        dt["what"] = ("configs[1] / configs[2] on the same ten minutes held as int16 (x 20000; integer results are trunc() of the "
                      "float64 result = the reference's integers: float64 transforms and overlap-add on the exact bit-path mask, "
                      "k_apply_fast64) and as float64 with precision='float64' (float64 pipeline: <= 1e-12 of peak against the "
                      "reference); adaptive settle, then 10 back-to-back calls between HIP events")
        oc["dtypes_10min"] = dt
    except Exception as e:
        oc["dtypes_10min"] = {"error": repr(e)}
    # PCIe-inclusive: numpy in -> numpy out (H2D + compute + D2H), wall clock
    yh = y.cpu().numpy()
    ts = []
    for i in range(7):
        t0 = time.perf_counter()
        nr.reduce_noise(y=yh, sr=SR, stationary=True)
        ts.append(time.perf_counter() - t0)
    medh = float(np.median(ts[2:]))
    oc["config2_numpy_to_numpy_pcie_inclusive"] = {
        "ms_median": round(medh * 1e3, 3), "Msamples_s": round(yh.size / medh / 1e6, 1),
        "what": "host float32 array in, host array out: the recording goes up in pieces while the previous piece is gated "
                "with its output stored straight into the page-locked result (spectralgate/base.py _get_traces_pipelined; "
                "profiles/r05_pcie_overlap.txt: 56 GB/s one way alone, ~40 GB/s each way at once); never the headline value"}
    os.environ["NOISEREDUCE_AMD_PIPELINE"] = "0"
    try:
        ts = []
        for i in range(7):
            t0 = time.perf_counter()
            nr.reduce_noise(y=yh, sr=SR, stationary=True)
            ts.append(time.perf_counter() - t0)
        oc["config2_numpy_to_numpy_pcie_inclusive"]["ms_median_one_upload"] = round(float(np.median(ts[2:])) * 1e3, 3)
    finally:
        del os.environ["NOISEREDUCE_AMD_PIPELINE"]
    res["other_configs"] = oc
    return res


if __name__ == "__main__":
    main()
