#!/usr/bin/env python3
"""TEST / MEASUREMENT INFRASTRUCTURE -- not part of the product path.

Times the UNMODIFIED reference (timsainb/noisereduce, staged by oracle/make_ref.sh under oracle/_ref/) on this host's
cores and prints ONE JSON object: bench.py's `cpu_baseline` with "kind": "reference" (SURVEY.md 8(d) "CPU baseline
beside it").  bench.py runs this file as a SUBPROCESS (no HIP context, no state of the bench process);
nothing under noisereduce_amd/ imports it.

Protocol: the benchmark's own synthetic signal (`spectralgate_oracle.synth_signal`, seed 1234),
1 warm-up + median of >= 5 runs where the time budget allows (never fewer than 2 timed runs),
  * `reduce_noise(y, 48000, stationary=True)`, `use_torch=False`, n_jobs=1           (reference noisereduce.py:13-185)
  * the same, stationary=False                                                        (configs[2])
  * the same, n_jobs=os.cpu_count() on the full 10 min (48 chunks; base.py:206-216)   (joblib, loky processes)
  * reference `TorchGate(sr=16000)` on CPU tensors, 256 x 16000 float32               (configs[4], torchgate.py:200-264)
The single-core legs use the first `--seconds` of the workload (default 60 s = 5 chunks), float64 input (what the
oracle is fed); the sample is stated in every entry.
"""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")


def _cpu_model():
    try:
        for ln in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _timed(fn, min_runs, want_runs, budget_s):
    """1 warm-up, then up to `want_runs` timed runs; stops early (never below `min_runs`) once `budget_s` is spent."""
    t_begin = time.perf_counter()
    fn()
    ts = []
    while len(ts) < want_runs:
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if len(ts) >= min_runs and time.perf_counter() - t_begin > budget_s:
            break
    ts.sort()
    n = len(ts)
    med = ts[n // 2] if n % 2 else 0.5 * (ts[n // 2 - 1] + ts[n // 2])
    return med, ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=int, default=60, help="length of the single-core sample (audio seconds)")
    ap.add_argument("--budget", type=float, default=40.0, help="total wall-clock budget (s), split over the legs")
    ap.add_argument("--no-multicore", action="store_true")
    ap.add_argument("--no-torchgate", action="store_true")
    args = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, "noisereduce")):
        print(json.dumps({"error": "oracle/_ref/noisereduce not staged (run oracle/make_ref.sh where /root/reference exists)"}))
        return 3
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.dirname(HERE))
    import numpy as np
    import noisereduce as ref                                  # the staged, unmodified reference
    assert os.path.realpath(ref.__file__).startswith(os.path.realpath(REF)), ref.__file__
    from oracle import spectralgate_oracle as O                # only for the benchmark's signal generator

    SR = 48000
    ncpu = os.cpu_count() or 1
    manifest = os.path.join(REF, "MANIFEST.sha256")
    import hashlib
    res = {"unit": "Msamples/s", "kind": "reference",
           "cpu_model": _cpu_model(), "os_cpu_count": ncpu,
           "reference": {"package": "noisereduce %s" % (open(os.path.join(REF, "SOURCE.txt")).read().strip().replace("\n", "; ")),
                         "manifest_sha256": hashlib.sha256(open(manifest, "rb").read()).hexdigest()},
           "versions": {"numpy": np.__version__}}
    try:
        import scipy
        res["versions"]["scipy"] = scipy.__version__
    except Exception:
        pass
    n = SR * args.seconds
    y = O.synth_signal(n, dtype=np.float32).astype(np.float64)
    share = args.budget / 4.0

    med, ts = _timed(lambda: ref.reduce_noise(y=y, sr=SR, stationary=True, n_fft=1024, n_jobs=1), 2, 5, share)
    res["value"] = round(n / med / 1e6, 3)
    res["cores"] = 1
    res["sample"] = ("configs[1]: first %d s (%d samples, %d chunks) of the workload, float64 input, reference reduce_noise("
                     "stationary=True, n_fft=1024, use_torch=False, n_jobs=1); 1 warm-up + median of %d runs"
                     % (args.seconds, n, -(-n // 600000), len(ts)))
    res["runs_s"] = [round(t, 4) for t in ts]

    med, ts = _timed(lambda: ref.reduce_noise(y=y, sr=SR, stationary=False, n_fft=1024, n_jobs=1), 2, 5, share)
    res["nonstationary"] = {"value": round(n / med / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "reference",
                            "sample": "configs[2]: the same %d s, stationary=False, n_jobs=1; 1 warm-up + median of %d runs"
                                      % (args.seconds, len(ts)),
                            "runs_s": [round(t, 4) for t in ts]}

    if not args.no_multicore:
        try:
            nfull = SR * 600
            yfull = O.synth_signal(nfull, dtype=np.float32).astype(np.float64)
            med, ts = _timed(lambda: ref.reduce_noise(y=yfull, sr=SR, stationary=True, n_fft=1024, n_jobs=ncpu), 2, 5, share)
            res["multicore"] = {"value": round(nfull / med / 1e6, 3), "unit": "Msamples/s", "cores": ncpu, "kind": "reference",
                                "sample": "configs[1] in full (10 min, 48 chunks), reference reduce_noise(n_jobs=os.cpu_count()=%d): "
                                          "joblib workers over chunks (base.py:206-216), result through the reference's own "
                                          "memmap; 1 warm-up + median of %d runs" % (ncpu, len(ts)),
                                "runs_s": [round(t, 4) for t in ts]}
            del yfull
        except Exception as e:     # the single-core numbers stand on their own
            res["multicore"] = {"error": repr(e)}

    if not args.no_torchgate:
        try:
            import torch
            from noisereduce.torchgate import TorchGate as RefTG
            res["versions"]["torch"] = torch.__version__
            torch.manual_seed(0)
            tt = torch.arange(16000, dtype=torch.float64) / 16000
            x = (0.1 * torch.randn(256, 16000) + 0.5 * torch.sin(2 * np.pi * 440 * tt).float()).float()
            tg = RefTG(sr=16000)
            with torch.no_grad():
                med, ts = _timed(lambda: tg(x), 2, 5, share)
            res["torchgate_cpu"] = {"value": round(x.numel() / med / 1e6, 3), "unit": "Msamples/s",
                                    "cores": torch.get_num_threads(), "kind": "reference",
                                    "sample": "configs[4]: reference TorchGate(sr=16000).forward on CPU tensors, 256 x 16000 "
                                              "float32, torch's own thread pool (%d threads); 1 warm-up + median of %d runs"
                                              % (torch.get_num_threads(), len(ts)),
                                    "runs_s": [round(t, 4) for t in ts]}
        except Exception as e:
            res["torchgate_cpu"] = {"error": repr(e)}
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
