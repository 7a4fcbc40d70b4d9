#!/usr/bin/env python3
"""How fast is bench.py's CPU baseline (the numpy oracle = "port") compared with the LIVE reference?  Runs in the BUILD
container only (needs /root/reference on PYTHONPATH; the GPU box does not have it): both are timed on the same host
cores on the same inputs, results compared, ratios written to profiles/r03_ref_vs_port.json, which bench.py quotes
in cpu_baseline.  Usage: PYTHONPATH=/root/reference python tools/ref_vs_port.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import noisereduce as ref                      # the live reference
from noisereduce.torchgate import TorchGate as RefTG
import torch
from oracle import spectralgate_oracle as O
from oracle.torchgate_torch_port import torchgate_cpu


def med(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), r


res = {"host": {"cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
                "note": "build container (shared VM: timings vary run to run; the RATIOS are what bench.py quotes)"}}
n = 48000 * 27 + 4000          # 1.3 M samples: 3 chunks of the default grid
y32 = O.synth_signal(n, dtype=np.float32)
y = y32.astype(np.float64)
for stationary in (True, False):
    tr, a = med(lambda: ref.reduce_noise(y=y, sr=48000, stationary=stationary), 3)
    tp, b = med(lambda: O.reduce_noise_S(y, 48000, stationary=stationary), 3)
    res["stationary" if stationary else "nonstationary"] = {
        "samples": n, "reference_s": round(tr, 3), "port_s": round(tp, 3), "reference_Msamples_s": round(n / tr / 1e6, 3),
        "port_Msamples_s": round(n / tp / 1e6, 3), "port_over_reference_speed": round(tr / tp, 2),
        "rel_err_port_vs_reference": O.rel_err(b, a)}
    print(res, flush=True)
# configs[4]: TorchGate on the CPU, float32, 256 x 16000
torch.manual_seed(0)
t = torch.arange(16000, dtype=torch.float64) / 16000
x = (0.1 * torch.randn(256, 16000) + 0.5 * torch.sin(2 * np.pi * 440 * t).float()).float()
tg = RefTG(sr=16000)
tr, a = med(lambda: tg(x), 3)
tp, b = med(lambda: torchgate_cpu(x, 16000), 3)
tn, c = med(lambda: O.torchgate_T(x[:32].numpy().astype(np.float64), 16000, window=torch.hann_window(1024).double().numpy()), 2)
res["torchgate_256x16000_f32"] = {
    "samples": x.numel(), "reference_s": round(tr, 3), "torch_port_s": round(tp, 3),
    "reference_Msamples_s": round(x.numel() / tr / 1e6, 3), "torch_port_Msamples_s": round(x.numel() / tp / 1e6, 3),
    "torch_port_over_reference_speed": round(tr / tp, 2), "rel_err_torch_port_vs_reference": O.rel_err(b.numpy(), a.numpy()),
    "numpy_port_Msamples_s (32 rows, float64)": round(32 * 16000 / tn / 1e6, 3)}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(ROOT, "profiles", "r03_ref_vs_port.json"), "w"), indent=1)
