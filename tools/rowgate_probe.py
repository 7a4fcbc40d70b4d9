#!/usr/bin/env python3
"""GPU probe of the one-kernel TorchGate row gate (rowgate.hpp): decisions against the float64 four-kernel path
(SG_OPT_FORCE_NOROWGATE), output against the CPU oracle, exact-path rate, HIP-event timing of both paths."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noisereduce_amd import _ffi
from noisereduce_amd.torchgate import TorchGate
from oracle import spectralgate_oracle as O

res = {}
dev = "cuda:0"


def gate_of(tg):
    (g,) = list(tg._gates.values())
    return g


def run_case(name, x, sr=16000, check_oracle=True):
    tg = TorchGate(sr=sr).to(dev)
    xd = x.to(dev)
    tg(xd)
    g = gate_of(tg)
    g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 2)      # the row gate whatever the batch size
    y_new = tg(xd).clone()
    bits_new = g.debug_field(3)
    n_ex = g.debug_counter(0)
    g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 1)
    y_old = tg(xd).clone()
    bits_old = g.debug_field(3)
    g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0)
    fin = torch.isfinite(y_old)
    peak = float(y_old[fin].abs().max()) if fin.any() else 1.0
    d = {"shape": list(x.shape), "dtype": str(x.dtype), "bit_flips": int((bits_new != bits_old).sum()),
         "cells": int(bits_new.size), "exact_pairs_total": n_ex,
         "nan_pattern_equal": bool(torch.equal(torch.isnan(y_new), torch.isnan(y_old))),
         "max_abs_diff_vs_old_over_peak": float((y_new[fin] - y_old[fin]).abs().max() / max(peak, 1e-30)) if fin.any() else 0.0}
    if check_oracle:
        want = O.torchgate_T(x.numpy().astype(np.float64), sr, window=torch.hann_window(1024).double().numpy())
        d["rel_err_vs_oracle"] = O.rel_err(y_new.cpu().numpy(), want)
    res[name] = d
    print(name, d, flush=True)


torch.manual_seed(0)
t16 = torch.arange(16000, dtype=torch.float64) / 16000
x = (0.1 * torch.randn(24, 16000) + 0.5 * torch.sin(2 * np.pi * 440 * t16).float()).float()
run_case("noise+tone 24x16000", x)
run_case("T=64 rows 5x16383", x[:5, :].repeat(1, 2)[:, :16383].contiguous())
run_case("short rows 7x3000", x[:7, :3000].contiguous())
run_case("float64 3x16000", x[:3].double())
xs = x[:6].clone(); xs[1] = 0; xs[3] *= 1e-7; xs[4, 5000] = float("nan")
run_case("silent / tiny / NaN rows", xs, check_oracle=False)
sp = torch.from_numpy(np.stack([O.synth_signal(16000, sr=16000, seed=s, tone_hz=300.0 + 50 * s) for s in range(16)]))
run_case("synth_signal 16x16000", sp)

# timing: configs[4]
torch.manual_seed(0)
xb = (0.1 * torch.randn(256, 16000, device=dev) + 0.5 * torch.sin(2 * np.pi * 440 * t16.to(dev)).float()).float()
tg = TorchGate(sr=16000).to(dev)


def ev_time(fn, warm=10, reps=50):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    for i in range(reps):
        ev[i].record(); fn()
    ev[reps].record(); torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2], ts[0]


med, mn = ev_time(lambda: tg(xb))
g = gate_of(tg)
c0 = g.debug_counter(0); tg(xb); c1 = g.debug_counter(0)
res["config5 forward (row gate)"] = {"ms_median": med, "ms_min": mn, "exact_pairs_per_call": c1 - c0,
                                     "exact_rate": (c1 - c0) / (256 * 513)}
g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 1)
med, mn = ev_time(lambda: tg(xb))
res["config5 forward (four kernels)"] = {"ms_median": med, "ms_min": mn}
g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0)
xg = xb.clone().requires_grad_()


def fb():
    xg.grad = None
    tg(xg).sum().backward()


med, mn = ev_time(fb)
res["config5 forward+backward (row gate)"] = {"ms_median": med, "ms_min": mn}
print(json.dumps(res, indent=1))
out = os.path.join(ROOT, "gpurun_out", "rowgate_probe.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(res, open(out, "w"), indent=1)
