#!/usr/bin/env python3
"""How large is the error of the row gate's float32 transform, measured against float64 -- in units of the decision
margin delta = 2^-16 ||x w||_2 that k_row_gate / k_decide_fast / k_gate_onepass assume?  Fetches the float32 power tile
(SG_OPT_ROWGATE_TAP) for several signal families and compares |2X| with numpy's float64 rfft of the same windowed frames.
Writes gpurun_out/rowgate_margin.json (copy to profiles/)."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from noisereduce_amd import _ffi
from noisereduce_amd.torchgate import TorchGate

dev = "cuda:0"
rng = np.random.default_rng(7)
L, sr = 16000, 16000
t = np.arange(L) / sr
w = torch.hann_window(1024).double().numpy()


def families(n):
    out = {}
    out["white noise 0.1"] = 0.1 * rng.standard_normal((n, L))
    out["noise 0.1 + tone 0.5 @440"] = 0.1 * rng.standard_normal((n, L)) + 0.5 * np.sin(2 * np.pi * 440 * t)
    out["noise 1e-3 + tone 0.9 @1234.5"] = 1e-3 * rng.standard_normal((n, L)) + 0.9 * np.sin(2 * np.pi * 1234.5 * t)
    out["two tones + noise 1e-4"] = 1e-4 * rng.standard_normal((n, L)) + 0.5 * np.sin(2 * np.pi * 1000 * t) + 0.4 * np.sin(2 * np.pi * 3000.3 * t)
    ch = np.sin(2 * np.pi * (200 * t + 3000 * t * t))
    out["chirp + noise 0.01"] = 0.01 * rng.standard_normal((n, L)) + 0.7 * ch
    imp = np.zeros((n, L)); imp[:, ::997] = 1.0
    out["impulse train + noise 1e-3"] = imp + 1e-3 * rng.standard_normal((n, L))
    out["speech-like (AM noise)"] = rng.standard_normal((n, L)) * (0.05 + 0.5 * np.abs(np.sin(2 * np.pi * 3 * t)))
    return out


res = {}
tg = TorchGate(sr=sr).to(dev)
for name, x in families(32).items():
    x32 = x.astype(np.float32)
    xd = torch.from_numpy(x32).to(dev)
    tg(xd)
    (g,) = list(tg._gates.values())
    g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 2)
    g.set_option(_ffi.SG_OPT_ROWGATE_TAP, 1)
    tg(xd)
    P4 = g.debug_field(4).astype(np.float64)            # (rows, T, 513): 4 |X|^2 in float32
    g.set_option(_ffi.SG_OPT_ROWGATE_TAP, 0)
    T = P4.shape[1]
    xp = np.pad(x32.astype(np.float64), ((0, 0), (512, 512)))
    idx = 256 * np.arange(T)[:, None] + np.arange(1024)[None, :]
    fr = xp[:, idx] * w                                  # (rows, T, 1024)
    X = np.fft.rfft(fr, axis=-1)
    a64 = 2.0 * np.abs(X)
    a32 = np.sqrt(P4)
    nrm = np.sqrt((fr ** 2).sum(-1))[..., None]          # ||x w||_2 (float64; the kernel's is float32 of the same sum)
    err = np.abs(a32 - a64) / (2.0 * nrm + 1e-300)       # error of |X| in units of ||x w||
    unit = 2.0 ** -16
    res[name] = {"cells": int(err.size), "max_err_over_delta": float(err.max() / unit), "rms_err_over_delta": float(np.sqrt((err ** 2).mean()) / unit),
                 "p99.99_over_delta": float(np.quantile(err, 0.9999) / unit)}
    print(name, res[name], flush=True)
res["note"] = "error of the float32 |X| against float64, divided by delta = 2^-16 ||x w||_2 (the margin the decision kernels assume): max must stay well below 1"
out = os.path.join(ROOT, "gpurun_out", "rowgate_margin.json")
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
