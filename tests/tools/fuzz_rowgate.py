#!/usr/bin/env python3
"""Randomised sweep of the one-kernel TorchGate row gate (rowgate.hpp) against the four-kernel float64 path: the mask bits
must be IDENTICAL on every cell (the float32 statistics' error bound is a statistical claim -- this is its test), the
outputs within 1e-6 of each other and, on a subset, within the 1e-4 bar of the CPU oracle.
usage (GPU box): python tests/tools/fuzz_rowgate.py [first_seed] [count]   -> gpurun_out/fuzz_rowgate.json"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from noisereduce_amd import _ffi
from noisereduce_amd.torchgate import TorchGate
from oracle import spectralgate_oracle as O

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = "cuda:0"
tot = dict(cases=0, rowgate_cases=0, cells=0, flips=0, exact_pairs=0, pairs=0, worst_vs_f64=0.0, worst_vs_oracle=0.0, failures=[])


def signal(rng, B, L, sr):
    t = np.arange(L) / sr
    kind = rng.integers(0, 7)
    x = rng.uniform(1e-4, 0.3) * rng.standard_normal((B, L))
    if kind in (1, 2, 5):
        for _ in range(rng.integers(1, 4)):
            x += rng.uniform(0.01, 0.9) * np.sin(2 * np.pi * rng.uniform(50, sr / 2 - 50) * t + rng.uniform(0, 6.28))
    if kind == 2:
        x *= (0.05 + np.abs(np.sin(2 * np.pi * rng.uniform(0.5, 8) * t)))            # amplitude modulation
    if kind == 3:
        x += 0.7 * np.sin(2 * np.pi * (rng.uniform(50, 500) * t + rng.uniform(500, 3000) * t * t))   # chirp
    if kind == 4:
        x[:, :: int(rng.integers(200, 2000))] += rng.uniform(0.2, 1.0)                 # impulse train
    if kind == 5:
        x[rng.integers(0, B)] = 0.0                                                    # a silent row
    if kind == 6:
        x *= 10.0 ** rng.uniform(-6, 1)                                                # very quiet ... loud
        x += rng.uniform(-0.1, 0.1)                                                    # DC offset
    return x


for seed in range(first, first + count):
    rng = np.random.default_rng(90000 + seed)
    sr = int(rng.choice([11025, 16000, 16000, 22050, 32000, 44100, 48000]))
    L = int(rng.integers(2048, 16384))
    B = int(rng.integers(1, 48))
    kw = dict(n_std_thresh_stationary=float(rng.uniform(0.3, 3.0)))
    if rng.random() < 0.5:
        kw["freq_mask_smooth_hz"] = float(rng.uniform(sr / 512 * 1.01, min(900.0, sr / 512 * 29)))
    if rng.random() < 0.5:
        kw["time_mask_smooth_ms"] = float(rng.uniform(256e3 / sr * 1.01, 256e3 / sr * 15))
    dtype = torch.float64 if rng.random() < 0.2 else torch.float32
    x = torch.from_numpy(signal(rng, B, L, sr)).to(dtype)
    try:
        tg = TorchGate(sr=sr, **kw).to(dev)
        xd = x.to(dev)
        tg(xd)
        (g,) = list(tg._gates.values())
        c0 = g.debug_counter(0)
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 2)
        g.set_option(_ffi.SG_OPT_ROWGATE_SHAPE, 8 if seed % 5 == 4 else 16)
        g.profile_read(reset=True); g.profile_enable(True)
        y_new = tg(xd).clone()
        used = any("k_row_gate" in k for k in g.profile_read(reset=True)); g.profile_enable(False)
        if not used:      # shape not eligible (filter weight total beyond uint16, width beyond the kernel's window): old path
            g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0); g.set_option(_ffi.SG_OPT_ROWGATE_SHAPE, 16)
            tot["cases"] += 1
            continue
        b_new = g.debug_field(3)
        c1 = g.debug_counter(0)
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 1)
        y_old = tg(xd).clone()
        b_old = g.debug_field(3)
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0); g.set_option(_ffi.SG_OPT_ROWGATE_SHAPE, 16)
        tot["cases"] += 1
        tot["rowgate_cases"] += 1
        flips = int((b_new != b_old).sum())
        tot["cells"] += int(b_new.size); tot["flips"] += flips
        tot["exact_pairs"] += c1 - c0; tot["pairs"] += B * 513
        fin = torch.isfinite(y_old)
        peak = float(y_old[fin].abs().max()) if fin.any() else 0.0
        e = float((y_new[fin] - y_old[fin]).abs().max() / peak) if peak > 0 else 0.0
        tot["worst_vs_f64"] = max(tot["worst_vs_f64"], e)
        bad = flips > 0 or e > 1e-6 or not torch.equal(torch.isnan(y_new), torch.isnan(y_old))
        if seed % 8 == 0 and B <= 12:
            want = O.torchgate_T(x.numpy().astype(np.float64), sr, window=torch.hann_window(1024).double().numpy(),
                                 **{k: v for k, v in kw.items()})
            eo = O.rel_err(y_new.cpu().numpy(), want)
            tot["worst_vs_oracle"] = max(tot["worst_vs_oracle"], float(eo))
            bad = bad or eo > 1e-4
        if bad:
            tot["failures"].append(dict(seed=seed, sr=sr, L=L, B=B, kw=kw, flips=flips, err=e))
            print("FAIL", tot["failures"][-1], flush=True)
    except Exception as ex:      # noqa: BLE001
        tot["failures"].append(dict(seed=seed, error=repr(ex)))
        print("ERROR", seed, repr(ex), flush=True)
    if (seed - first) % 50 == 49:
        print(seed, {k: v for k, v in tot.items() if k != "failures"}, "failures", len(tot["failures"]), flush=True)
tot["exact_rate"] = tot["exact_pairs"] / max(tot["pairs"], 1)
tot["seeds"] = [first, first + count]
print(json.dumps(tot, indent=1))
json.dump(tot, open(os.path.join(ROOT, "gpurun_out", "fuzz_rowgate.json"), "w"), indent=1)
