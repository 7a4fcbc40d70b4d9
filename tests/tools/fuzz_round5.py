#!/usr/bin/env python3
"""Randomised sweeps of the round-5 kernels against the paths they replace.
  (a) k_row_backward vs k_env_scale + k_apply_fast (SG_OPT_FORCE_NOROWGATE 1) random row counts,
      lengths (1 .. 64 frames, ragged ends), sample rates / smoothing widths, float32 / float64, stationary / non-stationary;
  (b) the float64 pipeline (k_apply_fast64 on K counts / on the float64 mask field, tile-parallel recurrence, LDS-tiled smoothing,
      register float64 STFT) vs its round-4 form (SG_OPT_EXACT_MATERIALISED): random recordings, chunk grids, sub-ranges,
      channel counts, dtypes int16 / int32 / float64, both gates, prop_decrease.
usage (GPU box): python tests/tools/fuzz_round5.py [first_seed] [count]   -> gpurun_out/fuzz_round5.json"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from noisereduce_amd import _ffi
from noisereduce_amd.torchgate import TorchGate
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = "cuda:0"
tot = dict(bwd_cases=0, bwd_worst_vs_old=0.0, x64_cases=0, x64_worst_f64=0.0, x64_int_mismatch_decided=0,
           x64_int_max_lsb=0, failures=[])


def sig(rng, shape, sr):
    n = shape[-1]
    t = np.arange(n) / sr
    x = rng.uniform(0.01, 0.3) * rng.standard_normal(shape)
    for _ in range(rng.integers(0, 3)):
        x = x + rng.uniform(0.05, 0.8) * np.sin(2 * np.pi * rng.uniform(60, sr / 2 - 60) * t + rng.uniform(0, 6.28))
    if rng.random() < 0.2:
        x[..., : n // 3] = 0.0            # a silent stretch
    return x


for seed in range(first, first + count):
    rng = np.random.default_rng(50000 + seed)
    # ---------------------------------------------------------------- (a) backward
    try:
        sr = int(rng.choice([16000, 16000, 22050, 44100, 48000]))
        L = int(rng.integers(2048, 16640))          # up to 65 frames: the last ones fall back to the tiled kernels
        B = int(rng.integers(1, 40))
        nonstat = rng.random() < 0.25
        dtype = torch.float64 if rng.random() < 0.3 else torch.float32
        kw = dict(nonstationary=bool(nonstat))
        if rng.random() < 0.3:
            kw["prop_decrease"] = float(rng.uniform(0.3, 1.0))
        x = torch.from_numpy(sig(rng, (B, L), sr)).to(dtype).to(dev)
        tg = TorchGate(sr=sr, **kw).to(dev)
        w = torch.from_numpy(rng.standard_normal((B, 256 * (L // 256)))).to(dtype).to(dev)
        grads = []
        tg(x)
        (g,) = list(tg._gates.values())
        for mode in (0, 1):
            g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, mode)
            xg = x.clone().requires_grad_()
            (tg(xg) * w).sum().backward()
            grads.append(xg.grad.detach().double().cpu().numpy())
        g.set_option(_ffi.SG_OPT_FORCE_NOROWGATE, 0)
        # (mode 0 and mode 1 may pick different forward kernels: their masks are bit-identical, outputs equal to ~1e-7)
        # (a silent stretch makes the non-stationary mask NaN there, like the reference's 0 / 0: the gradients are NaN in the
        # same places on both paths)
        fin0, fin1 = np.isfinite(grads[0]), np.isfinite(grads[1])
        same_nan = bool(np.array_equal(fin0, fin1))
        both = fin0 & fin1
        e = float(np.abs(grads[0][both] - grads[1][both]).max() / max(np.abs(grads[1][both]).max(), 1e-30)) if both.any() else 0.0
        tot["bwd_worst_vs_old"] = max(tot["bwd_worst_vs_old"], e)
        tot["bwd_cases"] += 1
        tot["bwd_cases_with_nan"] = tot.get("bwd_cases_with_nan", 0) + int(not fin1.all())
        if not (e < 1e-5) or not same_nan:
            tot["failures"].append(("bwd", seed, sr, B, L, str(dtype), kw, e, same_nan, int((~fin0).sum()), int((~fin1).sum())))
    except Exception as ex:  # noqa: BLE001
        tot["failures"].append(("bwd-exc", seed, repr(ex)))
    # ---------------------------------------------------------------- (b) float64 pipeline
    try:
        sr = int(rng.choice([44100, 48000, 48000, 32000]))
        C = int(rng.integers(1, 4))
        n = int(rng.integers(30000, 140000))
        cs = int(rng.choice([600000, 40000, 25000, 17000]))
        pad = int(rng.choice([30000, 4000, 3000, 1500])) if cs < 600000 else 30000
        stationary = rng.random() < 0.5
        dtype = rng.choice([np.int16, np.int32, np.float64])
        scale = {np.int16: 20000.0, np.int32: 1.2e9, np.float64: 1.0}[dtype]
        y = sig(rng, (C, n), sr) * scale
        y = (np.round(y) if dtype != np.float64 else y).astype(dtype)
        if C == 1 and rng.random() < 0.5:
            y = y[0]
        base = dict(sr=sr, prop_decrease=1.0 if rng.random() < 0.6 else float(rng.uniform(0.2, 0.95)), chunk_size=cs, padding=pad,
                    n_fft=1024, win_length=None, hop_length=None, time_constant_s=float(rng.choice([2.0, 0.5, 1.0])),
                    freq_mask_smooth_hz=float(rng.choice([500, 300, 900])), time_mask_smooth_ms=float(rng.choice([50, 30, 80])),
                    tmp_folder=None, use_tqdm=False, n_jobs=1)
        if stationary:
            sg = SpectralGateStationary(y=y, y_noise=None, n_std_thresh_stationary=float(rng.uniform(0.5, 2.5)),
                                        clip_noise_stationary=True, precision="float64", **base)
        else:
            sg = SpectralGateNonStationary(y=y, thresh_n_mult_nonstationary=float(rng.uniform(1, 3)),
                                           sigmoid_slope_nonstationary=float(rng.uniform(5, 15)), precision="float64", **base)
        a, b = (None, None)
        if rng.random() < 0.3 and n > 3 * cs:
            a = int(rng.integers(0, n // 3)); b = int(rng.integers(2 * n // 3, n))
        new = sg.get_traces(a, b)
        with sg._gate.with_options([(_ffi.SG_OPT_EXACT_MATERIALISED, 1)]):
            old = sg.get_traces(a, b)
        tot["x64_cases"] += 1
        if dtype == np.float64:
            e = float(np.nanmax(np.abs(new - old)) / max(np.nanmax(np.abs(old)), 1e-300))
            tot["x64_worst_f64"] = max(tot["x64_worst_f64"], e)
            if not (e < 1e-12) or not np.array_equal(np.isnan(new), np.isnan(old)):
                tot["failures"].append(("x64-f64", seed, stationary, cs, pad, n, C, e))
        else:
            d = np.abs(new.astype(np.int64) - old.astype(np.int64))
            # both are truncations of float64 values that agree to ~1e-13 relative: they may differ by one LSB where the value
            # sits within that of an integer -- rare
            frac = float(np.count_nonzero(d)) / d.size
            tot["x64_int_max_lsb"] = max(tot["x64_int_max_lsb"], int(d.max()))
            tot["x64_int_mismatch_decided"] = max(tot["x64_int_mismatch_decided"], frac)
            if d.max() > 1 or frac > (2e-4 if dtype == np.int32 else 1e-6):
                tot["failures"].append(("x64-int", seed, str(dtype), stationary, cs, pad, n, C, int(d.max()), frac))
    except Exception as ex:  # noqa: BLE001
        tot["failures"].append(("x64-exc", seed, repr(ex)))

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tot["x64_int_mismatch_fraction_max"] = tot.pop("x64_int_mismatch_decided")
json.dump(tot, open(os.path.join(ROOT, "gpurun_out", "fuzz_round5.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in tot.items() if k != "failures"}), "failures:", len(tot["failures"]))
for f in tot["failures"][:12]:
    print("  ", f)
