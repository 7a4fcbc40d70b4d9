#!/usr/bin/env python3
"""Randomised sweep of the round-5 short-frame kernels.
  (a) n_fft = 256 / hop 64 on fast256.hpp (four frames per register transform; k_iir_mask<25 / 34 / 37> where the smoothing
      width has an instantiation) against the SAME call on the LDS kernels with 16-lane teams (SG_OPT_FORCE_NOFAST) and, for
      recordings the numpy oracle finishes quickly, against the oracle: random sample rates / lengths / chunk grids /
      channel counts / prop_decrease / smoothing widths / dtypes, both gates, silent stretches, sub-ranges;
  (b) stationary decisions: mask bits of k_decide_fast256 against the all-float64 decision kernel (SG_OPT_FORCE_F64_DECIDE);
  (c) other short frames on the team kernels only (n_fft = 64 .. 256 with win_length < n_fft or a hop that is not n_fft / 4)
      against the oracle.
usage (GPU box): python tests/tools/fuzz_fast256.py [first_seed] [count]   -> gpurun_out/fuzz_fast256.json"""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
from oracle import spectralgate_oracle as O

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = "cuda:0"
tot = dict(cases=0, worst_fast_vs_lds=0.0, worst_vs_oracle=0.0, oracle_cases=0, bit_cases=0, bit_mismatches=0,
           team_cases=0, team_worst_vs_oracle=0.0, failures=[])


def sig(rng, shape, sr):
    n = shape[-1]
    t = np.arange(n) / sr
    x = rng.uniform(0.01, 0.3) * rng.standard_normal(shape)
    for _ in range(rng.integers(0, 3)):
        x = x + rng.uniform(0.05, 0.8) * np.sin(2 * np.pi * rng.uniform(60, sr / 2 - 60) * t + rng.uniform(0, 6.28))
    if rng.random() < 0.25:
        a = int(rng.integers(0, max(1, n - 2000)))
        x[..., a:a + int(rng.integers(300, 4000))] = 0.0            # a silent stretch
    if rng.random() < 0.15:
        a = int(rng.integers(0, max(1, n - 500)))
        x[..., a:a + 300] *= 150.0                                   # a burst
    return x


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))


def rel_nan(got, want):
    """rel_err where both hold NaN in the SAME places (digital silence in the non-stationary gate is 0 / 0 in the reference
    too, nonstationary.py:75): the NaN patterns must agree, the rest is compared."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        return float("nan")
    m = ~np.isnan(want)
    if not m.any():
        return 0.0
    return float(np.max(np.abs(got[m] - want[m])) / max(1e-30, np.max(np.abs(want[m]))))


def make(stationary, y, sr, n_fft, kw):
    base = dict(y=y, sr=sr, chunk_size=kw["chunk_size"], padding=kw["padding"], prop_decrease=kw["prop_decrease"], n_fft=n_fft,
                win_length=kw.get("win_length"), hop_length=kw.get("hop_length"), time_constant_s=kw["time_constant_s"],
                freq_mask_smooth_hz=kw["freq_mask_smooth_hz"], time_mask_smooth_ms=kw["time_mask_smooth_ms"], tmp_folder=None,
                use_tqdm=False, n_jobs=1)
    if stationary:
        return SpectralGateStationary(y_noise=None, n_std_thresh_stationary=kw["n_std"], clip_noise_stationary=True, **base)
    return SpectralGateNonStationary(thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, **base)


for seed in range(first, first + count):
    rng = np.random.default_rng(70000 + seed)
    try:
        sr = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000]))
        C = int(rng.choice([1, 1, 2, 3]))
        n = int(rng.integers(300, 120000))
        stationary = bool(rng.random() < 0.55)
        cs = int(rng.choice([600000, int(rng.integers(3000, 40000))]))
        kw = dict(chunk_size=cs, padding=int(rng.integers(256, 3000)), prop_decrease=float(rng.choice([1.0, 1.0, rng.uniform(0.2, 0.99)])),
                  time_constant_s=float(rng.choice([2.0, 0.5])), n_std=float(rng.choice([1.5, 0.5, 2.5])),
                  freq_mask_smooth_hz=rng.choice([500, 500, None, 1000]), time_mask_smooth_ms=rng.choice([50, 50, None, 20, 80]))
        if kw["freq_mask_smooth_hz"] is not None and kw["freq_mask_smooth_hz"] < sr / 128:
            kw["freq_mask_smooth_hz"] = None if rng.random() < 0.5 else int(sr / 128) + 1
        if kw["time_mask_smooth_ms"] is not None and kw["time_mask_smooth_ms"] < 64e3 / sr:
            kw["time_mask_smooth_ms"] = None
        dt = rng.choice(["f32", "f32", "f64", "i16"])
        x = sig(rng, (C, n) if C > 1 else (n,), sr)
        if dt == "i16":
            xt = torch.from_numpy(np.clip(x * 12000, -32768, 32767).astype(np.int16))
        elif dt == "f64":
            xt = torch.from_numpy(x.astype(np.float64))
        else:
            xt = torch.from_numpy(x.astype(np.float32))
        y = xt.to(dev)
        sg = make(stationary, y, sr, 256, kw)
        a = sg.get_traces().cpu().numpy().astype(np.float64)
        with sg._gate.with_options([(_ffi.SG_OPT_FORCE_NOFAST, 1)]):
            b = make(stationary, y, sr, 256, kw).get_traces().cpu().numpy().astype(np.float64)
        tol = 2e-6 if dt != "i16" else 1.01 / max(1.0, float(np.max(np.abs(b))))     # integers: both are the float64 pipeline
        d = rel_nan(a, b)
        tot["cases"] += 1
        tot["worst_fast_vs_lds"] = max(tot["worst_fast_vs_lds"], d if dt != "i16" else 0.0)
        if not (d <= tol) or not np.array_equal(np.isnan(a), np.isnan(b)):
            tot["failures"].append(dict(seed=seed, what="fast256 vs LDS", d=d, sr=sr, n=n, C=C, stat=stationary, dt=str(dt), kw={k: (None if v is None else float(v)) for k, v in kw.items()}))
        if n <= 40000 and dt != "i16":
            okw = dict(stationary=stationary, n_fft=256, chunk_size=kw["chunk_size"], padding=kw["padding"], prop_decrease=kw["prop_decrease"],
                       time_constant_s=kw["time_constant_s"], freq_mask_smooth_hz=kw["freq_mask_smooth_hz"],
                       time_mask_smooth_ms=kw["time_mask_smooth_ms"], n_std_thresh_stationary=kw["n_std"])
            want = O.reduce_noise_S(xt.numpy().astype(np.float64), sr, **okw)
            e = rel_nan(a, want)
            tot["oracle_cases"] += 1
            tot["worst_vs_oracle"] = max(tot["worst_vs_oracle"], float(e))
            if not e < 1e-4:
                tot["failures"].append(dict(seed=seed, what="fast256 vs oracle", e=float(e), sr=sr, n=n, C=C, stat=stationary, dt=str(dt)))
        if stationary and dt == "f32" and kw["prop_decrease"] == 1.0:
            bits = sg._gate.debug_field(3) if hasattr(sg._gate, "debug_field") else None
            sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 1)
            try:
                sg.get_traces()
                bits64 = sg._gate.debug_field(3)
            finally:
                sg._gate.set_option(_ffi.SG_OPT_FORCE_F64_DECIDE, 0)
            sg.get_traces()
            bits = sg._gate.debug_field(3)
            tot["bit_cases"] += 1
            if bits.shape != bits64.shape or not np.array_equal(bits, bits64):
                tot["bit_mismatches"] += 1
                tot["failures"].append(dict(seed=seed, what="decision bits", sr=sr, n=n))
    except Exception as ex:     # noqa
        tot["failures"].append(dict(seed=seed, what="exception (a)", err=repr(ex)))
    # ---------------------------------------------------------------- (c) team kernels on other short frames
    try:
        n_fft = int(rng.choice([64, 128, 256]))
        sr = int(rng.choice([8000, 16000]))
        win = int(rng.choice([n_fft, n_fft, n_fft // 2, n_fft - 7]))
        hop = int(rng.choice([win // 4, win // 4, win // 2, max(1, win // 3)]))
        n = int(rng.integers(n_fft + 1, 30000))
        stationary = bool(rng.random() < 0.5)
        x = sig(rng, (n,), sr).astype(np.float32)
        fm = 500 if 500 >= sr / (n_fft / 2) else None
        okw = dict(stationary=stationary, n_fft=n_fft, win_length=win, hop_length=hop, chunk_size=int(rng.choice([600000, 7000])),
                   padding=int(rng.integers(n_fft, 2000)), freq_mask_smooth_hz=fm, time_mask_smooth_ms=50 if 50 >= hop * 1e3 / sr else None)
        import noisereduce_amd as nr
        got = nr.reduce_noise(y=x, sr=sr, **okw)
        want = O.reduce_noise_S(x.astype(np.float64), sr, **okw)
        e = rel_nan(got, want)
        tot["team_cases"] += 1
        tot["team_worst_vs_oracle"] = max(tot["team_worst_vs_oracle"], e)
        if not e < 1e-4:
            tot["failures"].append(dict(seed=seed, what="team kernels vs oracle", e=e, n_fft=n_fft, win=win, hop=hop, sr=sr, n=n, stat=stationary))
    except Exception as ex:     # noqa
        tot["failures"].append(dict(seed=seed, what="exception (c)", err=repr(ex)))

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
tot["seeds"] = [first, first + count]
json.dump(tot, open(os.path.join(ROOT, "gpurun_out", "fuzz_fast256.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in tot.items() if k != "failures"}), "failures:", len(tot["failures"]))
for f in tot["failures"][:12]:
    print(f)
