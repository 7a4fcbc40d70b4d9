"""Randomised sweeps of round 6's new code paths (deterministic seeds):

  A. mixed-radix frame lengths (mixed.hpp): random even n_fft = 2 N with N = 2^a 3^b 5^c 7^d 11^e 13^f <= 2048 (not a power of
     two), win_length <= n_fft, random hop, sample rate, channels, chunk grid, dtype, both gates, against the numpy oracle
     (float inputs: 1e-4 of peak; int16: the truncated oracle);
  B. the persistent one-pass gate (onepass.hpp PERSIST): random recordings at the default geometry -- channels, length
     (a few tiles ... more tiles than resident workgroups), chunk size, padding, prop_decrease, smoothing widths, sub-ranges --
     SG_OPT_TILE_ORDER 0 against 2 (one ticket-drawn tile per workgroup) bit for bit, every 6th case against the oracle;
  C. the floor test inside the decision kernels of n_fft = 512 / 256 / 2048 (thresh.hpp FloorLazy): SG_OPT_FLOOR_TEST 1
     (a priori) against 2 (in the kernel + REDO launch) on recordings with loud bursts, digital silence, NaN / Inf at random
     places, quiet or loud noise clips: bit for bit (NaN for NaN);
  D. the one-pass gates of n_fft = 512 / 256 / 2048 (onepass512.hpp, onepass256.hpp, onepass2048.hpp) against the kernels they
     replace (SG_OPT_FORCE_SPLIT: decide + smooth + apply): random sample rate (so: smoothing half-widths on both sides of the
     kernels' limits), channels, length, chunk grid, padding, sub-range, input dtype, bursts / silence / NaN: bit for bit, every
     6th case against the oracle.

  python tests/tools/fuzz_round6.py [first_seed] [count]   -> gpurun_out/fuzz_round6.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import noisereduce_amd as nr                                                  # noqa: E402
from noisereduce_amd import _ffi                                              # noqa: E402
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary    # noqa: E402
from oracle import spectralgate_oracle as O                                   # noqa: E402

SMOOTH_N = sorted({a * b * c * d
                   for a in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024)
                   for b in (1, 3, 9, 27, 81)
                   for c in (1, 5, 25, 125, 625)
                   for d in (1, 7, 11, 13, 49, 77, 91)
                   if 8 <= a * b * c * d <= 2048 and (a * b * c * d) & (a * b * c * d - 1)})


def case_a(seed):
    rng = np.random.default_rng(60000 + seed)
    N = int(rng.choice(SMOOTH_N))
    n_fft = 2 * N
    win = n_fft if rng.random() < 0.6 else int(rng.integers(max(4, n_fft // 2), n_fft + 1))
    hop = max(1, win // 4) if rng.random() < 0.6 else int(rng.integers(max(1, win // 8), max(2, win // 2 + 1)))
    sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
    C = int(rng.choice([1, 1, 2]))
    n = int(rng.integers(6 * n_fft + 100, max(6 * n_fft + 200, 50000)))
    cs = int(rng.integers(4 * n_fft, max(4 * n_fft + 1, 30000)))
    pad = int(rng.integers(0, 3 * n_fft))
    kw = dict(stationary=bool(rng.random() < 0.5), n_fft=n_fft, win_length=win, hop_length=hop, chunk_size=cs, padding=pad,
              prop_decrease=float(rng.choice([1.0, 1.0, 0.7])))
    kw["freq_mask_smooth_hz"] = float(rng.choice([1.5, 3.0, 5.5])) * sr / (n_fft / 2) + 1.0
    kw["time_mask_smooth_ms"] = float(rng.choice([1.5, 2.5, 6.0])) * hop / sr * 1000.0 + 0.01
    dtype = str(rng.choice(["float32", "float32", "float64", "int16"]))
    return sr, C, n, dtype, kw


def run_a(seed, res):
    sr, C, n, dtype, kw = case_a(seed)
    y = np.stack([O.synth_signal(n, sr=sr, seed=seed * 5 + c, tone_hz=sr / (17.0 + 6 * c)).astype(np.float64) for c in range(C)])
    y = np.round(y * 20000).astype(np.int16) if dtype == "int16" else y.astype(dtype)
    if C == 1:
        y = y[0]
    try:
        want = O.reduce_noise_S(y.astype(np.float64), sr, **kw)
    except ValueError:
        return
    got = nr.reduce_noise(y=y, sr=sr, **kw)
    res["a_cases"] += 1
    res["a_sizes"].add(kw["n_fft"])
    if dtype == "int16":
        diff = got.astype(np.int64) - want.astype(np.int16).astype(np.int64)
        decided = np.abs(want - np.round(want)) > 1e-9
        if np.max(np.abs(diff)) > 1 or np.count_nonzero(diff[decided]):
            res["a_fail"].append((seed, kw["n_fft"], "int16", int(np.count_nonzero(diff[decided]))))
    else:
        e = float(O.rel_err(got.astype(np.float64), want))
        res["a_worst"] = max(res["a_worst"], e)
        if not e < 1e-4:
            res["a_fail"].append((seed, kw["n_fft"], dtype, e))


def run_b(seed, res):
    rng = np.random.default_rng(70000 + seed)
    C = int(rng.choice([1, 1, 2, 4]))
    n = int(rng.integers(8000, 48000 * int(rng.choice([1, 3, 12, 40]))))
    cs = int(rng.choice([600000, 600000, int(rng.integers(12000, 200000))]))
    pad = int(rng.integers(0, min(cs, 30000) + 1))
    sr = int(rng.choice([44100, 48000, 48000, 84000]))
    prop = float(rng.choice([1.0, 1.0, 0.8, 0.35]))
    y = np.stack([O.synth_signal(n, sr=sr, seed=seed * 3 + c, tone_hz=300.0 * (c + 1)) for c in range(C)]).astype(np.float32)
    if rng.integers(0, 3) == 0:      # silent stretches and bursts: tiles with nothing / everything passing
        a = int(rng.integers(0, n - 100)); y[:, a:a + int(rng.integers(50, 20000))] = 0.0
        a = int(rng.integers(0, n - 100)); y[int(rng.integers(0, C)), a:a + int(rng.integers(50, 3000))] *= 30.0
    kw = dict(sr=sr, y_noise=None, prop_decrease=prop, n_std_thresh_stationary=1.5, chunk_size=cs, clip_noise_stationary=True,
              padding=pad, n_fft=1024, win_length=None, hop_length=None, time_constant_s=2.0,
              freq_mask_smooth_hz=float(rng.choice([500, 500, 200, 350])), time_mask_smooth_ms=float(rng.choice([50, 50, 20, 80])),
              tmp_folder=None, use_tqdm=False, n_jobs=1)
    try:
        sg = SpectralGateStationary(y=y if C > 1 else y[0], **kw)
    except ValueError:
        return
    g = sg._gate
    args = {}
    if rng.integers(0, 4) == 0 and n > 40000:
        a = int(rng.integers(0, n // 2)); args = dict(start_frame=a, end_frame=int(rng.integers(a + 3000, n)))
    try:
        g.set_option(_ffi.SG_OPT_TILE_ORDER, 0)
        a0 = sg.get_traces(**args)
        a1 = sg.get_traces(**args)
        g.set_option(_ffi.SG_OPT_TILE_ORDER, 2)
        b = sg.get_traces(**args)
    finally:
        g.set_option(_ffi.SG_OPT_TILE_ORDER, 0)   # (the default)
    g.check_errors()
    res["b_cases"] += 1
    if not (np.array_equal(a0, b, equal_nan=True) and np.array_equal(a0, a1, equal_nan=True)):
        res["b_fail"].append(seed)
    if seed % 6 == 0 and not args and n <= 48000 * 3:
        want = O.reduce_noise_S(y.astype(np.float64) if C > 1 else y[0].astype(np.float64), sr, stationary=True, chunk_size=cs,
                                padding=pad, prop_decrease=prop, freq_mask_smooth_hz=kw["freq_mask_smooth_hz"],
                                time_mask_smooth_ms=kw["time_mask_smooth_ms"])
        e = float(O.rel_err(a0, want))
        res["b_oracle"] += 1
        res["b_worst"] = max(res["b_worst"], e)
        if not e < 1e-4:
            res["b_fail"].append((seed, e))


def run_c(seed, res):
    rng = np.random.default_rng(80000 + seed)
    n_fft = int(rng.choice([256, 512, 2048]))
    C = int(rng.choice([1, 1, 2]))
    n = int(rng.integers(20 * n_fft, 300000))
    cs = int(rng.integers(8 * n_fft, 120000))
    pad = int(rng.integers(0, min(cs, 20000)))
    base = 10.0 ** rng.uniform(-6, -0.5)
    y = (base * rng.standard_normal((C, n))).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):
        a = int(rng.integers(0, n - 10)); b = min(n, a + int(rng.integers(1, 6000)))
        kind, ch = rng.integers(0, 3), int(rng.integers(0, C))
        if kind == 0:
            y[ch, a:b] = (10.0 ** rng.uniform(-1, 1) * rng.standard_normal(b - a)).astype(np.float32)
        elif kind == 1:
            y[ch, a:b] = 0.0
        else:
            y[ch, a:b] *= np.float32(1e-4)
    special = rng.integers(0, 8)
    if special == 0:
        y[int(rng.integers(0, C)), int(rng.integers(0, n))] = np.nan
    elif special == 1:
        y[int(rng.integers(0, C)), int(rng.integers(0, n))] = np.inf * (1 if rng.integers(0, 2) else -1)
    noise = (10.0 ** rng.uniform(-8, -0.5) * rng.standard_normal(int(rng.integers(4 * n_fft, 40000)))).astype(np.float32)
    kw = dict(sr=48000, y_noise=noise, prop_decrease=float(rng.choice([1.0, 1.0, 0.7])), n_std_thresh_stationary=1.5, chunk_size=cs,
              clip_noise_stationary=True, padding=pad, n_fft=n_fft, win_length=None, hop_length=None, time_constant_s=2.0,
              freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = SpectralGateStationary(y=y if C > 1 else y[0], **kw)
    g = sg._gate
    try:
        g.set_option(_ffi.SG_OPT_FLOOR_TEST, 1)
        a = sg.get_traces()
        g.set_option(_ffi.SG_OPT_FLOOR_TEST, 2)
        e0 = g.debug_counter(3)
        b = sg.get_traces()
        res["c_reported"] += int(g.debug_counter(3) != e0)
    finally:
        g.set_option(_ffi.SG_OPT_FLOOR_TEST, 0)
    res["c_cases"] += 1
    if not np.array_equal(a, b, equal_nan=True):
        res["c_fail"].append((seed, n_fft))
    if seed % 8 == 0 and np.isfinite(y).all():
        want = O.reduce_noise_S(y.astype(np.float64) if C > 1 else y[0].astype(np.float64), 48000, stationary=True,
                                y_noise=noise.astype(np.float64), chunk_size=cs, padding=pad, n_fft=n_fft,
                                prop_decrease=kw["prop_decrease"])
        e = float(O.rel_err(b, want))
        res["c_worst"] = max(res["c_worst"], e)
        if not e < 1e-4:
            res["c_fail"].append((seed, n_fft, e))


def run_d(seed, res):
    rng = np.random.default_rng(90000 + seed)
    n_fft = int(rng.choice([256, 512, 2048]))
    C = int(rng.choice([1, 1, 2, 3]))
    sr = int(rng.choice([48000, 48000, 44100, 32000, 22050, 96000]))
    n = int(rng.integers(12 * n_fft, int(rng.choice([60000, 300000, 1500000]))))
    cs = int(rng.choice([600000, int(rng.integers(6 * n_fft, 200000))]))
    pad = int(rng.integers(0, min(cs, 30000) + 1))
    y = np.stack([O.synth_signal(n, sr=sr, seed=seed * 7 + c, tone_hz=250.0 * (c + 1)) for c in range(C)]).astype(np.float32)
    kind = rng.integers(0, 5)
    if kind == 0:
        a = int(rng.integers(0, n - 100)); y[:, a:a + int(rng.integers(50, 20000))] = 0.0
        a = int(rng.integers(0, n - 100)); y[int(rng.integers(0, C)), a:a + int(rng.integers(50, 3000))] *= 300.0
    elif kind == 1:
        y[int(rng.integers(0, C)), int(rng.integers(0, n))] = np.nan
    dtype = str(rng.choice(["float32", "float32", "float64"]))
    prop = float(rng.choice([1.0, 1.0, 0.8, 0.35]))
    kw = dict(sr=sr, y_noise=None, prop_decrease=prop, n_std_thresh_stationary=float(rng.choice([1.5, 1.5, 0.5, 3.0])), chunk_size=cs,
              clip_noise_stationary=True, padding=pad, n_fft=n_fft, win_length=None, hop_length=None, time_constant_s=2.0,
              freq_mask_smooth_hz=float(rng.choice([500, 500, 200, 100])), time_mask_smooth_ms=float(rng.choice([50, 50, 20, 90])),
              tmp_folder=None, use_tqdm=False, n_jobs=1)
    yy = y.astype(dtype)
    try:
        sg = SpectralGateStationary(y=yy if C > 1 else yy[0], **kw)
    except ValueError:
        return
    g = sg._gate
    args = {}
    if rng.integers(0, 4) == 0 and n > 40000:
        a = int(rng.integers(0, n // 2)); args = dict(start_frame=a, end_frame=int(rng.integers(a + 3000, n)))
    try:
        a0 = sg.get_traces(**args)
        a1 = sg.get_traces(**args)
        g.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
        b = sg.get_traces(**args)
    finally:
        g.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    g.check_errors()
    res["d_cases"] += 1
    res["d_by_nfft"][str(n_fft)] = res["d_by_nfft"].get(str(n_fft), 0) + 1
    if prop == 1.0:
        same = np.array_equal(a0, b, equal_nan=True)
    else:   # (the split kernels form p K / ktot + (1 - p) as a float mask field: last-bit differences)
        fin = np.isfinite(a0)
        same = np.array_equal(fin, np.isfinite(b)) and (not fin.any() or float(np.max(np.abs(a0[fin] - b[fin]))) <= 2e-6 * max(1e-30, float(np.max(np.abs(b[fin])))))
    if not (same and np.array_equal(a0, a1, equal_nan=True)):
        res["d_fail"].append((seed, n_fft, prop))
    if seed % 6 == 0 and not args and n <= 300000 and np.isfinite(y).all():
        want = O.reduce_noise_S(yy.astype(np.float64) if C > 1 else yy[0].astype(np.float64), sr, stationary=True, chunk_size=cs,
                                padding=pad, n_fft=n_fft, n_std_thresh_stationary=kw["n_std_thresh_stationary"], prop_decrease=prop,
                                freq_mask_smooth_hz=kw["freq_mask_smooth_hz"], time_mask_smooth_ms=kw["time_mask_smooth_ms"])
        e = float(O.rel_err(a0, want))
        res["d_oracle"] += 1
        res["d_worst"] = max(res["d_worst"], e)
        if not e < 1e-4:
            res["d_fail"].append((seed, n_fft, e))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    res = {"seeds": [first, first + count], "a_cases": 0, "a_sizes": set(), "a_fail": [], "a_worst": 0.0,
           "b_cases": 0, "b_fail": [], "b_oracle": 0, "b_worst": 0.0, "c_cases": 0, "c_fail": [], "c_reported": 0, "c_worst": 0.0,
           "d_cases": 0, "d_by_nfft": {}, "d_fail": [], "d_oracle": 0, "d_worst": 0.0}
    for seed in range(first, first + count):
        run_a(seed, res)
        run_b(seed, res)
        run_c(seed, res)
        run_d(seed, res)
    torch.cuda.synchronize()
    res["a_sizes"] = sorted(res["a_sizes"])
    res["failures"] = len(res["a_fail"]) + len(res["b_fail"]) + len(res["c_fail"]) + len(res["d_fail"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fuzz_round6.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "a_sizes"}), "distinct n_fft:", len(res["a_sizes"]))
    return 1 if res["failures"] else 0


if __name__ == "__main__":
    sys.exit(main())
