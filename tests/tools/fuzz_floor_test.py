"""Randomised sweep of the one-pass gate's floor test: SG_OPT_FLOOR_TEST 1 (a-priori pass over the recording) against 2 (in
the gate kernel + second launch) on random recordings -- channels, length, chunk size, padding, amplitude profile, loud
bursts / digital silence / NaN / Inf at random places (often inside a chunk's padding), quiet or loud noise clips,
prop_decrease, sub-ranges (start_frame / end_frame).  The two must agree bit for bit (NaN for NaN); every 8th case is also
held against the float64 oracle.

  python tests/tools/fuzz_floor_test.py [first_seed] [count]   -> gpurun_out/fuzz_floor_test.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from noisereduce_amd import _ffi                                              # noqa: E402
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary    # noqa: E402
from oracle import spectralgate_oracle as O                                   # noqa: E402


def make_case(seed):
    rng = np.random.default_rng(seed)
    C = int(rng.choice([1, 1, 2, 3]))
    n = int(rng.integers(30000, 400000))
    cs = int(rng.integers(20000, 120000))
    pad = int(rng.integers(0, min(cs, 20000)))
    base = 10.0 ** rng.uniform(-6, -0.5)
    y = (base * rng.standard_normal((C, n))).astype(np.float32)
    for _ in range(int(rng.integers(0, 4))):                       # loud bursts / silent gaps
        a = int(rng.integers(0, n - 10)); L = int(rng.integers(1, 6000)); b = min(n, a + L)
        kind = rng.integers(0, 3)
        ch = int(rng.integers(0, C))
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
    noise = (10.0 ** rng.uniform(-8, -0.5) * rng.standard_normal(int(rng.integers(3000, 40000)))).astype(np.float32)
    prop = float(rng.choice([1.0, 1.0, 0.7]))
    sub = None
    if rng.integers(0, 4) == 0 and n > 60000:
        a = int(rng.integers(0, n // 2)); sub = (a, int(rng.integers(a + 5000, n)))
    return y, noise, cs, pad, prop, sub


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    res = {"cases": 0, "mismatch": [], "reported_cases": 0, "oracle_checked": 0, "worst_vs_oracle": 0.0, "seeds": [first, first + count]}
    for seed in range(first, first + count):
        y, noise, cs, pad, prop, sub = make_case(seed)
        kw = dict(sr=48000, y_noise=noise, prop_decrease=prop, n_std_thresh_stationary=1.5, chunk_size=cs,
                  clip_noise_stationary=True, padding=pad, n_fft=1024, win_length=None, hop_length=None,
                  time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False,
                  n_jobs=1)
        sg = SpectralGateStationary(y=y if y.shape[0] > 1 else y[0], **kw)
        g = sg._gate
        args = {} if sub is None else dict(start_frame=sub[0], end_frame=sub[1])
        try:
            g.set_option(_ffi.SG_OPT_FLOOR_TEST, 1)
            a = sg.get_traces(**args)
            g.set_option(_ffi.SG_OPT_FLOOR_TEST, 2)
            e0 = g.debug_counter(3)
            b = sg.get_traces(**args)
            res["reported_cases"] += int(g.debug_counter(3) != e0)
        finally:
            g.set_option(_ffi.SG_OPT_FLOOR_TEST, 0)
        res["cases"] += 1
        if not np.array_equal(a, b, equal_nan=True):
            res["mismatch"].append(seed)
        if seed % 8 == 0 and sub is None and np.isfinite(y).all():
            want = O.reduce_noise_S(y.astype(np.float64) if y.shape[0] > 1 else y[0].astype(np.float64), 48000, stationary=True,
                                    y_noise=noise.astype(np.float64), prop_decrease=prop, chunk_size=cs, padding=pad)
            peak = float(np.max(np.abs(want)))
            if peak > 0:
                res["oracle_checked"] += 1
                res["worst_vs_oracle"] = max(res["worst_vs_oracle"], float(np.max(np.abs(b - want))) / peak)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "fuzz_floor_test.json"), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
