"""Error of the float64 pipeline (SG_OPT_FORCE_EXACT) against the oracle for chosen fuzz_wide seeds.
usage: python tests/tools/exact_probe.py seed [seed ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv, seeds = sys.argv[:1], [int(a) for a in sys.argv[1:]]
import importlib.util
spec = importlib.util.spec_from_file_location("fw", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_wide.py"))
fw = importlib.util.module_from_spec(spec)
src = open(spec.origin).read().split("if __name__")[0]
exec(compile(src, spec.origin, "exec"), fw.__dict__)
import noisereduce_amd as nr
from noisereduce_amd import _ffi
from oracle import spectralgate_oracle as O
os.environ["SG_FORCE_EXACT_ALL"] = "1"
for seed in seeds:
    sr, C, n, dtype, noise, kw = fw.case_S(seed)
    y = np.stack([O.synth_signal(n, seed=seed * 5 + c, tone_hz=250.0 * (c + 1)).astype(np.float64) for c in range(C)])
    yi = np.round(y * 15000)
    if C == 1:
        yi = yi[0]
    want = O.reduce_noise_S(yi, sr, **kw)
    got_i = nr.reduce_noise(y=yi.astype(np.int16), sr=sr, **kw)
    diff = got_i.astype(np.int64) - np.trunc(want).astype(np.int64)
    frac = np.abs(want - np.round(want))
    print(seed, "n_fft", kw["n_fft"], "win", kw["win_length"], "hop", kw["hop_length"], "stat", kw["stationary"], "prop", kw["prop_decrease"],
          "| int16 mismatches", int(np.count_nonzero(diff)), "of", diff.size,
          "| samples of the ORACLE within 1e-6 of an integer:", int(np.count_nonzero(frac < 1e-6)),
          "| at mismatches: min/max distance of the oracle value from an integer", float(frac[diff != 0].min()) if np.any(diff) else None,
          float(frac[diff != 0].max()) if np.any(diff) else None, flush=True)
