#!/usr/bin/env python3
"""One-pass kernel vs the three-kernel path (SG_OPT_FORCE_SPLIT): outputs must be bit-identical (same
transforms, same decisions, same integer smoothing); prints timings of both."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__; __graft_entry__.build()
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from oracle import spectralgate_oracle as O
import bench

dev = torch.device("cuda", 0)
KW = dict(y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, clip_noise_stationary=True,
          n_fft=1024, win_length=None, hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500,
          time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
res = {}
def both(y, sr, cs, pad, label, reps=0):
    sg = SpectralGateStationary(y=y, sr=sr, chunk_size=cs, padding=pad, **KW)
    g = sg._gate
    g.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    a = sg.get_traces()
    a2 = sg.get_traces()
    g.set_option(_ffi.SG_OPT_FORCE_SPLIT, 1)
    b = sg.get_traces()
    g.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    eq = bool(torch.equal(a, b)) if isinstance(a, torch.Tensor) else bool(np.array_equal(a, b))
    det = bool(torch.equal(a, a2)) if isinstance(a, torch.Tensor) else bool(np.array_equal(a, a2))
    r = {"equal_to_split": eq, "deterministic": det}
    if not eq:
        d = (a.double() - b.double()).abs() if isinstance(a, torch.Tensor) else np.abs(a.astype(np.float64) - b)
        r["max_abs_diff"] = float(d.max()); r["n_diff"] = int((d > 0).sum())
    if reps:
        for mode in (0, 1):
            g.set_option(_ffi.SG_OPT_FORCE_SPLIT, mode)
            for _ in range(3): sg.get_traces()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): sg.get_traces()
            torch.cuda.synchronize()
            r["ms_split" if mode else "ms_onepass"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        g.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    res[label] = r
    return a

y = bench.synth_on_device(bench.N_PER_GPU, 1234, dev)
out = both(y, 48000, 600000, 30000, "config2", reps=20)
# oracle on chunk 5
yh = y.cpu().numpy().astype(np.float64)
thr, _, _ = O.noise_threshold_S(yh[None, :600000], 1024, 1024, 256, 1.5, 600000)
chunk = O.read_chunk(yh[None, :], 5 * 600000 - 30000, 6 * 600000 + 30000)
ref = O.gate_stationary_S(chunk, thr, 1024, 1024, 256, 1.0, O.smoothing_filter(5, 9))[0, 30000:630000]
res["config2"]["rel_err_chunk5"] = O.rel_err(out[5 * 600000:6 * 600000].cpu().numpy(), ref)
# odd shapes: small chunks, 2 channels, other sample rates (nt = 8 / nf = 5 at 44.1 kHz; nt = 4 at 24 kHz), float64 input
for i, (sr, n, cs, pad, C, dt) in enumerate([(48000, 130000, 40000, 5000, 1, np.float32), (44100, 200542, 600000, 30000, 1, np.float64),
                                      (48000, 99999, 20000, 3000, 2, np.float32), (24000, 77777, 30000, 2000, 1, np.float32),
                                      (48000, 6000, 600000, 30000, 1, np.float32), (48000, 300001, 100000, 0, 3, np.float32)]):
    yy = np.stack([O.synth_signal(n, sr=sr, seed=50 + i + c).astype(dt) for c in range(C)])
    if C == 1: yy = yy[0]
    a = both(yy, sr, cs, pad, f"case{i}_sr{sr}_n{n}_cs{cs}_C{C}_{np.dtype(dt).name}")
    want = O.reduce_noise_S(yy.astype(np.float64), sr, stationary=True, chunk_size=cs, padding=pad)
    res[f"case{i}_sr{sr}_n{n}_cs{cs}_C{C}_{np.dtype(dt).name}"]["rel_err"] = O.rel_err(a, want)
print(json.dumps(res, indent=1))
