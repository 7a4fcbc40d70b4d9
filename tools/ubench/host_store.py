"""How fast do the gate kernels store into page-locked host memory (the download leg of base.py _get_traces_pipelined)?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import noisereduce_amd as nr
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
n = 28_800_000
rng = np.random.default_rng(1234)
y = (0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1000.0 * np.arange(n) / 48000.0)).astype(np.float32)
def T(f, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 3)
for name, yy in (("float32", y), ("int16", (y * 20000).astype(np.int16))):
    d = torch.from_numpy(yy).cuda()[None, :]
    host = torch.empty((1, n), dtype=d.dtype).pin_memory()
    devo = torch.empty_like(d)
    for stat in (True, False):
        os.environ["NOISEREDUCE_AMD_PIPELINE"] = "0"
        nr.reduce_noise(y=d[0], sr=48000, stationary=stat)
        from noisereduce_amd import _ffi
        # the engine handle the call above used
        sg = (SpectralGateStationary if stat else SpectralGateNonStationary)
        import inspect
        kw = dict(y=d[0], sr=48000, prop_decrease=1.0, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
                  tmp_folder=None, chunk_size=600000, padding=30000, n_fft=1024, win_length=None, hop_length=None, use_tqdm=False, n_jobs=1)
        if stat: kw.update(y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True)
        else: kw.update(thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10)
        o = sg(**kw)
        g = o._gate
        def run(out, a=0, b=n):
            with g.lock:
                o._bind(); g.process_chunks(d, out=out[:, a:b], start_frame=a, end_frame=b, chunked=True)
        print(name, "stationary" if stat else "non-stationary", ": out in HBM", T(lambda: run(devo)), " out in page-locked host", T(lambda: run(host)),
              " 6M-sample piece to host", T(lambda: run(host, 6000000, 12000000)), " D2H copy", T(lambda: host.copy_(devo, non_blocking=True)), flush=True)
