import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import noisereduce_amd as nr
n = 28_800_000
rng = np.random.default_rng(1234)
y = (0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1000.0 * np.arange(n) / 48000.0)).astype(np.float32)
for _ in range(3): out = nr.reduce_noise(y=y, sr=48000, stationary=True)
def T(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 2)
print("reduce_noise numpy->numpy", T(lambda: nr.reduce_noise(y=y, sr=48000, stationary=True)))
print("np.array(y)", T(lambda: np.array(y)))
print("np.asarray(y)", T(lambda: np.asarray(y)))
d = torch.from_numpy(y).cuda()
print("H2D", T(lambda: torch.from_numpy(y).to("cuda")))
print("compute (tensor in/out)", T(lambda: nr.reduce_noise(y=d, sr=48000, stationary=True)))
o = nr.reduce_noise(y=d, sr=48000, stationary=True)
print("D2H .cpu()", T(lambda: o.cpu()))
buf = np.empty_like(y)
print("D2H into existing numpy", T(lambda: torch.from_numpy(buf).copy_(o)))
print("np.empty_like + D2H", T(lambda: torch.from_numpy(np.empty_like(y)).copy_(o)))
print(".cpu().numpy().astype", T(lambda: o.cpu().numpy().astype(np.float32, copy=False)))
# piece-size sweep of the pipelined host path (base.py: _get_traces_pipelined)
os.environ["NOISEREDUCE_AMD_PIPELINE"] = "0"
ref = nr.reduce_noise(y=y, sr=48000, stationary=True).copy()
y16 = (y * 20000).astype(np.int16)
for pb in (0, 12, 24, 36, 48):
    if pb == 0:
        os.environ["NOISEREDUCE_AMD_PIPELINE"] = "0"
    else:
        os.environ["NOISEREDUCE_AMD_PIPELINE"] = "1"
        os.environ["NOISEREDUCE_AMD_PIPELINE_PIECE_BYTES"] = str(pb << 20)
    for _ in range(2): r = nr.reduce_noise(y=y, sr=48000, stationary=True)
    print("piece MB =", pb or "off", ": numpy->numpy", T(lambda: nr.reduce_noise(y=y, sr=48000, stationary=True), reps=9),
          "identical" if np.array_equal(r, ref) else "DIFFERENT",
          " non-stationary", T(lambda: nr.reduce_noise(y=y, sr=48000, stationary=False), reps=5),
          " int16", T(lambda: nr.reduce_noise(y=y16, sr=48000, stationary=True), reps=5),
          " 90 s recording", T(lambda: nr.reduce_noise(y=y[:4_320_000], sr=48000, stationary=True), reps=9),
          " 3 min", T(lambda: nr.reduce_noise(y=y[:8_640_000], sr=48000, stationary=True), reps=9), flush=True)
