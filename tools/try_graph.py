import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
dev = torch.device("cuda", 0)
y = bench.synth_on_device(bench.N_PER_GPU, 1234, dev)[None, :]
kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
          clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None,
          time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
          use_tqdm=False, n_jobs=1)
sg = SpectralGateStationary(y=y, **kw); g = sg._gate
out = torch.empty_like(y)
def step():
    g.noise_stats(y[:, :600000])
    g.process_chunks(y, chunked=True, out=out)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 50
print(f"eager  {te*1e3:.4f} ms/step")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gr, stream=s):
        step()
    torch.cuda.synchronize()
    ref = out.clone(); out.zero_()
    gr.replay(); torch.cuda.synchronize()
    print("graph replay equals eager:", bool(torch.equal(ref, out)))
    t0 = time.perf_counter()
    for _ in range(50): gr.replay()
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50
    print(f"graph  {tg*1e3:.4f} ms/step")
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
