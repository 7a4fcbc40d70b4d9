import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, noisereduce_amd as nr
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
dev = torch.device("cuda", 0)
C, N = 8, 48000 * 1800
y = torch.empty((C, N), dtype=torch.float32, device=dev)
for c in range(C):
    y[c] = bench.synth_on_device(N, 1234 + c, dev, tone_hz=200.0 * (c + 1))
kw = dict(sr=48000, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=600000,
          clip_noise_stationary=True, padding=30000, n_fft=1024, win_length=None, hop_length=None,
          time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None,
          use_tqdm=False, n_jobs=1)
def sync(): torch.cuda.synchronize()
for _ in range(2): SpectralGateStationary(y=y, **kw).get_traces()
sync()
for rep in range(3):
    t0 = time.perf_counter(); sg = SpectralGateStationary(y=y, **kw); sync(); t1 = time.perf_counter()
    out = sg.get_traces(); t2 = time.perf_counter(); sync(); t3 = time.perf_counter()
    print(f"ctor {1e3*(t1-t0):.2f} ms | get_traces enqueue {1e3*(t2-t1):.2f} ms | +sync {1e3*(t3-t2):.2f} ms")
