"""Development: N calls of the one-pass gate at configs[1] with SG_OPT_FLOOR_TEST = argv[1] (for OP_TRACE builds:
SG_LIB_PATH=noisereduce_amd/_ab/lib_trace.so python tools/experiments/floor_mode_run.py 2)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                                       # noqa: E402
from noisereduce_amd import _ffi                                                    # noqa: E402
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary          # noqa: E402

dev = torch.device("cuda:0")
y = bench.synth_on_device(bench.N_PER_GPU, 0, dev)
sg = SpectralGateStationary(
    y=y, sr=bench.SR, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=bench.CHUNK,
    clip_noise_stationary=True, padding=bench.PAD, n_fft=bench.NFFT, win_length=None, hop_length=None,
    time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1,
    device=dev)
sg._gate.set_option(_ffi.SG_OPT_FLOOR_TEST, int(sys.argv[1]))
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    sg.get_traces()
torch.cuda.synchronize()
