"""A/B of the non-stationary gate's tile chain: k_iir_chain_par (default) against k_iir_comb + k_iir_chain (SG_OPT_FORCE_SPLIT), same
process, alternating blocks of calls.  usage: python tools/ab_chain.py   (also under rocprofv3 --kernel-trace --stats)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
sr = 48000
y = bench.synth_on_device(sr * 600, 1, torch.device("cuda", 0))
for n_fft, secs in ((1024, 600), (256, 120), (512, 120), (2048, 120)):
    yy = y[:sr * secs].contiguous()
    def make():
        return SpectralGateNonStationary(y=yy, sr=sr, chunk_size=600000, padding=30000, prop_decrease=1.0, n_fft=n_fft, win_length=None,
                                         hop_length=None, time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50,
                                         thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, tmp_folder=None, use_tqdm=False, n_jobs=1)
    sg = make()
    for _ in range(300): sg.get_traces()
    torch.cuda.synchronize()
    res = {0: [], 1: []}
    for rnd in range(4):
        for mode in (0, 1):
            sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, mode)
            for _ in range(10): sg.get_traces()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40): sg.get_traces()
            b.record(); torch.cuda.synchronize()
            res[mode].append(a.elapsed_time(b) / 40)
    sg._gate.set_option(_ffi.SG_OPT_FORCE_SPLIT, 0)
    print(f"n_fft {n_fft} {secs} s: parallel chain {np.median(res[0]):.4f} ms {['%.4f' % t for t in res[0]]}  serial {np.median(res[1]):.4f} ms {['%.4f' % t for t in res[1]]}", flush=True)
