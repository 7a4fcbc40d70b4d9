"""One-pass stationary gate, BASELINE.json configs[1] (10 min mono 48 kHz): the a-priori floor test (k_unit_absmax before
the gate, SG_OPT_FLOOR_TEST = 1) against the in-kernel one (2) and the predicted default (0), same process, same data,
interleaved rounds.  Per mode: ms per reduce_noise (HIP events over 50 back-to-back calls) and the gate kernel's own time.

  python tools/floor_test_ab.py > gpurun_out/floor_test_ab.json
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                       # noqa: E402
from noisereduce_amd import _ffi                                                    # noqa: E402
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary          # noqa: E402


MODES = tuple(int(a) for a in sys.argv[1:]) or (1, 2, 0)


def main():
    dev = torch.device("cuda:0")
    y = bench.synth_on_device(bench.N_PER_GPU, 0, dev)
    sg = SpectralGateStationary(
        y=y, sr=bench.SR, y_noise=None, prop_decrease=1.0, n_std_thresh_stationary=1.5, chunk_size=bench.CHUNK,
        clip_noise_stationary=True, padding=bench.PAD, n_fft=bench.NFFT, win_length=None, hop_length=None,
        time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1,
        device=dev)
    gate = sg._gate
    outs = {}
    rows = {m: {"ms": [], "gate_ms": []} for m in MODES}
    for rnd in range(4):
        for mode in MODES:
            gate.set_option(_ffi.SG_OPT_FLOOR_TEST, mode % 10)
            gate.set_option(_ffi.SG_OPT_TILE_ORDER, mode // 10)   # modes 1x: tiles by block index instead of tickets
            for _ in range(30):
                o = sg.get_traces()
            torch.cuda.synchronize()
            if mode not in outs:
                outs[mode] = o.clone() if torch.is_tensor(o) else np.array(o)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                sg.get_traces()
            e1.record()
            e1.synchronize()
            rows[mode]["ms"].append(e0.elapsed_time(e1) / 50)
            gate.profile_read(reset=True)
            gate.profile_select(["k_gate_onepass (fft+decide+smooth+mask+ifft+ola)"])
            gate.profile_enable(True)
            for _ in range(20):
                sg.get_traces()
            prof = gate.profile_read(reset=True)
            gate.profile_enable(False)
            t, n = prof["k_gate_onepass (fft+decide+smooth+mask+ifft+ola)"]
            rows[mode]["gate_ms"].append(t / n)
    gate.set_option(_ffi.SG_OPT_FLOOR_TEST, 0)
    first = outs[MODES[0]]
    same = all((torch.equal(first, outs[m]) if torch.is_tensor(first) else np.array_equal(first, outs[m])) for m in outs)
    print(json.dumps({"workload": "configs[1]: 28.8 M samples, chunk 600000, padding 30000, n_fft 1024",
                      "outputs_identical": bool(same),
                      "modes": {{1: "a_priori", 2: "in_kernel", 0: "predicted", 12: "in_kernel_blockidx"}[m]:
                                {"ms_per_call_rounds": [round(v, 4) for v in r["ms"]], "ms_per_call_median": round(float(np.median(r["ms"])), 4),
                                 "gate_kernel_ms_rounds": [round(v, 4) for v in r["gate_ms"]]} for m, r in rows.items()},
                      "counters": {"in_kernel_batches": gate.debug_counter(1), "a_priori_batches": gate.debug_counter(2)}}, indent=1))


if __name__ == "__main__":
    main()
