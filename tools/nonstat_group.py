"""Non-stationary gate (BASELINE.json configs[2]): time per call against the number of chunks per batch.

The chunks of the grid are independent units, and every intermediate field of the gate (|X|, float mask) is written by one
kernel and read by the next.  A batch whose fields fit the 256 MiB Infinity Cache keeps them on the die; the default batch
(the whole 10-minute recording: 254 MB of |X| + 254 MB of mask) streams them through HBM.  This script times the same call
with sg_params.max_workspace_bytes chosen for 48 (everything), 24, 16, 12, 8, 6, 4 chunks per batch.

  python tools/nonstat_group.py > gpurun_out/nonstat_group.json
"""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd import _ffi                                                    # noqa: E402
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary    # noqa: E402

SR, N, CHUNK, PAD, NFFT = 48000, 28_800_000, 600000, 30000, 1024


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    y = (torch.randn(N, generator=g) * 0.1).to(dev)
    sgn = SpectralGateNonStationary(
        y=y, sr=SR, chunk_size=CHUNK, padding=PAD, n_fft=NFFT, win_length=None, hop_length=None,
        time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, thresh_n_mult_nonstationary=2,
        sigmoid_slope_nonstationary=10, tmp_folder=None, prop_decrease=1.0, use_tqdm=False, n_jobs=1, device=dev)
    base = sgn._gate
    kw = dict(variant=_ffi.SG_VARIANT_S, stationary=False, n_fft=NFFT, win_length=NFFT, hop_length=NFFT // 4,
              n_grad_freq=base.params.n_grad_freq, n_grad_time=base.params.n_grad_time, smooth_mask=True,
              chunk_size=CHUNK, padding=PAD, prop_decrease=1.0, nonstat_thresh=2.0, nonstat_slope=10.0,
              iir_b=base.params.iir_b)
    y2 = y[None, :]
    ref = base.process_chunks(y2, chunked=True)
    torch.cuda.synchronize()
    # workspace per chunk as the library sizes it (api.hip unit_bytes): cells * 18 + T * n * 4 + ...
    per_unit = base.workspace_bytes(1, CHUNK + 2 * PAD, chunked=False)
    out = {"per_unit_bytes": per_unit, "rows": []}
    for ub in (48, 24, 16, 12, 8, 6, 4, 3, 2):
        gate = _ffi.Gate(dev, max_workspace_bytes=int(per_unit * ub + per_unit // 2) if ub < 48 else 0, **kw)
        o = gate.process_chunks(y2, chunked=True)
        same = bool(torch.equal(o, ref))
        for _ in range(20):
            gate.process_chunks(y2, chunked=True)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gate.process_chunks(y2, chunked=True)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        gate.profile_read(reset=True)
        gate.profile_select(None)
        gate.profile_enable(True)
        for _ in range(3):
            gate.process_chunks(y2, chunked=True)
        prof = gate.profile_read(reset=True)
        gate.profile_enable(False)
        out["rows"].append({"chunks_per_batch_target": ub, "ms_median": float(np.median(ts)), "ms_min": float(min(ts)),
                            "identical_to_one_batch": same,
                            "stages_ms_per_call": {k: round(v[0] / 3, 4) for k, v in prof.items() if v[0] > 0},
                            "launch_scopes_per_call": {k: v[1] // 3 for k, v in prof.items() if v[0] > 0}})
        gate.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
