"""Soak test of the in-launch hand-offs (ticket protocol of k_gate_onepass and k_apply_fast<LEAN>): three host threads on
three HIP streams run a stationary gate, a non-stationary gate and TorchGate forward + backward against each other for a
given number of seconds; every result must equal, bit for bit, the result of the same call run alone, and no handle may
report a lost hand-off.  usage: [N_FFT=512] python tests/tools/soak_handoff.py [seconds]   (N_FFT: the gates' frame length -- 512 / 256 /
2048 run the one-pass gates of round 6 and their seam kernels)"""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import spectralgate_oracle as O
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from noisereduce_amd.spectralgate.nonstationary import SpectralGateNonStationary
from noisereduce_amd.torchgate import TorchGate

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
KW = dict(sr=48000, prop_decrease=1.0, chunk_size=100000, padding=8000, n_fft=int(os.environ.get("N_FFT", "1024")), win_length=None, hop_length=None,
          time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
big = torch.from_numpy(np.stack([O.synth_signal(1500000, seed=10 + c, tone_hz=300.0 * (c + 1)) for c in range(4)]).astype(np.float32)).cuda()
mid = torch.from_numpy(np.stack([O.synth_signal(700000, seed=20 + c) for c in range(3)]).astype(np.float32)).cuda()
ss = SpectralGateStationary(y=big, y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, slot=1, **KW)
sn = SpectralGateNonStationary(y=mid, thresh_n_mult_nonstationary=2, sigmoid_slope_nonstationary=10, **KW)
if os.environ.get("TILE_ORDER"):   # SG_OPT_TILE_ORDER of the stationary gate (0 persistent workgroups, 2 one ticket per workgroup)
    from noisereduce_amd import _ffi
    ss._gate.set_option(_ffi.SG_OPT_TILE_ORDER, int(os.environ["TILE_ORDER"]))
tg = TorchGate(sr=16000).cuda()
x = torch.from_numpy(np.stack([O.synth_signal(16000, sr=16000, seed=s, tone_hz=440.0) for s in range(32)])).cuda()
ref_s, ref_n = ss.get_traces().clone(), sn.get_traces().clone()
xg = x.clone().requires_grad_()
y0 = tg(xg); y0.sum().backward()
ref_y, ref_g = y0.detach().clone(), xg.grad.clone()
torch.cuda.synchronize()
bad, counts = [], {"stationary": 0, "nonstationary": 0, "torchgate": 0}
t_end = time.time() + secs

def loop(name, stream, body):
    with torch.cuda.stream(stream):
        while time.time() < t_end and not bad:
            body()
            counts[name] += 1
    stream.synchronize()

def b_s():
    out = ss.get_traces()
    if not torch.equal(out, ref_s):
        d = ~((out == ref_s) | (torch.isnan(out) & torch.isnan(ref_s)))
        idx = d.nonzero()
        runs = []
        if idx.numel():   # (diagnosis) where: channel, first / last wrong sample, how many, a few values
            for ch in idx[:, 0].unique().tolist()[:4]:
                ii = idx[idx[:, 0] == ch][:, 1]
                runs.append((ch, int(ii.min()), int(ii.max()), int(ii.numel()), [round(float(v), 4) for v in out[ch, ii[:3]].tolist()],
                             [round(float(v), 4) for v in ref_s[ch, ii[:3]].tolist()]))
        bad.append(("stationary", float((out - ref_s).abs().max()), runs))
def b_n():
    out = sn.get_traces()
    if not torch.equal(out, ref_n): bad.append(("nonstationary", float((out - ref_n).abs().max())))
xs = x.clone().requires_grad_()
T_MODE = os.environ.get("T_MODE", "")   # (diagnosis) what the third thread runs: default TorchGate forward + backward; "fwd": forward only;
mm = torch.randn(2048, 2048, device="cuda") if T_MODE == "matmul" else None      # "matmul": no kernel of this library at all
def b_t():
    if T_MODE == "fwd":
        with torch.no_grad():
            y = tg(x)
        if not torch.equal(y, ref_y): bad.append(("torchgate fwd", float((y - ref_y).abs().max())))
        return
    if T_MODE == "matmul":
        (mm @ mm).sum().item()
        return
    xs.grad = None
    y = tg(xs); y.sum().backward()
    if not torch.equal(y.detach(), ref_y): bad.append(("torchgate fwd", float((y.detach() - ref_y).abs().max())))
    if not torch.equal(xs.grad, ref_g): bad.append(("torchgate bwd", float((xs.grad - ref_g).abs().max())))

which = os.environ.get("THREADS", "s,n,t").split(",")   # subset of the three loops (diagnosis)
th = [threading.Thread(target=loop, args=(n, torch.cuda.Stream(), b)) for n, b in (("stationary", b_s), ("nonstationary", b_n), ("torchgate", b_t)) if n[0] in which]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize()
for nm, gg in (("stationary", ss._gate), ("nonstationary", sn._gate)):
    try:
        gg.check_errors()
    except Exception as e:   # noqa: BLE001
        bad.append((nm, str(e)[:90]))
for g in tg._gates.values(): g.check_errors()
print("soak", secs, "s:", counts, "mismatches:", bad[:5] if bad else "none")
sys.exit(1 if bad else 0)
