"""Timeline of one pipelined numpy->numpy call (base.py _get_traces_pipelined): when each upload and each gate ran."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1: os.environ["NOISEREDUCE_AMD_PIPELINE_PIECE_BYTES"] = str(int(sys.argv[1]) << 20)
import noisereduce_amd as nr
from noisereduce_amd.spectralgate import base
from noisereduce_amd import _ffi
n = 28_800_000
rng = np.random.default_rng(1234)
y = (0.1 * rng.standard_normal(n) + 0.5 * np.sin(2 * np.pi * 1000.0 * np.arange(n) / 48000.0)).astype(np.float32)
for _ in range(3): nr.reduce_noise(y=y, sr=48000, stationary=True)
log = []
Ev = lambda: torch.cuda.Event(enable_timing=True)
o_up, o_pc = base.SpectralGate._pipeline_upload, _ffi.Gate.process_chunks
def up(self, i):
    t0 = time.perf_counter(); o_up(self, i); log.append(("upload %d" % i, t0, time.perf_counter(), None, None))
def pc(self, x, **kw):
    e0, e1 = Ev(), Ev(); t0 = time.perf_counter(); e0.record(); r = o_pc(self, x, **kw); e1.record()
    log.append(("gate [%d, %d)" % (kw.get("start_frame", 0), kw.get("end_frame") or 0), t0, time.perf_counter(), e0, e1)); return r
base.SpectralGate._pipeline_upload, _ffi.Gate.process_chunks = up, pc
torch.cuda.synchronize()
eb = Ev(); eb.record(); tb = time.perf_counter()
out = nr.reduce_noise(y=y, sr=48000, stationary=True)
te = time.perf_counter()
torch.cuda.synchronize()
print("call %.3f ms" % ((te - tb) * 1e3))
for name, t0, t1, e0, e1 in log:
    print("%-28s host %.3f .. %.3f ms" % (name, (t0 - tb) * 1e3, (t1 - tb) * 1e3), "" if e0 is None else "  gpu %.3f .. %.3f ms" % (eb.elapsed_time(e0), eb.elapsed_time(e1)))
