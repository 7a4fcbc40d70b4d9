"""(diagnosis, -DOP_TILECOUNT=1 library) persistent gate next to a matmul loop on a second stream: are all tickets taken up exactly once?"""
import os, sys, threading, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from noisereduce_amd import _ffi
from noisereduce_amd.spectralgate.stationary import SpectralGateStationary
from oracle import spectralgate_oracle as O
KW = dict(sr=48000, prop_decrease=1.0, chunk_size=100000, padding=8000, n_fft=1024, win_length=None, hop_length=None,
          time_constant_s=2.0, freq_mask_smooth_hz=500, time_mask_smooth_ms=50, tmp_folder=None, use_tqdm=False, n_jobs=1)
big = torch.from_numpy(np.stack([O.synth_signal(1500000, seed=10 + c, tone_hz=300.0 * (c + 1)) for c in range(4)]).astype(np.float32)).cuda()
ss = SpectralGateStationary(y=big, y_noise=None, n_std_thresh_stationary=1.5, clip_noise_stationary=True, **KW)
g = ss._gate
ref = ss.get_traces().clone()
g.set_option(_ffi.SG_OPT_TILE_ORDER, 0)
stop = False
def mm_loop():
    s2 = torch.cuda.Stream()
    mm = torch.randn(2048, 2048, device="cuda")
    with torch.cuda.stream(s2):
        while not stop: (mm @ mm).sum().item()
th = threading.Thread(target=mm_loop if not os.environ.get("NO_MM") else (lambda: None)); th.start()
c0, s0 = g.debug_counter(4), g.debug_counter(5)
per, bad, slow = None, 0, 0.0
for i in range(3000):
    t_call = time.perf_counter()
    try: out = ss.get_traces(); torch.cuda.synchronize(); dt_ms = (time.perf_counter() - t_call) * 1e3; slow = max(slow, dt_ms) if i > 20 else slow
    except Exception as e:
        print("call", i, "raised", type(e).__name__); bad += 1
        try: g.check_errors()
        except Exception: pass
        if bad >= 5: break
        continue
    c1, s1 = g.debug_counter(4), g.debug_counter(5)
    dc, ds = (c1 - c0) & 0xffffffff, (s1 - s0) & 0xffffffff
    c0, s0 = c1, s1
    if per is None: per = (dc, ds)
    ok = torch.equal(out, ref)
    if (dc, ds) != per or not ok:
        bad += 1
        print("call", i, "took %.2f ms; slowest good call before it %.3f ms" % (dt_ms, slow), flush=True)
        if os.environ.get("WHO") and bad == 1:   # (-DOP_WHO=1 library) the tickets around the first poll that gave up, before the next launch clears the record
            t8 = g.debug_counter(8)
            print("first bits poll that gave up: ticket", t8, "granule index", g.debug_counter(9), flush=True)
            g.debug_counter(16, t8)
            if os.environ.get("WHO") == "all": g.debug_counter(16, -1)
            t12 = g.debug_counter(12)
            print("first partial poll that gave up: ticket", t12, flush=True)
            g.debug_counter(16, t12)
        if os.environ.get("TRACE_DUMP"):   # (-DOP_TRACE=1 library) the tiles that did not run to the end, before the next launch clears the trace
            try: print("incomplete tiles:", g.debug_counter(8), flush=True)
            except Exception as e: print("trace dump:", str(e)[:80])
        if bad <= 5: print("call", i, "tickets taken", dc, "sum", ds, "expected", per, "output equal", ok)
    if bad >= 5: break
stop = True; th.join()
if os.environ.get("WHO"):   # (-DOP_WHO=1 library) the first poll that gave up: its tile, what it waited for, the tag it saw
    w = [g.debug_counter(k) for k in (9, 10, 11, 12, 13, 14, 15)]
    try: w8 = g.debug_counter(8)
    except Exception: w8 = -1
    print("who: bits poll gave up in ticket", w8, "granule index", w[0], "tag seen", w[1], "epoch", w[2], "count", w[5], "| partial poll gave up in ticket", w[3], "tag seen", w[4], "count", w[6], flush=True)
print("late total_tiles seen (max)", g.debug_counter(6), "at iteration (max)", g.debug_counter(7))
try: g.check_errors(); print("no hand-off error")
except Exception as e: print("ERR", str(e)[:80])
print("calls", i + 1, "bad", bad, "tickets per call", per)
