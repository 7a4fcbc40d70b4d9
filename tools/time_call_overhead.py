"""Host-side cost of one sg_process_chunks / sg_noise_stats call (launch-bound regime: tiny input)."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd.sharded import HipStationaryBackend
dev = torch.device("cuda", 0)
b = HipStationaryBackend(48000, dev)
y = torch.randn(1, 2 * 600000, device=dev)
g = b.stats(y)
for _ in range(5): g.process_chunks(y, chunked=True)
torch.cuda.synchronize()
for name, f in (("process_chunks", lambda: g.process_chunks(y, chunked=True)), ("noise_stats", lambda: g.noise_stats(y[:, :600000]))):
    t0 = time.perf_counter()
    for _ in range(200): f()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(name, "host %.1f us/call, incl. drain %.1f us/call" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
