"""Host time of one bench step (enqueue only) vs its GPU time: is the step launch-bound?"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from noisereduce_amd.sharded import HipStationaryBackend, TimeShardedStationary
dev = torch.device("cuda", 0)
y = (0.1 * torch.randn(1, 28_800_000, device=dev))
b = HipStationaryBackend(48000, dev)
def step(): return TimeShardedStationary(b, 513).run(y)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.1f us/step, wall %.1f us/step" % ((t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e6))
