"""One pass of (a) the free-running interleave that overlaps and (b) the dependent pipeline that does not, for
rocprofv3 --memory-copy-trace."""
import time, sys, numpy as np, torch
n = 28_800_000; k = 8; m = n // k
y = np.random.default_rng(0).standard_normal(n).astype(np.float32)
d_in = torch.empty(n, dtype=torch.float32, device="cuda"); d_out = torch.randn(n, device="cuda")
h_out = torch.empty(n, dtype=torch.float32).pin_memory()
cur = torch.cuda.current_stream(); s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def free():
    for i in range(k):
        with torch.cuda.stream(s2): h_out[i*m:(i+1)*m].copy_(d_out[i*m:(i+1)*m], non_blocking=True)
        with torch.cuda.stream(s1): d_in[i*m:(i+1)*m].copy_(torch.from_numpy(y[i*m:(i+1)*m]), non_blocking=True)
def dep():
    with torch.cuda.stream(s1): d_in[0:m].copy_(torch.from_numpy(y[0:m]), non_blocking=True)
    s1.synchronize()
    for i in range(k):
        with torch.cuda.stream(s2): h_out[i*m:(i+1)*m].copy_(d_in[i*m:(i+1)*m], non_blocking=True)
        if i + 1 < k:
            with torch.cuda.stream(s1): d_in[(i+1)*m:(i+2)*m].copy_(torch.from_numpy(y[(i+1)*m:(i+2)*m]), non_blocking=True)
            s1.synchronize()
for f in (free, dep, free, dep):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
    print(f.__name__, round((time.perf_counter() - t0) * 1e3, 3), "ms")
    time.sleep(0.01)
