import time, numpy as np, torch
n = 28_800_000
y = np.random.default_rng(0).standard_normal(n).astype(np.float32)
d_in = torch.empty(n, dtype=torch.float32, device="cuda")
d_out = torch.randn(n, device="cuda")
h_out = torch.empty(n, dtype=torch.float32).pin_memory()
cur = torch.cuda.current_stream(); s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def T(f, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 3)
def both(k, order="d_first", wait=False, interleave=False, h2d_stream=None):
    m = n // k
    def d(i):
        with torch.cuda.stream(s2):
            if wait: s2.wait_event(cur.record_event())
            h_out[i*m:(i+1)*m].copy_(d_out[i*m:(i+1)*m], non_blocking=True)
    def h(i):
        with torch.cuda.stream(h2d_stream or s1):
            d_in[i*m:(i+1)*m].copy_(torch.from_numpy(y[i*m:(i+1)*m]), non_blocking=True)
    if interleave:
        for i in range(k): d(i); h(i)
    else:
        for i in range(k): d(i)
        for i in range(k): h(i)
for k in (2, 4, 6, 8, 10, 16, 24):
    print("k=%d (%.1f MB): all D2H then all H2D %.3f | interleaved %.3f | interleaved+wait_event %.3f | interleaved, H2D on the current stream %.3f"
          % (k, n * 4 / k / 1e6, T(lambda: both(k)), T(lambda: both(k, interleave=True)), T(lambda: both(k, interleave=True, wait=True)),
             T(lambda: both(k, interleave=True, h2d_stream=cur))))
