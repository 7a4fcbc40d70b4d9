"""Does an upload from a pageable numpy array overlap a download into page-locked memory?  (decides whether the
host path is worth pipelining: base.py _get_traces_pipelined)"""
import time, numpy as np, torch
n = 28_800_000
y = np.random.default_rng(0).standard_normal(n).astype(np.float32)
yp = torch.from_numpy(y).pin_memory()
d_in = torch.empty(n, dtype=torch.float32, device="cuda")
d_out = torch.randn(n, device="cuda")
h_out = torch.empty(n, dtype=torch.float32).pin_memory()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def T(f, reps=7):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 3)
def h2d_pageable(k=1):
    m = n // k
    for i in range(k):
        with torch.cuda.stream(s1): d_in[i*m:(i+1)*m].copy_(torch.from_numpy(y[i*m:(i+1)*m]), non_blocking=True)
def h2d_pinned(k=1):
    m = n // k
    for i in range(k):
        with torch.cuda.stream(s1): d_in[i*m:(i+1)*m].copy_(yp[i*m:(i+1)*m], non_blocking=True)
def d2h(k=1):
    m = n // k
    for i in range(k):
        with torch.cuda.stream(s2): h_out[i*m:(i+1)*m].copy_(d_out[i*m:(i+1)*m], non_blocking=True)
for k in (1, 4, 8):
    print("pieces", k)
    print("  H2D pageable", T(lambda: h2d_pageable(k)), " H2D pinned", T(lambda: h2d_pinned(k)), " D2H pinned", T(lambda: d2h(k)))
    print("  D2H then H2D pageable (concurrent streams)", T(lambda: (d2h(k), h2d_pageable(k))))
    print("  D2H then H2D pinned   (concurrent streams)", T(lambda: (d2h(k), h2d_pinned(k))))
# in-place registration of the caller's array
rt = torch.cuda.cudart()
t0 = time.perf_counter(); r = rt.cudaHostRegister(y.ctypes.data, y.nbytes, 0); t1 = time.perf_counter()
print("cudaHostRegister", r, round((t1 - t0) * 1e3, 3), "ms")
print("  H2D registered", T(lambda: h2d_pageable(1)), " with D2H", T(lambda: (d2h(1), h2d_pageable(1))), " 4 pieces with D2H", T(lambda: (d2h(4), h2d_pageable(4))))
t0 = time.perf_counter(); rt.cudaHostUnregister(y.ctypes.data); t1 = time.perf_counter()
print("cudaHostUnregister", round((t1 - t0) * 1e3, 3), "ms")
