import time, sys, numpy as np, torch
n = 28_800_000; k = 8; m = n // k
y = np.random.default_rng(0).standard_normal(n).astype(np.float32)
d_in = torch.empty(n, dtype=torch.float32, device="cuda"); d_out = torch.randn(n, device="cuda")
h_out = torch.empty(n, dtype=torch.float32).pin_memory()
cur = torch.cuda.current_stream(); s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def pat(sync_h2d, src_dep, shift, sync_kind="stream"):
    # D2H(i) then H2D(i + shift)
    for i in range(k):
        src = d_in if src_dep else d_out
        with torch.cuda.stream(s2): h_out[i*m:(i+1)*m].copy_(src[i*m:(i+1)*m], non_blocking=True)
        j = (i + shift) % k
        with torch.cuda.stream(s1): d_in[j*m:(j+1)*m].copy_(torch.from_numpy(y[j*m:(j+1)*m]), non_blocking=True)
        if sync_h2d:
            if sync_kind == "stream": s1.synchronize()
            else:
                e = s1.record_event(); e.synchronize()
def T(f, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(ts)), 3)
print("free (no sync, independent src)      ", T(lambda: pat(False, False, 0)))
print("no sync, D2H reads d_in piece i, H2D writes i+1", T(lambda: pat(False, True, 1)))
print("sync after each H2D, independent src ", T(lambda: pat(True, False, 0)))
print("event sync after each H2D, independent", T(lambda: pat(True, False, 0, "event")))
print("sync + dependent src                 ", T(lambda: pat(True, True, 1)))
# host time of one pageable H2D call (does it return before the copy is done?)
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(s1): d_in[0:m].copy_(torch.from_numpy(y[0:m]), non_blocking=True)
t1 = time.perf_counter(); s1.synchronize(); t2 = time.perf_counter()
print("pageable H2D call returns after %.3f ms, stream done after %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
