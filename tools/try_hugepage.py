"""Does a transparent-huge-page backed result array remove the first-touch page-fault cost of the
D2H copy?  (numpy result arrays are fresh mmap'ed pages: 28 k faults for 115 MB.)"""
import mmap, time, numpy as np, torch
n = 28_800_000
o = torch.randn(n, device="cuda")
torch.cuda.synchronize()
def T(f, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); del r
    return round(float(np.median(ts)), 2)
def fresh():
    out = np.empty(n, np.float32); torch.from_numpy(out).copy_(o); return out
def huge():
    m = mmap.mmap(-1, n * 4 + (2 << 20), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    m.madvise(mmap.MADV_HUGEPAGE)
    base = np.frombuffer(m, dtype=np.uint8)
    off = (-base.ctypes.data) % (2 << 20)
    out = base[off:off + n * 4].view(np.float32)
    torch.from_numpy(out).copy_(o); return out
def huge_populate():
    m = mmap.mmap(-1, n * 4 + (2 << 20), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    m.madvise(mmap.MADV_HUGEPAGE)
    try: m.madvise(23)   # MADV_POPULATE_WRITE (Linux 5.14)
    except OSError as e: print("populate:", e)
    base = np.frombuffer(m, dtype=np.uint8)
    off = (-base.ctypes.data) % (2 << 20)
    out = base[off:off + n * 4].view(np.float32)
    torch.from_numpy(out).copy_(o); return out
buf = np.empty(n, np.float32); buf[:] = 0
print("thp:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
print("D2H into warm array      ", T(lambda: torch.from_numpy(buf).copy_(o)))
print("np.empty + D2H           ", T(fresh))
print("THP mmap + D2H           ", T(huge))
print("THP mmap + populate + D2H", T(huge_populate))
pin = torch.empty(n, dtype=torch.float32, pin_memory=True)
print("D2H into pinned          ", T(lambda: pin.copy_(o, non_blocking=True)))
def via_pinned():
    pin.copy_(o, non_blocking=True); torch.cuda.synchronize(); out = np.empty(n, np.float32); out[:] = pin.numpy(); return out
print("pinned + memcpy to fresh ", T(via_pinned))
y = np.random.default_rng(0).standard_normal(n).astype(np.float32)
d = torch.empty(n, device="cuda")
print("H2D pageable             ", T(lambda: d.copy_(torch.from_numpy(y))))
def h2d_pinned():
    pin.numpy()[:] = y; d.copy_(pin, non_blocking=True)
print("memcpy to pinned + H2D   ", T(h2d_pinned))
