"""Which placement of upload / compute / download on streams overlaps the two PCIe directions?  (decides the shape of
base.py _get_traces_pipelined).  Full-size device buffers: no reuse hazards, so only true dependencies remain."""
import time, sys, numpy as np, torch
n = 28_800_000
y = np.random.default_rng(0).standard_normal(n).astype(np.float32)[None, :]
out_t = torch.empty((1, n), dtype=torch.float32).pin_memory()
x = torch.empty((1, n), dtype=torch.float32, device="cuda"); o = torch.empty_like(x)
cur = torch.cuda.current_stream(); s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def run(L, mode, compute=True, trace=False):
    pieces = [(p0, min(p0 + L, n)) for p0 in range(0, n, L)]
    tu, evs = [], []
    h2d_s = {"A": s1, "B": cur, "C": cur, "D": s1, "E": s1, "F": s1}[mode]
    cmp_s = {"A": cur, "B": cur, "C": cur, "D": cur, "E": s3, "F": cur}[mode]
    d2h_s = {"A": cur, "B": s2, "C": cur, "D": s2, "E": s3, "F": s2}[mode]
    def upload(i):
        p0, p1 = pieces[i]
        t0 = time.perf_counter()
        with torch.cuda.stream(h2d_s):
            x[:, p0:p1].copy_(torch.from_numpy(y[:, p0:p1]), non_blocking=True)
        h2d_s.synchronize()
        tu.append(time.perf_counter() - t0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    upload(0)
    for i, (p0, p1) in enumerate(pieces):
        with torch.cuda.stream(cmp_s):
            if compute: torch.mul(x[:, p0:p1], 2.0, out=o[:, p0:p1])
            if d2h_s is not cmp_s:
                if mode == "F":
                    cmp_s.synchronize()          # host-side wait instead of a device-side one
                else:
                    e = cmp_s.record_event()
        with torch.cuda.stream(d2h_s):
            if d2h_s is not cmp_s and mode != "F": d2h_s.wait_event(e)
            out_t[:, p0:p1].copy_(o[:, p0:p1], non_blocking=True)
        if i + 1 < len(pieces): upload(i + 1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, np.median(tu) * 1e3
names = {"A": "H2D side, gate+D2H current", "B": "H2D+gate current, D2H side (event)", "C": "all on current",
         "D": "H2D side1, gate current, D2H side2 (event)", "E": "H2D side1, gate+D2H side3", "F": "as D, host sync instead of event"}
for L in (1800000, 3000000, 3600000):
    for mode in "ABCDEF":
        for cmpt in (True, False):
            for _ in range(2): r = run(L, mode, cmpt)
            print("piece %.1f MB %-44s compute=%d : total %.3f ms, median upload %.3f ms" % (L * 4 / 1e6, names[mode], cmpt, *r))
