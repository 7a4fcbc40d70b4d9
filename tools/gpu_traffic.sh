#!/bin/bash
# HBM traffic per kernel launch from PMC counters (separate passes for FETCH_SIZE / WRITE_SIZE, as
# MI355X_MICROARCH.md prescribes), calibrated on a 1 GiB device copy in the same run.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d "$OUT" -o "$C" --output-format csv -- python "$REPO/tools/traffic_probe.py" > "$OUT/$C.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, sys, json, collections, glob, os
out = sys.argv[1]
res = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(out, "**", C + "_counter_collection.csv"), recursive=True)
    if not f: print("missing", C); continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != C: continue
        per[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    # (launches that return at once -- the one-pass gate's second launch when no chunk reported its floor test, the
    # float64 floor pre-pass without flagged units -- would halve a per-launch average: only launches that move at least
    # 1 % of the kernel's largest count)
    def avg(v):
        w = [x for x in v if x >= 0.01 * max(v)] if max(v) > 0 else v
        return sum(w) / len(w)
    res[C] = {k: (avg(v), len(v), max(v)) for k, v in per.items()}
GiB = float(1 << 30)
def find(d, pat):
    return [(k, v) for k, v in d.items() if pat in k]
# calibration: the largest single copyBuffer dispatch is the 1 GiB clone (reads 1 GiB, writes 1 GiB).
# Counter unit is KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads
# (MI355X_MICROARCH.md, HBM section) -- the factor below measures exactly that.
cal_f = [v[2] for k, v in res.get("FETCH_SIZE", {}).items() if "copyBuffer" in k]
cal_w = [v[2] for k, v in res.get("WRITE_SIZE", {}).items() if "copyBuffer" in k]
kf = GiB / (max(cal_f) * 1024) if cal_f else None
kw = GiB / (max(cal_w) * 1024) if cal_w else None
print("calibration factors (true bytes / (counter*1024)): fetch", kf, "write", kw)
# (the gate's first launch; k_gate_onepass<4, false, false, true, false> is the second launch of the in-kernel floor test)
# (round 6: <WAVES, PROP, LOSE, REDO, PERSIST> -- the first launch is <..., false, false> by default, <..., false, true> with SG_OPT_TILE_ORDER 0)
names = {"k_gate_onepass<4, false, false, false, false>": "k_gate_onepass (fft+decide+smooth+mask+ifft+ola)",
         "k_gate_onepass<4, false, false, false, true>": "k_gate_onepass (fft+decide+smooth+mask+ifft+ola)",
         "k_unit_absmax": "k_unit_absmax+k_prep_thresh"}
traffic = {}; detail = {}
import re
def short_name(k):
    m = re.search(r"sg::(?:fast::|big::|exact::)?(k_[a-z0-9_]+(?:<[^(]*>)?)", k)
    return m.group(1) if m else None
allk = set(res.get("FETCH_SIZE", {})) | set(res.get("WRITE_SIZE", {}))
for k in sorted(allk):
    sn = short_name(k)
    if not sn: continue
    fr = res.get("FETCH_SIZE", {}).get(k); wr = res.get("WRITE_SIZE", {}).get(k)
    if not fr or not wr: continue
    fb = fr[0] * 1024 * (kf or 1.0); wb = wr[0] * 1024 * (kw or 1.0)
    detail[sn] = {"launches": fr[1], "fetch_bytes": int(fb), "write_bytes": int(wb), "total_bytes": int(fb + wb),
                  "raw_FETCH_SIZE": fr[0], "raw_WRITE_SIZE": wr[0]}
    for short, stage in names.items():
        if sn.startswith(short): traffic[stage] = int(fb + wb)
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
json.dump({"calibration": {"fetch_factor": kf, "write_factor": kw, "method": "1 GiB torch clone in the same run"},
           "workloads": "tools/traffic_probe.py: configs[1] stationary, configs[2] non-stationary (10 min mono each), "
                        "configs[4] TorchGate 256 x 16000 forward; per-launch averages",
           "kernels": detail}, open(os.path.join(out, "traffic_detail.json"), "w"), indent=1)
print(json.dumps(detail, indent=1))
PY
