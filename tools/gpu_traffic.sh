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
    res[C] = {k: (sum(v) / len(v), len(v), max(v)) for k, v in per.items()}
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
names = {"k_gate_onepass": "k_gate_onepass (fft+decide+smooth+mask+ifft+ola)",
         "k_apply_fast": "k_apply_fast (fft+mask+ifft+ola)", "k_decide_fast": "k_decide_fast (f32 stft + exact f64 refine)",
         "k_smooth_bits2": "k_smooth_f+k_smooth_t", "k_unit_absmax": "k_unit_absmax+k_prep_thresh",
         "k_stft<double": "noise statistics: k_stft<double>", "k_colstats1(": "noise statistics: k_colstats1"}
traffic = {}; detail = {}
for short, stage in names.items():
    fr = [v for k, v in res.get("FETCH_SIZE", {}).items() if short in k]
    wr = [v for k, v in res.get("WRITE_SIZE", {}).items() if short in k]
    if not fr or not wr: continue
    fb = fr[0][0] * 1024 * (kf or 1.0); wb = wr[0][0] * 1024 * (kw or 1.0)
    traffic[stage] = int(fb + wb)
    detail[short] = {"fetch_bytes": int(fb), "write_bytes": int(wb), "raw_FETCH_SIZE": fr[0][0], "raw_WRITE_SIZE": wr[0][0]}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
json.dump({"calibration": {"fetch_factor": kf, "write_factor": kw, "method": "1 GiB torch clone in the same run"},
           "kernels": detail}, open(os.path.join(out, "traffic_detail.json"), "w"), indent=1)
print(json.dumps(detail, indent=1))
PY
