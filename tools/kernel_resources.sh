#!/usr/bin/env bash
# Per-kernel resource usage of the gfx950 code objects (VGPRs, AGPRs, spills, scratch, occupancy, LDS), from the
# compiler's own report (-Rpass-analysis=kernel-resource-usage).  No GPU needed.
#   usage: tools/kernel_resources.sh [out.txt]      (default profiles/kernel_resources.txt)
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$(realpath -m "${1:-$root/profiles/kernel_resources.txt}")"
tmp="$(mktemp -d)"
trap 'rm -rf "$tmp"' EXIT
cd "$root/noisereduce_amd/csrc"
for u in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden --cuda-device-only -c "$u" -o "$tmp/${u%.hip}.o" \
    -Xclang -target-feature -Xclang -packed-fp32-ops ${SG_HIPCC_FLAGS:-} -Rpass-analysis=kernel-resource-usage 2> "$tmp/${u%.hip}.log" &
done
wait
python3 - "$tmp" "$out" <<'PY'
import re, sys, glob, subprocess
rows = []
for f in sorted(glob.glob(sys.argv[1] + "/*.log")):
    cur = None
    for ln in open(f, errors="replace"):
        m = re.search(r"remark: .*Function Name: (\S+)", ln)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark: .*?\s{2,}([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
with open(sys.argv[2], "w") as o:
    o.write("# kernel | VGPRs | AGPRs | SGPRs | VGPR spill | SGPR spill | scratch B/lane | occupancy waves/SIMD | LDS B/block\n")
    for r, d in sorted(zip(rows, dem), key=lambda x: x[1]):
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*\)$", "", d)
        o.write("%s | %s | %s | %s | %s | %s | %s | %s | %s\n" % (
            d, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("SGPRs", "?"), r.get("VGPRs Spill", r.get("VGPR Spill", "?")),
            r.get("SGPRs Spill", r.get("SGPR Spill", "?")), r.get("ScratchSize", "?"), r.get("Occupancy", "?"),
            r.get("LDS Size", "?")))
print("wrote", sys.argv[2], len(rows), "kernels")
PY
