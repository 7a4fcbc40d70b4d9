#!/usr/bin/env bash
# Development: per-phase shader-clock trace of k_row_gate on configs[4] (256 x 16000).  Builds a SEPARATE library with
# -DRG_TRACE=1 under gpurun_out/ (the product library is untouched) and runs one forward on it.
#   usage (GPU box): tools/rowgate_trace.sh
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
out="$root/gpurun_out/rg_trace"
mkdir -p "$out"
cd "$root/noisereduce_amd/csrc"
for u in api nonstat_mask; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -c $u.hip -o "$out/$u.o" -DRG_TRACE=1 ${SG_HIPCC_FLAGS:-} &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -Wl,--version-script=exports.map "$out/api.o" "$out/nonstat_mask.o" -o "$out/libmi355gate.so" -ldl
cd "$root"
SG_LIB_PATH="$out/libmi355gate.so" python - <<'PY' 2>&1 | tee "$out/trace.txt"
import numpy as np, torch
from noisereduce_amd.torchgate import TorchGate
torch.manual_seed(0)
t = torch.arange(16000, device="cuda", dtype=torch.float64) / 16000
import os
B = int(os.environ.get("RG_ROWS", "256"))
x = (0.1 * torch.randn(B, 16000, device="cuda") + 0.5 * torch.sin(2 * np.pi * 440 * t).float()).float()
tg = TorchGate(sr=16000).cuda()
for _ in range(5):
    y = tg(x)
torch.cuda.synchronize()
print("done", float(y.abs().max()))
PY
