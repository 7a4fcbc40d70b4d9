cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in noisereduce_amd/_ab/*.so; do
  echo $f; SG_LIB_PATH=$PWD/$f python tools/prof_torchgate.py 2>/dev/null | tail -1 | cut -c1-200
  SG_LIB_PATH=$PWD/$f python bench.py --nonstationary --steps 20 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms_per_step'])"
done
