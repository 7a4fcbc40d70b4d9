#!/bin/bash
# On the GPU box: time every build under noisereduce_amd/_ab/ with bench.py, print the kernel table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for f in noisereduce_amd/_ab/*.so; do
  SG_LIB_PATH=$PWD/$f python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$f', 'step %.4f' % d['ms_per_step'], ' '.join('%s=%.4f' % (n.split(' ')[0], v) for n, v in k.items()))
"
done
