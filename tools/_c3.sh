cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "nonstat or non_stat or golden or config3 or iir" 2>&1 | tail -3
python bench.py --workload config3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3', d['ms_per_step'], d.get('ms_per_step_median'), d.get('parity_ok'), d.get('stages_ms'))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p3 -o p3 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload config3 --no-cpu-baseline --steps 10 > /dev/null 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/p3/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.reader(open(f)))[:9]: print(r[0][:50], r[1], r[3])
PY
