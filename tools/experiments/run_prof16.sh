cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/proffb -- python $GRAFT_REPO_ROOT/tools/prof_fwdbwd.py > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/proffb -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -c1-200
