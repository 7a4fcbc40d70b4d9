cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/prof_nfft.py 2048 512 2>&1 | grep -v amdgpu
rocprofv3 --kernel-trace --stats -d /tmp/p20 -o p20 --output-format csv -- python $GRAFT_REPO_ROOT/tools/prof_nfft.py 2048 > /dev/null 2>&1
F=$(find /tmp/p20 -name '*kernel_stats.csv' | head -1); head -25 $F | cut -d, -f1-5
