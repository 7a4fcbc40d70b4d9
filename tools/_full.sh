cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 900 python tests/tools/fuzz_wide.py 0 300 2>&1 | grep -v amdgpu | tail -4
