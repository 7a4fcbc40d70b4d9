cd $GRAFT_REPO_ROOT
for t in degenerate_check nan_check onepass_check fuzz_sharded; do echo "== $t"; timeout 600 python tests/tools/$t.py 2>&1 | grep -v amdgpu | tail -3; done
echo "== fuzz_sweep"; timeout 900 python tests/tools/fuzz_sweep.py 2>&1 | grep -v amdgpu | tail -2
echo "== soak"; timeout 600 python tests/tools/soak_handoff.py 2>&1 | grep -v amdgpu | tail -3
