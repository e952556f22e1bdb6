cd $GRAFT_REPO_ROOT
timeout 300 python tools/two_step_probe.py 2>&1 | grep -v amdgpu.ids
